"""raytracer_amd — MI355X (gfx950) ray→scene query engine behind the render loop of deggua/raytracer.

    host.py    ctypes binding of librtx_host.so (scene construction, OBJ/BMP, SAH kd build, flattener)
    hip.py     ctypes binding of librt_hip.so   (HIP kernels behind the C ABI of include/rtx.h)
    scenes.py  the BASELINE.json configs as scene recipes
    assets.py  packed meshes / generated stand-ins for the reference's missing blobs
    build.py   compiles both libraries in-tree (hipcc --offload-arch=gfx950)
"""
__version__ = "0.1.0"
