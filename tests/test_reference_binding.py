"""The reference-side binding (oracle/flatten_ref.{h,c}: Rtx_Flatten + Render_Start_HIP, the code INTEGRATION.md §2 tells
a maintainer of the reference to add) compiled against the UNMODIFIED reference and linked into oracle/_ref/ref_parity.

CPU: Rtx_Flatten walks the reference's own Scene / KDTree / Texture / Material structs (private layouts re-declared
     and static_assert-ed against the real ones in ref_driver.c); every pool it produces must equal, byte for byte,
     what the product's Scene_Flatten hands to rtx_scene_create for the same recipe.
GPU: the reference's render loop (Render_New, Render_Done polling, Render_Delete, its Scene, Camera, ImageRGB) with the
     one call changed to Render_Start_HIP renders the frame through librt_hip.so: the image is rtx_render's.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import assets, hip, scenes
from raytracer_amd.host import LIB_DIR

SCENES = ["teapot", "bunny", "pear", "materials", "dragon1"]
needs_ref = pytest.mark.skipif(O.ref_binary("parity") is None, reason="oracle/_ref/ref_parity is built only where /root/reference exists")


def _arr(ptr, ctype, n):
    return np.ctypeslib.as_array(ptr, (max(n, 1),))[:n] if n else np.zeros(0, ctype)


@needs_ref
@pytest.mark.parametrize("scene", SCENES)
def test_reference_side_flattener_equals_scene_flatten(built_scenes, scene, tmp_path):
    out = str(tmp_path / "flat.bin")
    O.ref_run("parity", scenes.write_recipe(scene), assets.cache_dir(), "flatten", out)
    raw = open(out, "rb").read()
    magic, nn, nr, root, T, S, P, M, K, max_depth, abi = struct.unpack_from("<11I", raw, 0)
    texels, = struct.unpack_from("<Q", raw, 44)
    sky = struct.unpack_from("<6i", raw, 52)
    assert magic == 0x314C4652
    d = built_scenes(scene).desc.contents
    assert (nn, nr, root, T, S, P, M, K, max_depth, abi, texels) == (d.n_nodes, d.n_refs, d.root_index, d.n_tris, d.n_spheres, d.n_planes,
                                                                      d.n_materials, d.n_textures, d.max_depth, d.abi_version, d.tex_texels)
    assert list(sky) == list(d.skybox_tex)
    pos = 76
    pools = [("node_w0", d.node_w0, 4 * nn), ("node_w1", d.node_w1, 4 * nn), ("leaf_ref", d.leaf_ref, 4 * nr),
             ("tri_pos", d.tri_pos, 36 * T), ("tri_nrm", d.tri_nrm, 36 * T), ("tri_uv", d.tri_uv, 24 * T), ("tri_mat", d.tri_mat, T), ("tri_obj", d.tri_obj, 4 * T),
             ("sph", d.sph, 16 * S), ("sph_mat", d.sph_mat, S), ("sph_obj", d.sph_obj, 4 * S), ("plane", d.plane, 24 * P), ("plane_mat", d.plane_mat, P),
             ("materials", d.materials, 64 * M), ("textures", d.textures, 16 * K), ("tex_data", d.tex_data, 12 * texels)]
    for name, ptr, nbytes in pools:
        ours = C.string_at(ptr, nbytes) if nbytes else b""
        assert raw[pos:pos + nbytes] == ours, name
        pos += nbytes
    assert pos == len(raw)
    assert T > 0 and M >= 2 and K >= 6


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("scene,w,h,spp,depth,devices", [("materials", 128, 72, 4, 8, None), ("dragon1", 96, 54, 8, 8, None), ("dragon1", 96, 54, 8, 8, "0,0,0")])
def test_reference_render_loop_through_render_start_hip(built_scenes, device_scenes, scene, w, h, spp, depth, devices, tmp_path):
    """devices = "0,0,0": RTX_DEVICES makes Render_Start_HIP share the frame among three scene handles through rtx_render_multi
    (tile split; on the one-GPU box all on device 0) — the image is still rtx_render's, byte for byte."""
    out = str(tmp_path / "frame.bin")
    env = dict(os.environ, RTX_HIP_LIB=os.path.join(LIB_DIR, "librt_hip.so"), RTX_SEED_LO="1", RTX_SEED_HI="2")
    if devices:
        env["RTX_DEVICES"] = devices
    exe = O.ref_binary("parity")
    r = subprocess.run([exe, scenes.write_recipe(scene), assets.cache_dir(), "render_hip", str(w), str(h), str(spp), str(depth), out],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    img = np.fromfile(out, np.uint8).reshape(h, w, 3)
    bgr, _, _ = device_scenes(scene).render(built_scenes(scene).rtx_camera(), hip.make_params(w, h, spp, depth, seed=(1, 2)))
    assert np.array_equal(img, bgr)
    assert img.std() > 5  # a picture, not a blank


def test_integration_md_holds_the_compiled_binding():
    """INTEGRATION.md §2 shows the binding a maintainer adds: it must be the very files that are compiled and tested."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(repo, "INTEGRATION.md")).read()
    for f in ("flatten_ref.h", "flatten_ref.c"):
        assert open(os.path.join(repo, "oracle", f)).read() in md, f
