#!/usr/bin/env python3
"""Host phases of a scene build: OBJ import, Mesh_AddToScene, Scene_Prepare (host kd), Scene_Flatten."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import scenes, host, assets
L = host.lib()
for name in sys.argv[1:] or ["dragon2"]:
    lines = scenes.recipe(name)
    mesh_ln = [l.split() for l in lines if l.startswith("mesh")][0]
    path = os.path.join(assets.cache_dir(), mesh_ln[1])
    if not os.path.exists(path):
        scenes.build("teapot").close()
    t0 = time.time()
    sc = scenes.BuiltScene([l for l in lines if not l.startswith("mesh")] )if False else None
    mesh = L.Mesh_New()
    t0 = time.time(); ok = L.Mesh_Import_OBJ_Path(mesh, path.encode()); t1 = time.time()
    assert ok, host.last_error()
    print("%s: import %.3f s (%d triangles, %.1f MB)" % (name, t1 - t0, L.Mesh_Triangle_Count(mesh), os.path.getsize(path) / 1e6), flush=True)
    L.Mesh_Delete(mesh)
    t0 = time.time(); b = scenes.build(name); t1 = time.time()
    d = b.desc; t2 = time.time()
    print("%s: whole host build %.3f s, flatten %.3f s" % (name, t1 - t0, t2 - t1), flush=True)
    b.close()
