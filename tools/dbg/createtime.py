#!/usr/bin/env python3
"""Phases of getting a scene onto the device: OBJ import, scene build with the kd-tree on the device, flatten, upload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import scenes, hip, host, assets
hip.device_count()
L = host.lib()
for name in sys.argv[1:] or ["dragon1", "dragon2"]:
    scenes.build("teapot" if name != "teapot" else "bunny").close()
    lines = scenes.recipe(name)
    obj = os.path.join(assets.cache_dir(), [l.split() for l in lines if l.startswith("mesh")][0][1])
    if not os.path.exists(obj):
        scenes.build(name, kd_device=0).close()  # writes the stand-in mesh once
    m = L.Mesh_New(); t0 = time.time(); assert L.Mesh_Import_OBJ_Path(m, obj.encode()); t_imp = time.time() - t0; L.Mesh_Delete(m)
    t0 = time.time(); sc = scenes.build(name, kd_device=0); t1 = time.time()
    d = sc.desc; t2 = time.time()
    ds = hip.DeviceScene(d, 0); t3 = time.time()
    print("%s: OBJ import %.3f s; scene build (import + add + device kd %.0f ms) %.3f s, flatten %.3f s, rtx_scene_create %.3f s, %d MB on device"
          % (name, t_imp, sc.kd_ms_device, t1 - t0, t2 - t1, t3 - t2, ds.device_bytes() >> 20), flush=True)
    ds.close(); sc.close()
