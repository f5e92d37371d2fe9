#!/usr/bin/env python3
"""Phases of getting a scene onto the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import scenes, hip
hip.device_count()
for name in sys.argv[1:] or ["dragon1", "dragon2"]:
    t0 = time.time(); sc = scenes.build(name, kd_device=0); t1 = time.time()
    d = sc.desc; t2 = time.time()
    ds = hip.DeviceScene(d, 0); t3 = time.time()
    ds2 = hip.DeviceScene(d, 0); t4 = time.time()
    print("%s: build (import + device kd) %.2f s, flatten %.2f s, rtx_scene_create %.2f s (second %.2f s), %d MB on device" % (name, t1 - t0, t2 - t1, t3 - t2, t4 - t3, ds.device_bytes() >> 20), flush=True)
    ds.close(); ds2.close(); sc.close()
