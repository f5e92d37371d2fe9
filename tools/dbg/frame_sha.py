#!/usr/bin/env python3
"""sha256 of the f32 accumulation buffer of a few frames (RTX_HIP_LIB selects the library): two builds that claim the same
bits print the same lines.    python tools/dbg/frame_sha.py [scene ...]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from raytracer_amd import hip, scenes
for name in (sys.argv[1:] or ["dragon1", "materials", "pear", "bunny", "teapotdof"]):
    sc = scenes.build(name); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
    for (w, h, spp, depth) in ((480, 270, 16, 8), (1920, 1080, 4, 12)):
        bgr, acc, st = ds.render(cam, hip.make_params(w, h, spp, depth, seed=(7, 9)), want_accum=True)
        print(name, w, h, spp, depth, hashlib.sha256(np.ascontiguousarray(acc).tobytes()).hexdigest()[:24], int(sum(st["rays_at_depth"])), flush=True)
    ds.close(); sc.close()
