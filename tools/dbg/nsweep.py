#!/usr/bin/env python3
"""Closest-hit time against queue size (primary rays, depth 1): where is the fixed cost of a launch?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
os.environ["RTX_OVERLAP"] = "0"
for spp in [int(x) for x in os.environ.get("NS_SPP", "64,16,4,1").split(",")]:
    for (w, h) in [tuple(int(y) for y in x.split("x")) for x in os.environ.get("NS_RES", "1920x1080,960x540,480x270").split(",")]:
        best = 1e9
        for _ in range(3):
            _, _, st = ds.render(cam, hip.make_params(w, h, spp, 1), want_accum=False, want_bgr=False)
            best = min(best, st["ms_trace_at_depth"][0])
        n = w * h * spp
        print("%4dx%-4d %2d spp: %9d rays  %.3f ms  %.2f Grays/s" % (w, h, spp, n, best, n / best / 1e6), flush=True)
