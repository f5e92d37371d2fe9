#!/usr/bin/env python3
"""Two overlapped parts (OVERLAP = 1) with the per-queue thresholds scaled to the halved queues, against one part."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
W, H, spp, depth = 1920, 1080, 64, 8
acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
def run(p, reps=5):
    best = 1e9
    for _ in range(reps):
        acc.zero_(); torch.cuda.synchronize(); t0 = time.time()
        ds.render_device(cam, p, acc.data_ptr())
        torch.cuda.synchronize(); best = min(best, (time.time() - t0) * 1e3)
    return best
for kn in ({}, {"overlap": 1}, {"overlap": 1, "tail_max": 1 << 20}, {"overlap": 1, "tail_max": 1 << 20, "sort_min": 750000}, {"overlap": 1, "tail_max": 0},
           {"overlap": 1, "ahead": 0}, {"overlap": 1, "tile_interleave": 0}, {}):
    hip.tune(**kn)
    print("%-60s whole %.2f  1/8 %.2f" % (kn or "defaults", run(hip.make_params(W, H, spp, depth)), run(hip.make_params(W, H, spp, depth, band=(8, 0, 8)))), flush=True)
