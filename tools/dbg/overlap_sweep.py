#!/usr/bin/env python3
"""Frame time of the whole bench frame and of one rank's share (band split) by the OVERLAP knob (parts on separate streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
W, H, spp, depth = 1920, 1080, 64, 8
acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
def run(p, reps=4):
    best = 1e9
    for _ in range(reps):
        acc.zero_(); torch.cuda.synchronize(); t0 = time.time()
        ds.render_device(cam, p, acc.data_ptr())
        torch.cuda.synchronize(); best = min(best, (time.time() - t0) * 1e3)
    return best
for ov in (0, 1, 2, 3, 0):
    hip.tune(overlap=ov)
    row = ["OVERLAP=%d" % ov, "whole %.2f" % run(hip.make_params(W, H, spp, depth))]
    for n in (2, 4, 8):
        row.append("1/%d %.2f" % (n, max(run(hip.make_params(W, H, spp, depth, band=(8, r, n))) for r in (0, n - 1))))
    print("  ".join(row), flush=True)
