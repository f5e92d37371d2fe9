#!/usr/bin/env python3
"""Whole frame and a rank's share at N = 2, 4, 8, 64 with the library RTX_HIP_LIB names (for same-box A/B of two builds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
W, H, spp, depth = 1920, 1080, 64, 8
acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
def run(p, reps=6):
    best = 1e9
    for _ in range(reps):
        acc.zero_(); torch.cuda.synchronize(); t0 = time.time()
        ds.render_device(cam, p, acc.data_ptr())
        torch.cuda.synchronize(); best = min(best, (time.time() - t0) * 1e3)
    return best
row = ["whole %.2f" % run(hip.make_params(W, H, spp, depth))]
for n in (2, 4, 8, 64):
    row.append("1/%d %.2f" % (n, run(hip.make_params(W, H, spp, depth, band=(8, 0, n)))))
print(os.path.basename(os.environ.get("RTX_HIP_LIB", "default")), "  ".join(row), flush=True)
