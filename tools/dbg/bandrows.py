#!/usr/bin/env python3
"""Slowest rank's share of the bench frame under the band split, by band height (one GPU renders each share alone)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
W, H, spp, depth = 1920, 1080, 64, 8
acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
def run(p):
    best = 1e9
    for _ in range(3):
        acc.zero_(); torch.cuda.synchronize(); t0 = time.time()
        ds.render_device(cam, p, acc.data_ptr())
        torch.cuda.synchronize(); best = min(best, (time.time() - t0) * 1e3)
    return best
whole = run(hip.make_params(W, H, spp, depth))
for n in (8, 4, 2):
    for rows in (2, 4, 8, 16, 32):
        ts = [run(hip.make_params(W, H, spp, depth, band=(rows, r, n))) for r in range(n)]
        print("N=%d band_rows=%2d: slowest %.2f ms, fastest %.2f ms -> efficiency %.1f%%" % (n, rows, max(ts), min(ts), 100 * whole / (n * max(ts))), flush=True)
    t = run(hip.make_params(W, H, spp, depth, sample_range=(0, spp // n)))
    print("N=%d sample split: %.2f ms -> efficiency %.1f%%" % (n, t, 100 * whole / (n * t)), flush=True)
