#!/usr/bin/env python3
"""One rank's share of the bench frame under an N-way band split, by tuning knob: share_knobs.py [N] (default 8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from raytracer_amd import hip, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
p = hip.make_params(W, H, spp, depth, band=(8, 0, N))
sweeps = [{}] + [{"tile_interleave": v} for v in (0, 1024, 16384)] + [{"trace_chunk": v} for v in (16, 32, 128)] + [{"band_tiles": 0}] + \
         [{"tail_max": v} for v in (0, 1 << 20, 4 << 20, 8 << 20)] + [{"sort_min": v} for v in (1000000, 1500000, 5000000)] + [{"ahead": 0}, {}]
for kn in sweeps:
    hip.tune(**kn)
    print("%-28s %.2f ms" % (kn or "defaults", run(p)), flush=True)
