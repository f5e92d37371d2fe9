#!/usr/bin/env python3
"""Rank 0's share of the bench frame at N = 8 (band split): wall time and closest-hit time per depth. Env knobs apply."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
W, H, spp, depth = 1920, 1080, 64, 8
acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
n = int(os.environ.get("SHARE_N", "8"))
best = None
for _ in range(4):
    acc.zero_(); torch.cuda.synchronize(); t0 = time.time()
    if os.environ.get("SHARE_SPLIT", "tile") == "sample":
        prm = hip.make_params(W, H, spp, depth, sample_range=(0, spp // n))
    else:
        prm = hip.make_params(W, H, spp, depth, band=(int(os.environ.get("SHARE_BAND", "16")), 0, n) if n > 1 else None)
    st = ds.render_device(cam, prm, acc.data_ptr())
    torch.cuda.synchronize(); dt = (time.time() - t0) * 1e3
    if best is None or dt < best[0]: best = (dt, st)
dt, st = best
print("%s N=%d wall %.2f | raygen %.2f resolve %.2f | trace %.2f shade %.2f sort %.2f | per depth %s" % (os.environ.get("TAG", ""), n, dt, st["ms_raygen"], st["ms_resolve"], st["ms_trace"], st["ms_shade"], st["ms_sort"], [round(x, 2) for x in st["ms_trace_at_depth"]]), flush=True)
