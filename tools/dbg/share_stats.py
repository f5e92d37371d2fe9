#!/usr/bin/env python3
"""Wave statistics of the camera pass and the first bounce on one rank's share: share_stats.py [N]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
W, H, spp = 1920, 1080, 64
acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
hip.tune(TAIL_MAX=0)
for N in [int(a) for a in sys.argv[1:]] or [1, 8, 64]:
    band = (8, 0, N) if N > 1 else None
    kw = dict(band=band) if band else {}
    prev = [0] * 8
    for d in (1, 2, 3):
        for _ in range(2): t = ds.render_device(cam, hip.make_params(W, H, spp, d, **kw), acc.data_ptr())
        st = ds.render_device(cam, hip.make_params(W, H, spp, d, flags=hip.RTX_RENDER_COUNT_VISITS, **kw), acc.data_ptr())
        ws = [a - b for a, b in zip(st["wave_stats"], prev)]; prev = st["wave_stats"]
        ms = t["ms_trace_at_depth"][d - 1]
        print("N=%2d depth %d: rays %9d  %.3f ms  iterations %9d (%.1f per wave of 6144)  us per iteration*6144waves %.2f  inner lanes/run %.1f" %
              (N, d - 1, st["rays_at_depth"][d - 1], ms, ws[0], ws[0] / 6144, ms * 1e3 / max(ws[0] / 6144, 1), ws[2] / max(ws[1], 1)), flush=True)
