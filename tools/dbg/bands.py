#!/usr/bin/env python3
"""One rank's share of the bench frame under the band split, against the whole frame / N (strong-scaling headroom)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
W, H, spp, depth = 1920, 1080, 64, 8
acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
def run(band):
    best = None
    for _ in range(3):
        acc.zero_(); torch.cuda.synchronize(); t0 = time.time()
        st = ds.render_device(cam, hip.make_params(W, H, spp, depth, band=band), acc.data_ptr())
        torch.cuda.synchronize(); dt = (time.time() - t0) * 1e3
        if best is None or dt < best[0]: best = (dt, st)
    return best
whole = run(None)
print("whole frame: wall %.1f ms, device %.1f ms, rays %d" % (whole[0], whole[1]["ms_total"], whole[1]["rays_total"]))
for n in [int(x) for x in os.environ.get("BANDS_N", "2,4,8").split(",")]:
    worst = 0
    for r in range(n):
        dt, st = run((16, r, n))
        worst = max(worst, dt)
        print("  N=%d rank %d: wall %.1f ms device %.1f ms rays %d (%.1f%% of the frame's)" % (n, r, dt, st["ms_total"], st["rays_total"], 100.0 * st["rays_total"] / whole[1]["rays_total"]))
    print("N=%d: slowest rank %.1f ms -> efficiency %.0f%% before the reduce" % (n, worst, 100.0 * whole[0] / (n * worst)))
for band in (None, (16, 0, 8)):
    dt, st = run(band)
    print(band, "wall %.1f | trace %.1f shade %.1f raygen %.1f sort %.1f resolve %.2f | sum %.1f | per depth %s" % (dt, st["ms_trace"], st["ms_shade"], st["ms_raygen"], st["ms_sort"], st["ms_resolve"],
          st["ms_trace"] + st["ms_shade"] + st["ms_raygen"] + st["ms_sort"] + st["ms_resolve"], [round(x, 2) for x in st["ms_trace_at_depth"]]))
