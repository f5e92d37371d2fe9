"""Per-depth closest-hit times of the bench frame under several hand-over points (knob LONG_AFTER), same process."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1")
ds = hip.DeviceScene(sc.desc, 0)
cam = sc.rtx_camera()
dev = torch.device("cuda", 0)
W, H = 1920, 1080
acc = torch.zeros(H, W, 3, device=dev)
band = None
if len(sys.argv) > 1 and sys.argv[1] == "share8":
    band = (8, 0, 8)
p = hip.make_params(W, H, 64, 8, band=band)
for la in [int(x) for x in (sys.argv[2:] or ["-1", "100000", "256", "128"])]:
    hip.tune(LONG_AFTER=la)
    best = None
    for _ in range(4):
        acc.zero_()
        st = ds.render_device(cam, p, acc.data_ptr())
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    print("LONG_AFTER %7d: total %.2f trace %.2f | per depth %s" % (la, best["ms_total"], best["ms_trace"], " ".join("%.2f" % x for x in best["ms_trace_at_depth"])))
