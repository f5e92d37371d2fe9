#!/usr/bin/env python3
"""Which camera rays of the bench frame make the longest walks?  One ray per pixel, traced in batches with visit counts;
the heaviest batches are bisected down to single rays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
W, H = 1920, 1080
p = hip.make_params(W, H, 64, 8)
pix = np.arange(W * H, dtype=np.uint32)
o, d, _ = ds.camera_rays(cam, p, pix, np.zeros(W * H, np.uint32))
def steps(idx):
    r = ds.trace_closest(o[idx], d[idx], flags=hip.RTX_RENDER_COUNT_VISITS)
    s = r["stats"]
    return s["visits_inner"] + s["visits_leaf"], r
B = 4096
tot = []
for b in range(0, W * H, B):
    idx = np.arange(b, min(b + B, W * H))
    tot.append((steps(idx)[0] / len(idx), b))
tot.sort(reverse=True)
print("mean steps per ray %.1f; heaviest batches: %s" % (np.mean([t for t, _ in tot]), [(round(t, 1), b) for t, b in tot[:5]]))
for _, b in tot[:6]:
    idx = np.arange(b, min(b + B, W * H))
    while len(idx) > 1:
        h = len(idx) // 2
        a, c = steps(idx[:h])[0], steps(idx[h:])[0]
        idx = idx[:h] if a / h >= c / (len(idx) - h) else idx[h:]   # follow the heavier half (per ray)
    n, r = steps(idx)
    i = int(idx[0])
    print("pixel (%d, %d): %d steps (inner %d, leaves %d, primitives %d, children culled %d), hit %#x t %.4f, o %s d %s" %
          (i % W, i // W, n, r["stats"]["visits_inner"], r["stats"]["visits_leaf"], r["stats"]["visits_prim"], r["stats"]["visits_culled"],
           int(r["prim_id"][0]), float(r["t"][0]), o[i], d[i]), flush=True)
# the distribution over a sample of single rays
rng = np.random.default_rng(1)
samp = rng.choice(W * H, 3000, replace=False)
st = np.array([steps(np.array([i]))[0] for i in samp])
print("single-ray sample: mean %.1f, median %d, p99 %d, p99.9 %d, max %d" % (st.mean(), np.median(st), np.percentile(st, 99), np.percentile(st, 99.9), st.max()))
