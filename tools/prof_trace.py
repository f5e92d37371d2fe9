#!/usr/bin/env python3
"""Profile helper: one frame of a scene, per-depth closest-hit timings (and visit counters with --count).
Usage: python3 tools/prof_trace.py [scene] [w] [h] [spp] [depth] [--count] [--frames N]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import hip, scenes  # noqa: E402

a = [x for x in sys.argv[1:] if not x.startswith("--")]
scene = a[0] if a else "dragon1"
w, h, spp, depth = (int(x) for x in (a[1:5] + ["1920", "1080", "8", "8"][len(a[1:5]):]))
frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 2
sc = scenes.build(scene)
ds = hip.DeviceScene(sc.desc, 0)
flags = hip.RTX_RENDER_COUNT_VISITS if "--count" in sys.argv else 0
for f in range(frames):
    _, _, st = ds.render(sc.rtx_camera(), hip.make_params(w, h, spp, depth, flags=flags), want_accum=False, want_bgr=False)
out = {"scene": scene, "w": w, "h": h, "spp": spp, "depth": depth, "rays": st["rays_at_depth"],
       "ms": [round(x, 3) for x in st["ms_trace_at_depth"]],
       "mrays_s": [round(r / m / 1e3, 1) if m > 0 else None for r, m in zip(st["rays_at_depth"], st["ms_trace_at_depth"])],  # None: depth ran inside the tail launch
       "ms_trace": st["ms_trace"], "ms_shade": st["ms_shade"], "ms_sort": st["ms_sort"], "ms_total": st["ms_total"],
       "total_mrays_s": st["rays_total"] / st["ms_total"] / 1e3}
if flags:
    out["visits_per_ray"] = [st[k] / st["rays_total"] for k in ("visits_inner", "visits_leaf", "visits_prim")]
print(json.dumps(out))
