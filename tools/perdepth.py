#!/usr/bin/env python3
"""Per-depth statistics of the closest-hit kernel on the bench frame (COUNT build): rays, kd visits and the wave statistics
(loop iterations, INNER / LEAF / POP block runs and the lanes they served) of every ray depth.  The counters of a frame
are sums over its launches; paths are deterministic, so the figures of depth d are the frame rendered with max_depth
d + 1 minus the frame rendered with max_depth d.   python tools/perdepth.py [scene] > profiles/rNN_perdepth.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import hip, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "dragon1"
W, H, SPP, D = 1920, 1080, 64, 8
sc = scenes.build(scene)
ds = hip.DeviceScene(sc.desc, 0)
cam = sc.rtx_camera()
acc = torch.zeros(H, W, 3, device="cuda:0")
hip.tune(TAIL_MAX=0)  # a pass per depth
prev, rows = None, []
timed = ds.render_device(cam, hip.make_params(W, H, SPP, D), acc.data_ptr())
timed = ds.render_device(cam, hip.make_params(W, H, SPP, D), acc.data_ptr())
for d in range(1, D + 1):
    st = ds.render_device(cam, hip.make_params(W, H, SPP, d, flags=hip.RTX_RENDER_COUNT_VISITS), acc.data_ptr())
    cur = dict(inner=st["visits_inner"], leaf=st["visits_leaf"], prim=st["visits_prim"], culled=st["visits_culled"], ws=st["wave_stats"],
               early=st["rays_early_miss"])
    p = prev or dict(inner=0, leaf=0, prim=0, culled=0, ws=[0] * 8, early=0)
    ws = [a - b for a, b in zip(cur["ws"], p["ws"])]
    rays = st["rays_at_depth"][d - 1] if len(st["rays_at_depth"]) >= d else 0
    # rays answered in the shade pass (early) never reach the kernel; the kernel's count at depth d-1:
    early_here = cur["early"] - p["early"]  # counted at the depth the ray WOULD have been traced
    rows.append(dict(depth=d - 1, rays=rays, inner=cur["inner"] - p["inner"], leaf=cur["leaf"] - p["leaf"], prim=cur["prim"] - p["prim"],
                     children_culled=cur["culled"] - p["culled"], iterations=ws[0], inner_runs=ws[1], inner_lanes=ws[2], leaf_runs=ws[3], leaf_lanes=ws[4],
                     pop_runs=ws[5], pop_lanes=ws[6], fin_runs=ws[7], ms_trace=timed["ms_trace_at_depth"][d - 1] if d - 1 < len(timed["ms_trace_at_depth"]) else None))
    prev = cur
# rays_at_depth includes the early-answered rays of that depth: take them out with the per-depth early counters of the full frame
full = ds.render_device(cam, hip.make_params(W, H, SPP, D, flags=hip.RTX_RENDER_COUNT_VISITS), acc.data_ptr())
out = {"workload": "%s %dx%d %dspp d%d" % (scene, W, H, SPP, D), "rays_early_miss_total": full["rays_early_miss"], "per_depth": rows}
for r in rows:
    n = max(r["rays"], 1)
    r["per_ray"] = {k: r[k] / n for k in ("inner", "leaf", "prim", "children_culled")}
    r["inner_lanes_per_run"] = r["inner_lanes"] / max(r["inner_runs"], 1)
    r["leaf_lanes_per_run"] = r["leaf_lanes"] / max(r["leaf_runs"], 1)
    r["pop_lanes_per_run"] = r["pop_lanes"] / max(r["pop_runs"], 1)
    r["prims_per_leaf_run"] = r["prim"] / max(r["leaf_runs"], 1)
print(json.dumps(out, indent=1))
