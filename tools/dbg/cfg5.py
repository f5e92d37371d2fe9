#!/usr/bin/env python3
"""BASELINE config 5, one rank's share: dragon stand-in 3840x2160, 1024 spp sample-split 8 ways (128 spp per GPU)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import hip, scenes
from raytracer_amd.distributed import FramePlan
sc = scenes.build("dragon1")
ds = hip.DeviceScene(sc.desc, 0)
plan = FramePlan(3840, 2160, 1024, 8, split="sample")
p = plan.rank_params(int(sys.argv[1]) if len(sys.argv) > 1 else 0, 8)
t0 = time.time()
_, _, st = ds.render(sc.rtx_camera(), p, want_accum=False, want_bgr=False)
print(json.dumps({"wall_s": time.time() - t0, "ms_total": st["ms_total"], "passes": st["passes"], "rays": st["rays_total"],
                  "mrays_s": st["rays_total"] / st["ms_total"] / 1e3, "ms_trace": st["ms_trace"], "ms_shade": st["ms_shade"], "ms_sort": st["ms_sort"]}))
