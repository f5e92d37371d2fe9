#!/usr/bin/env python3
"""Wave statistics of the closest-hit kernel per depth (COUNT build): how often each block runs and for how many lanes.
usage: wavestats.py [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
W, H, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 16
prev = np.zeros(8, np.int64); prevv = np.zeros(4, np.int64)
for depth in range(1, 6):
    _, _, st = ds.render(cam, hip.make_params(W, H, spp, depth, flags=hip.RTX_RENDER_COUNT_VISITS), want_accum=False, want_bgr=False)
    ws = np.array(st["wave_stats"], np.int64); vv = np.array([st["visits_inner"], st["visits_leaf"], st["visits_prim"], st["visits_culled"]], np.int64)
    w, v, n = ws - prev, vv - prevv, st["rays_at_depth"][-1]
    prev, prevv = ws, vv
    it = max(w[0], 1)
    print("depth %d rays %d | steps/ray node %.1f prim %.1f | wave iterations %d, rays per iteration-wave %.2f | INNER runs %.2f/iter, %.1f lanes | LEAF runs %.2f/iter, %.1f lanes, %.1f tests | POP %.2f/iter, %.1f lanes | FIN %.2f/iter"
          % (depth - 1, n, (v[0] + v[1] + v[3]) / n, v[2] / n, w[0], n / it, w[1] / it, w[2] / max(w[1], 1), w[3] / it, w[4] / max(w[3], 1), v[2] / max(w[3], 1), w[5] / it, w[6] / max(w[5], 1), w[7] / it))
