#!/usr/bin/env python3
"""The five BASELINE configs at their full size on one GPU: s/frame, Mrays/s, kernel times (config 5: one rank's share,
128 of 1024 samples).   python tools/configs_table.py > profiles/rNN_configs.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import hip, scenes

CONFIGS = [("1 teapot primary", "teapot", 1920, 1080, 1, 1, None), ("2 bunny metal on diffuse ground", "bunny", 1920, 1080, 64, 8, None),
           ("3 dragon stand-in (277 804 tris) Disney glass + clearcoat", "dragon1", 1920, 1080, 64, 8, None),
           ("3' dragon stand-in (1 111 216 tris)", "dragon2", 1920, 1080, 64, 8, None),
           ("4 pear textured, cube map", "pear", 1920, 1080, 64, 16, None),
           ("5 dragon stand-in 4K, one rank's share: samples [0, 128) of 1024", "dragon1", 3840, 2160, 1024, 8, (0, 128))]
rows = []
for name, scene, w, h, spp, depth, srange in CONFIGS:
    sc = scenes.build(scene)
    with hip.DeviceScene(sc.desc, 0) as ds:
        acc = torch.zeros(h, w, 3, device="cuda:0")
        p = hip.make_params(w, h, spp, depth, sample_range=srange)
        best = None
        for _ in range(3):
            acc.zero_()
            st = ds.render_device(sc.rtx_camera(), p, acc.data_ptr())
            if best is None or st["ms_total"] < best["ms_total"]:
                best = st
        rows.append(dict(config=name, scene=scene, triangles=int(sc.desc.contents.n_tris), width=w, height=h, spp=spp, depth=depth, sample_range=srange,
                         s_per_frame=best["ms_total"] / 1e3, rays=best["rays_total"], mrays_s=best["rays_total"] / best["ms_total"] / 1e3,
                         ms=dict(trace=best["ms_trace"], shade=best["ms_shade"], binning=best["ms_sort"], raygen=best["ms_raygen"], resolve=best["ms_resolve"]),
                         slabs=best["passes"]))
    del acc
print(json.dumps(rows, indent=1))
