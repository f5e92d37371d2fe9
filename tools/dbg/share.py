#!/usr/bin/env python3
"""One rank's share (band split N=8, rank 0) of the bench frame, a few times: for the kernel-trace timeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
W, H, spp, depth = 1920, 1080, 64, 8
acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for _ in range(3):
    acc.zero_(); torch.cuda.synchronize(); t0 = time.time()
    st = ds.render_device(cam, hip.make_params(W, H, spp, depth, band=(16, 0, n)), acc.data_ptr())
    torch.cuda.synchronize(); print("wall %.2f ms" % ((time.time() - t0) * 1e3), flush=True)
