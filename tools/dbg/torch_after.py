import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from raytracer_amd import hip, scenes
mode = sys.argv[1]
sc = scenes.build("teapot")
ds = hip.DeviceScene(sc.desc, 0)
if mode == "render":
    ds.render(sc.rtx_camera(), hip.make_params(96, 54, 8, 8))
elif mode == "trace":
    ds.trace_closest(np.zeros((16, 3), np.float32), np.ones((16, 3), np.float32))
elif mode == "bounce":
    pass
import torch
print(mode, "device_count", torch.cuda.device_count(), flush=True)
try:
    x = torch.zeros(4, device="cuda:0")
    print(mode, "torch ok", flush=True)
except Exception as e:
    print(mode, "torch FAILED", str(e)[:100], flush=True)
