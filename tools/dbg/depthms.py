import os, sys
sys.path.insert(0, os.getcwd())
import torch
from raytracer_amd import hip, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "dragon1"
sc = scenes.build(scene)
ds = hip.DeviceScene(sc.desc, 0)
acc = torch.zeros(1080, 1920, 3, device="cuda:0")
p = hip.make_params(1920, 1080, int(os.environ.get("SPP", "64")), int(os.environ.get("DEPTH", "8")))
for _ in range(2):
    st = ds.render_device(sc.rtx_camera(), p, acc.data_ptr())
print(scene, os.environ.get("RTX_CULL_RO"), "total %.2f trace %.2f" % (st["ms_total"], st["ms_trace"]), [round(x, 2) for x in st["ms_trace_at_depth"]], st["rays_at_depth"], "early", st["rays_early_miss"])
