import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from raytracer_amd import scenes, hip
from util import golden, ulp_diff
rec = golden("primary_teapot_A.npz")["rec"]
sc = scenes.build("teapot"); ds = hip.DeviceScene(sc.desc, 0)
got = ds.trace_closest(rec["o"], rec["d"], flags=hip.RTX_TRACE_UNIT_DIR)
d = ulp_diff(got["t"], rec["t"])
print("max", d.max(), "nonzero frac", (d>0).mean())
for i in list(range(3)) + list(np.argsort(d)[-5:]):
    print(i, hex(rec["id"][i]), rec["d"][i], "t ref", rec["t"][i].view(np.uint32), float(rec["t"][i]), "gpu", got["t"][i].view(np.uint32), float(got["t"][i]),
          "np", float(np.float32(-20)/rec["d"][i][2]), "ulp", d[i])
plane = rec["id"] == 0x80000000
print("plane max", d[plane].max(), "tri max", d[~plane].max(initial=0), (d[~plane]>0).mean())
