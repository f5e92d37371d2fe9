#!/bin/bash
# Kernel dispatches of ONE rank's share of the bench frame (band split, rank r of n), in launch order with gaps.
#   tools/dbg/share_timeline.sh <rank> <world>     -> gpurun_out/share_timeline.txt
set -e -o pipefail
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
cat > /tmp/share_frame.py <<PY
import sys, time, torch
sys.path.insert(0, "$PWD")
from raytracer_amd import hip, scenes
sc = scenes.build("dragon1"); ds = hip.DeviceScene(sc.desc, 0); cam = sc.rtx_camera()
W, H = 1920, 1080
acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
p = hip.make_params(W, H, 64, 8, band=(8, $1, $2))
for i in range(4):
    acc.zero_(); torch.cuda.synchronize(); t0 = time.time()
    ds.render_device(cam, p, acc.data_ptr()); torch.cuda.synchronize()
    print("frame %d: %.2f ms" % (i, (time.time() - t0) * 1e3), flush=True)
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/stl" -o tl -- python3 /tmp/share_frame.py > "$OUT/stl.log" 2>&1
cat "$OUT/stl.log" | grep frame
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/stl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last frame = from the last k_raygen on
last = max(i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"])
rows = rows[last:]
t0 = int(rows[0]["Start_Timestamp"]); prev = None; lines = []; busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    busy += e - s
    lines.append("%9.3f ms  +%7.1f us gap  %8.3f ms  %s" % ((s - t0) / 1e6, gap, (e - s) / 1e6, r["Kernel_Name"][:70]))
    prev = max(prev or e, e)
lines.append("span %.3f ms, sum of kernel durations %.3f ms" % ((prev - t0) / 1e6, busy / 1e6))
open(out + "/share_timeline.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf "$OUT/stl"
