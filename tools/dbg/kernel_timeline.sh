#!/bin/bash
# Every kernel dispatch of one bench frame with its duration, in launch order (rocprofv3 --kernel-trace).
#   tools/dbg/kernel_timeline.sh [bench args]     -> gpurun_out/timeline.txt
set -e -o pipefail
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tl" -o tl -- python3 $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$OUT/tl.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = None
lines = []
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    lines.append("%10.3f ms  +%7.1f us gap  %9.3f ms  %s" % ((s - t0) / 1e6, gap, (e - s) / 1e6, r["Kernel_Name"][:60]))
    prev_end = e
open(out + "/timeline.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-70:]))
PY
rm -rf "$OUT/tl"
