#!/bin/bash
# kernel trace of the device kd build
set -e -o pipefail
OUT=$PWD/gpurun_out/kdprof; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o kd -- python3 $PWD/tools/dbg/kdtime.py ${1:-dragon1} > "$OUT/log.txt" 2>&1
find "$OUT/t" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/t" -name '*kernel_trace.csv' -exec cp {} "$OUT/kernel_trace.csv" \;
rm -rf "$OUT/t"
tail -2 "$OUT/log.txt"
head -6 "$OUT/kernel_stats.csv"
python3 - "$OUT/kernel_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [r for r in rows if r["Kernel_Name"].startswith("k_kd_")]
# the last build: find the last run of k_kd kernels (a build = decide, scan, emit per level)
n = len(ks) // 2
last = ks[-n:]
lvl = 0
for i in range(0, len(last), 3):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last[i:i + 3]]
    print("level %2d grid %8s: decide %8.1f us scan %7.1f us emit %8.1f us" % (lvl, last[i]["Grid_Size_X"] if "Grid_Size_X" in last[i] else last[i].get("Grid_Size", "?"), *d))
    lvl += 1
PY
