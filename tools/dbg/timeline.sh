#!/bin/bash
# kernel timeline of the last frame of tools/dbg/share.py
set -e -o pipefail
OUT=$PWD/gpurun_out/timeline; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -o tl -- python3 $PWD/tools/dbg/share.py ${1:-8} > "$OUT/log.txt" 2>&1
find "$OUT/t" -name '*kernel_trace.csv' -exec cp {} "$OUT/kernel_trace.csv" \;
rm -rf "$OUT/t"
tail -3 "$OUT/log.txt"
python3 - "$OUT/kernel_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame: from the last k_raygen pair backwards
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_raygen")]
lanes = 2 if len(idx) >= 6 and int(rows[idx[-1]]["Start_Timestamp"]) - int(rows[idx[-2]]["Start_Timestamp"]) < 5e6 else 1
first = idx[-lanes]
t0 = int(rows[first]["Start_Timestamp"])
qs = {}
for r in rows[first:]:
    q = r.get("Queue_Id", "?")
    qs.setdefault(q, len(qs))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    print("q%d %9.1f -> %9.1f us (%8.1f)  %s grid %s" % (qs[q], s, e, e - s, name, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
PY
