#!/bin/bash
# one rocprofv3 --pmc pass over one frame of the profile helper:  tools/pmc_pass.sh <name> <counters...>
set -e -o pipefail
NAME=$1; shift
OUT=$PWD/gpurun_out/pmc; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$NAME" -o pmc -- python3 $PWD/tools/prof_trace.py ${SCENE:-dragon1} 1920 1080 ${SPP:-64} 8 --frames 1 > "$OUT/$NAME.log" 2>&1
find "$OUT/$NAME" -name '*counter_collection.csv' -exec cp {} "$OUT/$NAME.csv" \;
rm -rf "$OUT/$NAME"
python3 tools/pmc_summary.py "$OUT/$NAME.csv" ${KPAT:-k_trace} | tee "$OUT/$NAME.txt"
