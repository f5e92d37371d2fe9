#!/bin/bash
# One PMC pass over the bench frame, summarised for the kernels whose name contains <pattern>.
#   tools/dbg/pmc_kernel.sh <pattern> <tag> COUNTER...      -> gpurun_out/pmc_<tag>.txt
set -e -o pipefail
PAT=$1; TAG=$2; shift 2
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pk_$TAG" -o pmc -- python3 $PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pk_$TAG.log" 2>&1
find "$OUT/pk_$TAG" -name '*counter_collection.csv' -exec cp {} "$OUT/pk_$TAG.csv" \;
rm -rf "$OUT/pk_$TAG"
python3 tools/pmc_summary.py "$OUT/pk_$TAG.csv" "$PAT" > "$OUT/pmc_$TAG.txt"
rm -f "$OUT/pk_$TAG.csv"
tail -9 "$OUT/pmc_$TAG.txt"
