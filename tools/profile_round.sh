#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the bench command, then the PMC passes the
# roofline line needs (each in a run of its own, never mixed with tracing).
#   tools/profile_round.sh <tag>        results under gpurun_out/prof_<tag>/
set -e -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
PMCB="python3 $PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline"

echo "== bench (plain)"
timeout -k 10 300 $BENCH > "$OUT/bench.json"
tail -c 600 "$OUT/bench.json"

echo "== kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
head -12 "$OUT/kernel_stats.csv"

pass() { # name counters...
    local name=$1
    shift
    echo "== pmc $name: $*"
    timeout -s KILL 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o pmc -- $PMCB > "$OUT/pmc_$name.log" 2>&1
    find "$OUT/pmc_$name" -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_$name.csv" \;
    rm -rf "$OUT/pmc_$name"
    python3 tools/pmc_summary.py "$OUT/pmc_$name.csv" k_trace > "$OUT/pmc_$name.txt"
    tail -4 "$OUT/pmc_$name.txt"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY
rm -rf "$OUT/trace"
# HBM-side traffic of the closest-hit kernel per launch (timed frame = the last 8 k_trace dispatches):
# FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
def col(name):
    rows = [l.split() for l in open("%s/pmc_%s.txt" % (out, name)).read().splitlines()[1:]]
    return [float(r[1]) for r in rows][-8:]
f, w = col("fetch"), col("write")
per_launch = [(2.0 * a + b) * 1024.0 for a, b in zip(f, w)]
json.dump({"workload": "dragon1 1920x1080 64spp d8", "hbm_bytes_per_launch": sum(per_launch) / len(per_launch),
           "fetch_size_kib": f, "write_size_kib": w, "note": "per k_trace launch of one frame; FETCH_SIZE x2 (gfx950), + WRITE_SIZE, x1024"},
          open("%s/traffic.json" % out, "w"), indent=1)
print(open("%s/traffic.json" % out).read())
PY
echo done
