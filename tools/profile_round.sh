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
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE
rm -rf "$OUT/trace"
# HBM-side traffic of the closest-hit kernel per launch (timed frame = the last 8 k_trace dispatches):
# FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950
python3 - "$OUT" "$TAG" <<'PY'
import csv, hashlib, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
h = hashlib.sha256()
for f in ("rt_hip.hip", "rtx_device.h", "kd_build.h"):  # = bench.py kernel_source_hash(): the build these counters belong to
    h.update(open(os.path.join("raytracer_amd", "csrc", f), "rb").read())
bench = json.loads(open("%s/bench.json" % out).read().strip().splitlines()[-1])
traced = bench["config"]["rays_per_frame"] - bench["config"]["rays_answered_in_shade"]  # rays handed to the closest-hit kernel
def table(name):
    lines = open("%s/pmc_%s.txt" % (out, name)).read().splitlines()
    cols = lines[0].split()[1:]
    rows = [[float(x) for x in l.split()[1:]] for l in lines[1:]][-8:]  # the timed frame's 8 k_trace dispatches
    return {c: [r[i] for r in rows] for i, c in enumerate(cols)}
f, w, tcc, sq = table("fetch")["FETCH_SIZE"], table("write")["WRITE_SIZE"], table("tcc"), table("sq")
per_launch = [(2.0 * a + b) * 1024.0 for a, b in zip(f, w)]
# mean duration of the timed kernel from the kernel-trace run of the same command
ms = None
for r in csv.DictReader(open("%s/kernel_stats.csv" % out)):
    if r["Name"].startswith("void k_trace_flat<false"):
        ms = float(r["AverageNs"]) / 1e6
simd_cycles = [g / 8.0 * 1024.0 for g in sq["GRBM_GUI_ACTIVE"]]  # GRBM_GUI_ACTIVE sums the 8 XCDs; 256 CUs x 4 SIMDs
d = {"workload": "dragon1 1920x1080 64spp d8", "hbm_bytes_per_launch": sum(per_launch) / len(per_launch),
     "fetch_size_kib": f, "write_size_kib": w, "note": "per k_trace launch of one frame; FETCH_SIZE x2 (gfx950), + WRITE_SIZE, x1024",
     "ms_per_launch_kernel_trace": ms,
     "hbm_gbs": (sum(per_launch) / len(per_launch)) / (ms * 1e-3) / 1e9 if ms else None,
     "l2_hit_rate": sum(tcc["TCC_HIT_sum"]) / sum(tcc["TCC_REQ_sum"]),
     "valu_busy": sum(4.0 * x for x in sq["SQ_INSTS_VALU"]) / sum(simd_cycles),
     "valu_busy_per_launch": [4.0 * x / c for x, c in zip(sq["SQ_INSTS_VALU"], simd_cycles)],
     "valu_lane_utilisation": sum(sq["SQ_THREAD_CYCLES_VALU"]) / (64.0 * sum(sq["SQ_ACTIVE_INST_VALU"])),
     # per depth (= per launch of the timed frame): what VERDICT r01 asked to keep
     "per_launch": {"SQ_INSTS_VALU": sq["SQ_INSTS_VALU"], "SQ_THREAD_CYCLES_VALU": sq["SQ_THREAD_CYCLES_VALU"], "SQ_ACTIVE_INST_VALU": sq["SQ_ACTIVE_INST_VALU"],
                    "lane_utilisation": [t / (64.0 * a) for t, a in zip(sq["SQ_THREAD_CYCLES_VALU"], sq["SQ_ACTIVE_INST_VALU"])]},
     "valu_insts_per_frame": sum(sq["SQ_INSTS_VALU"]), "rays_traced_per_frame": traced,
     "valu_insts_per_traced_ray": sum(sq["SQ_INSTS_VALU"]) / traced, "measured_at": tag, "kernel_source_hash": h.hexdigest()[:16]}
json.dump(d, open("%s/traffic.json" % out, "w"), indent=1)
print(json.dumps({k: v for k, v in d.items() if not isinstance(v, list)}, indent=1))
PY
echo done
