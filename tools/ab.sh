#!/bin/bash
# A/B loop for kernel work: build, then on the GPU box the closest-hit parity subset and the bench frame (no CPU baseline).
#   tools/ab.sh <tag> [extra bench args]      -> gpurun_out/ab_<tag>.json, gpurun_out/ab_<tag>.log
set -e
cd "$(dirname "$0")/.."
TAG=${1:-x}; shift || true
python -m raytracer_amd.build >/dev/null
/usr/local/graft/bin/gpurun --timeout 600 -- "timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread -k 'primary_hits or checksums or culled or awkward or literal or tail_kernel or incoherent or visit or answered' > gpurun_out/ab_$TAG.log 2>&1 && timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 $* > gpurun_out/ab_$TAG.json 2>> gpurun_out/ab_$TAG.log; tail -2 gpurun_out/ab_$TAG.log" 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)" | tail -4
python - "$TAG" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
k = d["config"]["kernel_ms_per_frame"]
print("%s: %.2f ms/frame  %.0f Mrays/s  trace %.2f shade %.2f bin %.2f  frac %.3f" % (sys.argv[1], d["ms_per_step"], d["value"], k["trace"], k["shade"], k["ray_binning"], d["roofline"]["frac"]))
PY
