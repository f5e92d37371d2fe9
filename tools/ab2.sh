#!/bin/bash
# Two builds of librt_hip.so timed on the SAME box, alternating (box-to-box variance is of the order of the effects).
#   tools/ab2.sh save <name>        build the tree as it is and keep it as raytracer_amd/lib/ab_<name>.so
#   tools/ab2.sh run <n> <a> <b> ...  bench a, b, ..., a, b, ... (n rounds) in one gpurun call
set -e
cd "$(dirname "$0")/.."
case "$1" in
save)
    python -m raytracer_amd.build >/dev/null
    cp raytracer_amd/lib/librt_hip.so raytracer_amd/lib/ab_$2.so
    ;;
run)
    N=$2; shift 2; V="$*"
    /usr/local/graft/bin/gpurun --timeout 900 -- "for i in \$(seq $N); do for v in $V; do RTX_HIP_LIB=\$PWD/raytracer_amd/lib/ab_\$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/ab2_\$v.json 2>/dev/null || exit 1; python -c \"
import json; d=json.load(open('gpurun_out/ab2_\$v.json')); print('\$v', round(d['ms_per_step'],2), round(d['config']['kernel_ms_per_frame']['trace'],2))\"; done; done" 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)\|amdgpu.ids"
    ;;
esac
