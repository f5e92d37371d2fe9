#!/bin/bash
# One build, one box: bench.py under several settings of one tuning knob (the environment sets a knob's default), alternating.
#   tools/knob_sweep.sh <rounds> <KNOB> <v1> <v2> ...     e.g. tools/knob_sweep.sh 2 LONG_AFTER -1 32 64 128
set -e
cd "$(dirname "$0")/.."
N=$1; K=$2; shift 2; V="$*"
EXTRA=${BENCH_ARGS:-}
tools/gpu.sh "for i in \$(seq $N); do for v in $V; do RTX_$K=\$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 2 $EXTRA > gpurun_out/sweep_\$v.json 2>/dev/null || exit 1; python -c \"
import json; d=json.load(open('gpurun_out/sweep_\$v.json')); k=d['config']['kernel_ms_per_frame']; print('$K=\$v', 'frame %.2f ms' % d['ms_per_step'], 'trace %.2f' % k['trace'], 'shade %.2f' % k['shade'])\"; done; done" 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\|every\)\|amdgpu.ids"
