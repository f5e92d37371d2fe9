#!/bin/bash
# env sweeps of the closest-hit kernel on the bench frame: tools/dbg/sweep.sh  (run on the GPU box)
run() { echo -n "$* :: "; env "$@" timeout -k 10 120 python tools/dbg/depthms.py dragon1 2>&1 | grep -v amdgpu.ids | cut -c1-120; }
run A=1
for t in 4,8,8 4,8,24 4,8,32 2,8,16 8,8,16 4,4,16 4,16,16 4,24,16 8,16,24; do run RTX_TRACE_TUNE=$t; done
for c in 32 48 96 128; do run RTX_TRACE_CHUNK=$c; done
for b in 5 4; do run RTX_TRACE_BLOCKS_PER_CU=$b; done
