#!/bin/bash
# sweep RTX_TRACE_TUNE (leaf_min,inner_min,refill_min)
for t in "$@"; do
  echo "tune $t: $(RTX_TRACE_TUNE=$t timeout -k 10 100 python3 tools/prof_trace.py dragon1 1920 1080 64 8 | python3 -c 'import json,sys; d=json.load(sys.stdin); print(round(d["ms_trace"],1), [round(x,1) for x in d["ms"]])')" || exit 1
done
