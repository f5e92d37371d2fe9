#!/bin/bash
for b in "$@"; do
  echo "blocks/CU $b: $(RTX_TRACE_BLOCKS_PER_CU=$b timeout -k 10 100 python3 tools/prof_trace.py dragon1 1920 1080 64 8 | python3 -c 'import json,sys; d=json.load(sys.stdin); print(round(d["ms_trace"],1), [round(x,1) for x in d["ms"]])')" || exit 1
done
