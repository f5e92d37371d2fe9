#!/bin/bash
for d in "$@"; do
  echo "sort_depth $d: $(RTX_SORT_DEPTH=$d timeout -k 10 100 python3 tools/prof_trace.py dragon1 1920 1080 64 8 | python3 -c 'import json,sys; d=json.load(sys.stdin); print("total", round(d["ms_total"],1), "trace", round(d["ms_trace"],1), "sort", round(d["ms_sort"],1), [round(x,1) for x in d["ms"]])')" || exit 1
done
