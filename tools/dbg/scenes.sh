#!/bin/bash
for s in teapot bunny pear materials; do
  for v in ${VARIANTS:-2 5 3 4}; do
    echo "$s variant $v: $(RTX_TRACE_VARIANT=$v timeout -k 10 200 python3 tools/prof_trace.py $s 1920 1080 ${SPP:-16} 8 | python3 -c 'import json,sys; d=json.load(sys.stdin); print("trace", round(d["ms_trace"],1), "shade", round(d["ms_shade"],1), "total", round(d["ms_total"],1), "Mrays/s", round(d["total_mrays_s"]), [round(x,1) for x in d["ms"]])')" || exit 1
  done
done
