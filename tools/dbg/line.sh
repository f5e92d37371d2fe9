#!/bin/bash
set -e -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/line; mkdir -p $OUT
for v in 0 1 2; do
  timeout -s KILL 60 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_MISS_sum --output-format csv -d $OUT/v$v -o pmc -- $PWD/tools/dbg/line $v > $OUT/v$v.log 2>&1
  find $OUT/v$v -name '*counter_collection.csv' -exec cp {} $OUT/v$v.csv \;
  rm -rf $OUT/v$v
  echo "variant $v"; python3 tools/pmc_summary.py $OUT/v$v.csv "k("
done
