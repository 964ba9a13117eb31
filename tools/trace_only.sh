#!/bin/bash
# usage: trace_only.sh <outname>  -> gpurun_out/<outname>.csv
OUT=gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace_tmp -- python3 bench.py --no-cpu-baseline --steps 12 --warmup 3 > $OUT/trace_tmp.log 2>&1 || exit 1
DB=$(find $OUT/trace_tmp -name "*.db" | head -1)
python3 tools/rocprof_summary.py stats $DB > gpurun_out/$1.csv
rm -rf $OUT/trace_tmp
