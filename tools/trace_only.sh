#!/bin/bash
# Kernel-trace statistics of one bench run (and, with a second argument, the timeline of one step):
#   gpurun -- 'bash tools/trace_only.sh <outname> [timeline-step] [bench args...]'   -> gpurun_out/<outname>.csv (+ _timeline.csv)
NAME=$1; STEP=$2; shift; shift
OUT=gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace_tmp -- python3 bench.py --no-cpu-baseline --steps 12 --warmup 3 "$@" > $OUT/trace_tmp.log 2>&1 || exit 1
DB=$(find $OUT/trace_tmp -name "*.db" | head -1)
python3 tools/rocprof_summary.py stats $DB > gpurun_out/$NAME.csv
if [ -n "$STEP" ]; then python3 tools/rocprof_summary.py timeline $DB $STEP > gpurun_out/${NAME}_timeline.csv; fi
rm -rf $OUT/trace_tmp
