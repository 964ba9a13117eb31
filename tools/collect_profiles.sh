#!/bin/bash
# Run on the GPU box from the repo root (gpurun -- 'bash tools/collect_profiles.sh r01'): collects the rocprofv3
# summaries kept under profiles/ (DESIGN.md §6).  Every counter pass is its own run; traces and counters are never mixed.
TAG=${1:-r01}
OUT=gpurun_out/prof
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p $OUT
BENCH="python3 bench.py --no-cpu-baseline"
cd /tmp && cd $ROOT

rocprofv3 --kernel-trace -d $OUT/trace -- $BENCH --steps 12 --warmup 3 > $OUT/trace.log 2>&1 || exit 1
DB=$(find $OUT/trace -name "*.db" | head -1)
python3 tools/rocprof_summary.py stats $DB > $OUT/${TAG}_kernel_stats.csv
python3 tools/rocprof_summary.py gaps $DB > $OUT/${TAG}_step_gaps.csv
python3 tools/rocprof_summary.py timeline $DB 8 > $OUT/${TAG}_step_timeline.csv
echo "trace done"

rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH --steps 3 --warmup 1 > $OUT/fetch.log 2>&1 || exit 2
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH --steps 3 --warmup 1 > $OUT/write.log 2>&1 || exit 3
python3 tools/rocprof_summary.py hbm-csv $(find $OUT/fetch -name "*counter_collection.csv" | head -1) \
    $(find $OUT/write -name "*counter_collection.csv" | head -1) 4096 > $OUT/${TAG}_pmc_hbm.csv
echo "hbm done"

rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
    SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $OUT/sq -- $BENCH --steps 3 --warmup 1 \
    > $OUT/sq.log 2>&1 || exit 4
python3 tools/rocprof_summary.py pmc-csv $(find $OUT/sq -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_pmc_sq.csv
echo "sq done"
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/sq
