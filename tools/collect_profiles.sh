#!/bin/bash
# Run on the GPU box from the repo root:
#   gpurun -- 'bash tools/collect_profiles.sh r02 n12 4096'        (tag, bench workload, per-GPU batch)
# Collects the rocprofv3 summaries kept under profiles/  into gpurun_out/prof/: a bench line, kernel-trace
# stats / step gaps, HBM traffic (FETCH_SIZE and WRITE_SIZE in separate passes) and SQ counters.  Every counter pass is its
# own run; traces and counters are never mixed.  "full" as 4th argument adds the step timeline.
TAG=${1:-r02}
WL=${2:-n12}
BATCH=${3:-4096}
SFX=${WL}_b${BATCH}
OUT=gpurun_out/prof
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="--workload $WL --batch $BATCH"
cd /tmp && cd $ROOT

python3 bench.py $ARGS --steps 50 --warmup 10 > $OUT/${TAG}_bench_${SFX}.json 2> $OUT/bench_${SFX}.err || exit 1
echo "bench done"

# the same command as the bench line above (50 timed steps behind 10 warm-up steps); the statistics cover the timed steps only
# (the last 5/6 of every kernel's launches), like bench.py's own HIP-event average
rocprofv3 --kernel-trace -d $OUT/trace_$SFX -- python3 bench.py --no-cpu-baseline $ARGS --steps 50 --warmup 10 > $OUT/trace_$SFX.log 2>&1 || exit 1
DB=$(find $OUT/trace_$SFX -name "*.db" | head -1)
python3 tools/rocprof_summary.py stats $DB 0.18 > $OUT/${TAG}_kernel_stats_${SFX}.csv
python3 tools/rocprof_summary.py gaps $DB > $OUT/${TAG}_step_gaps_${SFX}.csv
if [ "$4" == "full" ]; then python3 tools/rocprof_summary.py timeline $DB 8 > $OUT/${TAG}_step_timeline_${SFX}.csv; fi
echo "trace done"

rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$SFX -- python3 bench.py --no-cpu-baseline $ARGS --steps 3 --warmup 1 > $OUT/fetch_$SFX.log 2>&1 || exit 2
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$SFX -- python3 bench.py --no-cpu-baseline $ARGS --steps 3 --warmup 1 > $OUT/write_$SFX.log 2>&1 || exit 3
python3 tools/rocprof_summary.py hbm-csv $(find $OUT/fetch_$SFX -name "*counter_collection.csv" | head -1) \
    $(find $OUT/write_$SFX -name "*counter_collection.csv" | head -1) $BATCH > $OUT/${TAG}_pmc_hbm_${SFX}.csv
echo "hbm done"

rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
    SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $OUT/sq_$SFX -- python3 bench.py --no-cpu-baseline $ARGS --steps 3 --warmup 1 \
    > $OUT/sq_$SFX.log 2>&1 || exit 4
python3 tools/rocprof_summary.py pmc-csv $(find $OUT/sq_$SFX -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_pmc_sq_${SFX}.csv
python3 tools/rocprof_summary.py sq-derived $OUT/${TAG}_pmc_sq_${SFX}.csv > $OUT/${TAG}_pmc_sqd_${SFX}.csv
echo "sq done"
rm -rf $OUT/trace_$SFX $OUT/fetch_$SFX $OUT/write_$SFX $OUT/sq_$SFX
