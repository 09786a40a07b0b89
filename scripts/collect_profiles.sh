#!/bin/bash
# Runs ON THE GPU BOX (gpurun): GPU pytest log, bench line, rocprofv3 kernel stats of the same bench command, and the PMC passes
# (one counter per run, --pmc never combined with a trace domain other than --kernel-trace) that profiles/rNN/traffic.json is built
# from.  Output goes to gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/rNN/.
#   gpurun --timeout 1200 -- 'scripts/collect_profiles.sh c_pipeline'
# SKIP_PYTEST=1 skips the test run; COUNTERS="..." overrides the counter list.
set -o pipefail
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
COUNTERS=${COUNTERS:-FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE}
mkdir -p $OUT
cd $ROOT
if [ -z "$SKIP_PYTEST" ]; then
  python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1 || { tail -5 $OUT/pytest_gpu.log; exit 1; }
fi
python bench.py --steps 5 --warmup 1 > $OUT/bench_full.log 2>&1 || { tail -5 $OUT/bench_full.log; exit 1; }
grep '^{"metric"' $OUT/bench_full.log > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/stats_run.log 2>&1 || { tail -5 $OUT/stats_run.log; exit 1; }
for ctr in $COUNTERS; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/pmc_$ctr.log 2>&1 || { tail -5 $OUT/pmc_$ctr.log; exit 1; }
  echo "pmc $ctr done"
done
cd $ROOT
python scripts/pmc_aggregate.py $OUT > $OUT/aggregate.log 2>&1
tail -3 $OUT/aggregate.log
# keep the merge-back small: drop the raw per-dispatch CSVs, keep the summaries
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT/stats -name "*kernel_trace.csv" -delete 2>/dev/null
ls $OUT
