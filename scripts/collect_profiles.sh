#!/bin/bash
# Runs ON THE GPU BOX (gpurun).  Two parts, both into gpurun_out/prof_<tag>/ :
#   headline   rocprofv3 kernel stats of the bench's timed region (the driver's --steps 20 --warmup 5, --no-alone), the PMC passes and the
#              diagnostic build's clock stamps profiles/rNN/traffic.json is built from (scripts/pmc_aggregate.py), then the bench line by the
#              driver's exact command with those counters in place (bench.json = its last stdout line, bench_details.json = the rest);
#   workloads  for every other workload of the bench line (bench.workloads(): north_star, stereo, voc_fft, voc_bank, configs3, voice_*) and
#              for configs1 itself: `rocprofv3 --kernel-trace --stats` and three `--pmc` passes (the six SQ counters fit one pass: 8 SQ slots;
#              FETCH_SIZE and WRITE_SIZE need a pass each: MI355X_MICROARCH.md, PMC slots) of scripts/profile_workload.py <name> 3.
# --pmc is never combined with a trace domain other than --kernel-trace; the program goes directly after `--`.
#   gpurun --timeout 1200 -- 'scripts/collect_profiles.sh r03 headline'      gpurun --timeout 1200 -- 'scripts/collect_profiles.sh r03 workloads voc_fft voc_bank'
# then: python scripts/pmc_aggregate.py gpurun_out/prof_r03 ; python scripts/profile_aggregate.py gpurun_out/prof_r03/workloads profiles/r03
set -o pipefail
TAG=${1:-run}
PART=${2:-headline}
shift 2
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
SQ="SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
mkdir -p $OUT
cd $ROOT
if [ "$PART" = headline ]; then
  # 1. rocprofv3 summary of the headline's own timed region: the driver's step counts, nothing but pipeline launches (--no-alone)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-alone --details $OUT/stats_details.json > $OUT/stats_run.log 2>&1 || { tail -5 $OUT/stats_run.log; exit 1; }
  # 2. one counter per pass over one warm-up and one timed step
  for ctr in FETCH_SIZE WRITE_SIZE $SQ; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-alone --details $OUT/pmc_details.json > $OUT/pmc_$ctr.log 2>&1 || { tail -5 $OUT/pmc_$ctr.log; exit 1; }
    echo "pmc $ctr done"
  done
  cd $ROOT
  # 3. the clock the chip holds inside the two throughput kernels: the diagnostic build's stamps (make -C cpp_audio_amd/csrc clk)
  if [ -f $ROOT/cpp_audio_amd/libresynth_amd_clk.so ]; then
    RESYNTH_AMD_LIB=$ROOT/cpp_audio_amd/libresynth_amd_clk.so python bench.py --steps 20 --warmup 5 --clock-stamps --no-cpu-baseline --no-other-configs --no-alone --details $OUT/clock_details.json > $OUT/clock_run.log 2>&1 || { tail -5 $OUT/clock_run.log; exit 1; }
  fi
  python scripts/pmc_aggregate.py $OUT > $OUT/aggregate.log 2>&1
  tail -3 $OUT/aggregate.log
  # 4. the bench line itself, by the driver's exact command, with the fresh counters in place (bench.py reads profiles/r*/traffic.json by source hash)
  if [ -z "$SKIP_BENCH" ]; then
    mkdir -p $ROOT/profiles/$TAG && cp $OUT/traffic.json $ROOT/profiles/$TAG/traffic.json
    python bench.py --gpus 1 --steps 20 --warmup 5 --details $OUT/bench_details.json > $OUT/bench_full.log 2>&1 || { tail -5 $OUT/bench_full.log; exit 1; }
    tail -1 $OUT/bench_full.log > $OUT/bench.json
    wc -c $OUT/bench.json
  fi
else
  NAMES=${@:-configs1 north_star stereo voc_fft voc_bank configs3 voice_sweep voice_birds voice_robots voice_wind}
  cd /tmp && export TMPDIR=/tmp
  for w in $NAMES; do
    W=$OUT/workloads/$w
    mkdir -p $W
    rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -- python3 $ROOT/scripts/profile_workload.py $w 3 > $W/stats_run.log 2>&1 || { tail -5 $W/stats_run.log; exit 1; }
    rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $W/pmc_SQ -- python3 $ROOT/scripts/profile_workload.py $w 3 > $W/pmc_SQ.log 2>&1 || { tail -5 $W/pmc_SQ.log; exit 1; }
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $W/pmc_FETCH -- python3 $ROOT/scripts/profile_workload.py $w 3 > $W/pmc_FETCH.log 2>&1 || { tail -5 $W/pmc_FETCH.log; exit 1; }
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $W/pmc_WRITE -- python3 $ROOT/scripts/profile_workload.py $w 3 > $W/pmc_WRITE.log 2>&1 || { tail -5 $W/pmc_WRITE.log; exit 1; }
    grep '^{"workload"' $W/stats_run.log | tail -1 | cut -c1-200
  done
  cd $ROOT
fi
# keep the merge-back small: drop the raw per-dispatch CSVs that are not needed, keep the summaries
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
find $OUT -name "*counter_collection.csv" -size +12M -delete
du -sh $OUT
