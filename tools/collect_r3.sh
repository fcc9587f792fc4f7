#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round-3 evidence set -> gpurun_out/profiles_r3/
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench.py line (overlap on: resident four-character LBS) and of --no-overlap,
#      + the kernel timeline of the default step
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs) over the LBS kernel of the default schedule (bench.py --workload lbs)
#   3. in-kernel cycle stamps of the collision + pose kernels in three settings (tools/wave_prof.py)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_r3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err
for mode in default serial; do
  flags=""; [ $mode = serial ] && flags="--no-overlap"
  echo "== kernel trace ($mode)"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/trace_$mode -o run --output-format csv -- python3 $REPO/bench.py --steps 200 --warmup 20 $flags > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err || { echo "trace $mode failed"; tail -5 $OUT/bench_$mode.err; }
  cp $(find $OUT/trace_$mode -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$mode.csv
  [ $mode = default ] && python3 $REPO/tools/overlap_timeline.py $(find $OUT/trace_$mode -name "*kernel_trace.csv" | head -1) > $OUT/overlap_timeline.txt
  [ $mode = default ] && python3 $REPO/tools/step_gaps.py $(find $OUT/trace_$mode -name "*kernel_trace.csv" | head -1) > $OUT/step_chain.txt
  find $OUT/trace_$mode -name "*_kernel_trace.csv" -delete
done
echo "== pmc FETCH_SIZE / WRITE_SIZE (LBS kernel of the default schedule)"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload lbs > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload lbs > $OUT/pmc_write.json 2> $OUT/pmc_write.err
mkdir -p $OUT/trace; cp $OUT/kernel_stats_default.csv $OUT/trace/run_kernel_stats.csv; cp $OUT/bench_default.json $OUT/bench.json
python3 $REPO/tools/summarize_profiles.py $OUT > $OUT/summary.log 2>&1; tail -5 $OUT/summary.log
find $OUT -name "*counter_collection.csv" -size +2M -delete; find $OUT -name "*_kernel_trace.csv" -delete
echo "== wave stamps"
cd $REPO
for m in alone overlap; do python3 tools/wave_prof.py cheese $m > $OUT/wave_prof_$m.txt 2>&1; done
SGE_SKIN_PERSISTENT=0 python3 tools/wave_prof.py cheese overlap > $OUT/wave_prof_handover.txt 2>&1
ls $OUT
echo "== SQ counters on the move stage and the pose kernel (serial order so that counters belong to one kernel)"
bash $REPO/tools/move_pmc.sh profiles_r3/move_pmc_cheese --no-overlap > $OUT/move_pmc_cheese.log 2>&1
ls $OUT/move_pmc_cheese
