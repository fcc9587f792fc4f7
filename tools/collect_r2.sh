#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round-2 evidence set -> gpurun_out/profiles_r2/
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench.py line (overlap on) and of --no-overlap
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs) over the LBS kernel (bench.py --workload lbs --no-overlap)
#   3. SQ counter passes over the move-stage kernels (tools/move_pmc.sh, serial order so that counters belong to one kernel)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in default serial; do
  flags=""; [ $mode = serial ] && flags="--no-overlap"
  echo "== kernel trace ($mode)"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/trace_$mode -o run --output-format csv -- python3 $REPO/bench.py --steps 200 --warmup 20 $flags > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err || { echo "trace $mode failed"; tail -5 $OUT/bench_$mode.err; }
  cp $(find $OUT/trace_$mode -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$mode.csv
  [ $mode = default ] && python3 $REPO/tools/overlap_timeline.py $(find $OUT/trace_$mode -name "*kernel_trace.csv" | head -1) > $OUT/overlap_timeline.txt
  find $OUT/trace_$mode -name "*_kernel_trace.csv" -delete
done
echo "== pmc FETCH_SIZE / WRITE_SIZE (LBS)"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload lbs --no-overlap > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload lbs --no-overlap > $OUT/pmc_write.json 2> $OUT/pmc_write.err
mkdir -p $OUT/trace; cp $OUT/kernel_stats_default.csv $OUT/trace/run_kernel_stats.csv; cp $OUT/bench_default.json $OUT/bench.json
python3 $REPO/tools/summarize_profiles.py $OUT > $OUT/summary.log 2>&1; tail -5 $OUT/summary.log
find $OUT -name "*counter_collection.csv" -size +2M -delete; find $OUT -name "*_kernel_trace.csv" -delete
echo "== SQ counters on the move stage"
bash $REPO/tools/move_pmc.sh profiles_r2/move_pmc_cheese --no-overlap > $OUT/move_pmc_cheese.log 2>&1
bash $REPO/tools/move_pmc.sh profiles_r2/move_pmc_synth --no-overlap --scene synthetic > $OUT/move_pmc_synth.log 2>&1
ls $OUT
