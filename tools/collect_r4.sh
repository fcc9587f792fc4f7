#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round-4 evidence set -> gpurun_out/profiles_r4/
#   1. the driver's command (python bench.py --steps 20 --warmup 5) as it stands, with the per-launch HIP-event times of its skin launches
#   2. rocprofv3 --kernel-trace --stats of `bench.py --steps 200 --warmup 20 --no-extras` (overlap on, the default schedule) and of
#      --no-overlap. --no-extras: the default line's `world_sync` and `real_mesh` legs launch the same LBS kernel at other sizes and
#      under other schedules, which would enter rocprofv3's per-kernel average; the timed region and its roofline are the same.
#   3. --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) over the LBS kernel of the default schedule -> lbs_traffic.json
#   4. bench lines: --host-sync world, --workload agents, --workload mixed, --workload lbs, --refit, --mesh ybot
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_r4
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/bench.py --steps 20 --warmup 5 --dump-lbs-events $OUT/lbs_event_times_driver_style.txt > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err
python3 $REPO/bench.py --steps 200 --warmup 20 --dump-lbs-events $OUT/lbs_event_times_200_steps.txt > $OUT/bench_default_200.json 2> $OUT/bench_default_200.err
for mode in default serial; do
  flags=""; [ $mode = serial ] && flags="--no-overlap"
  echo "== kernel trace ($mode)"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/trace_$mode -o run --output-format csv -- python3 $REPO/bench.py --steps 200 --warmup 20 --no-extras $flags > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err || { echo "trace $mode failed"; tail -5 $OUT/bench_$mode.err; }
  cp $(find $OUT/trace_$mode -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$mode.csv
  [ $mode = default ] && python3 $REPO/tools/overlap_timeline.py $(find $OUT/trace_$mode -name "*kernel_trace.csv" | head -1) > $OUT/overlap_timeline.txt
  find $OUT/trace_$mode -name "*_kernel_trace.csv" -delete
done
echo "== pmc FETCH_SIZE / WRITE_SIZE (LBS kernel of the default schedule)"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --workload lbs > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --workload lbs > $OUT/pmc_write.json 2> $OUT/pmc_write.err
mkdir -p $OUT/trace; cp $OUT/kernel_stats_default.csv $OUT/trace/run_kernel_stats.csv; cp $OUT/bench_default.json $OUT/bench.json
python3 $REPO/tools/summarize_profiles.py $OUT > $OUT/summary.log 2>&1; tail -5 $OUT/summary.log
find $OUT -name "*counter_collection.csv" -size +2M -delete; find $OUT -name "*_kernel_trace.csv" -delete
echo "== other bench lines"
cd $REPO
python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --host-sync world > $OUT/bench_host_sync_world.json 2> $OUT/bench_host_sync_world.err
python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --workload agents > $OUT/bench_agents_1gpu.json 2> $OUT/bench_agents.err
python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --workload mixed --mesh ybot --chars 31250 > $OUT/bench_shard_configs3_31250_ybot_mixed.json 2> $OUT/bench_mixed.err
python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --workload lbs > $OUT/bench_lbs.json 2> $OUT/bench_lbs.err
python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --refit > $OUT/bench_refit_two_launches.json 2> $OUT/bench_refit.err
python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-overlap --refit --fuse > $OUT/bench_refit_serial_fused.json 2> $OUT/bench_refit_fused.err
timeout -k 10 300 python3 bench.py --gpus 2 --single-device --dist-backend gloo --workload agents --chars 5000 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_2ranks_one_gpu_gloo_self_launched.json 2> $OUT/bench_2ranks.err
timeout -k 10 200 python3 tools/ray_bench.py > $OUT/ray_bench_10k.txt 2>&1
timeout -k 10 300 python3 tools/ray_bench.py --chars 31250 > $OUT/ray_bench_31250_grid_ordered_instance_level.txt 2>&1
SGE_BLAS_FLAT_INSTANCES=1 timeout -k 10 300 python3 tools/ray_bench.py --chars 31250 > $OUT/ray_bench_31250_flat_instance_level.txt 2>&1
ls $OUT
