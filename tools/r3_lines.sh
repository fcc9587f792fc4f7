#!/bin/bash
# Runs ON THE GPU BOX: the other single-GPU bench lines of the round-3 build + the kernel timeline of the default step -> gpurun_out/profiles_r3/
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_r3
mkdir -p $OUT
cd $REPO
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err
python3 bench.py > $OUT/bench_default_untraced.json 2> $OUT/bench_default_untraced.err   # the default line (200 steps) without the profiler
bash tools/r2_trace.sh profiles_r3/trace_timeline > /dev/null 2>&1
python3 tools/overlap_timeline.py $(find $OUT/trace_timeline/trace -name "*kernel_trace.csv" | head -1) > $OUT/overlap_timeline.txt 2>&1
rm -rf $OUT/trace_timeline/trace
B="python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline"
$B --workload lbs > $OUT/bench_lbs.json 2>/dev/null
$B --workload mixed > $OUT/bench_merged.json 2>/dev/null
$B --mesh ybot > $OUT/bench_ybot_cheese.json 2>/dev/null
$B --scene synthetic > $OUT/bench_synthetic_scene.json 2>/dev/null
$B --workload agents > $OUT/bench_agents_1gpu.json 2>/dev/null
$B --mesh ybot --workload mixed --chars 31250 > $OUT/bench_shard_configs3_31250_ybot_mixed.json 2>/dev/null
$B --workload agents --chars 31250 > $OUT/bench_shard_configs4_31250_agents.json 2>/dev/null
python3 bench.py --chars 250000 --workload mixed --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_250k_single_gpu.json 2>/dev/null
for f in $OUT/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('%-50s %.4f ms  %.3f M chars/s  whole %.3f  lbs %.3f (%.3f)  move %.3f pose %.3f' % ('$(basename $f)', d['ms_per_step'], d['value']/1e6, d['whole_path_hbm_frac'], d['kernels_ms_per_step']['lbs'], d['roofline']['frac'], d['kernels_ms_per_step']['move_ccd'], d['kernels_ms_per_step']['pose']))
"; done
cat $OUT/overlap_timeline.txt | tail -9
