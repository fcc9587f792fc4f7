#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round-2 evidence of the refit row -> gpurun_out/refit_profiles_r2/
#   rocprofv3 --kernel-trace --stats of `bench.py --refit` in serial order, two launches and fused, and of the default
#   (overlapped) schedule with two launches; FETCH_SIZE / WRITE_SIZE passes over the fused kernel (tools/fuse_pmc.sh).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/refit_profiles_r2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "serial_two_launches:--refit --no-overlap" "serial_fused:--refit --fuse --no-overlap" "overlap_two_launches:--refit" "overlap_fused:--refit --fuse"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  echo "== $name"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/trace_$name -o run --output-format csv -- python3 $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline $flags > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { echo "$name failed"; tail -n 5 $OUT/bench_$name.err; exit 1; }
  cp $(find $OUT/trace_$name -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$name.csv
  rm -rf $OUT/trace_$name
done
bash $REPO/tools/fuse_pmc.sh > $OUT/fuse_pmc.log 2>&1 && cp $REPO/gpurun_out/fuse_pmc/fused_pmc.json $OUT/fused_pmc.json
ls $OUT
