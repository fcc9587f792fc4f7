#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel-trace stats of a bench.py run. Usage: tools/r2_trace.sh <tag> [bench flags...]
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/trace -o run --output-format csv -- python3 $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
find $OUT/trace -name "*_kernel_trace.csv" -size +20M -delete
cut -c1-200 $OUT/kernel_stats.csv | head -20
