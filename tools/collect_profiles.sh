#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats of the default bench.py run, then two separate
# PMC passes (FETCH_SIZE, WRITE_SIZE) over the LBS-only workload.  Summaries land under gpurun_out/profiles_new/.
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run --output-format csv -- python3 $REPO/bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
echo "== pmc FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload lbs > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "== pmc WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload lbs > $OUT/pmc_write.json 2> $OUT/pmc_write.err
python3 $REPO/tools/summarize_profiles.py $OUT
# keep only the summaries (the raw traces are large)
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
ls -la $OUT
