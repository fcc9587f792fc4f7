#!/bin/bash
# Runs ON THE GPU BOX: the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) over the LBS kernel of the default schedule
# -> gpurun_out/profiles_r3/lbs_traffic.json (the record bench.py prints as roofline.traffic; copy it to profiles/lbs_traffic.json)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_r3
mkdir -p $OUT; rm -rf $OUT/pmc_fetch $OUT/pmc_write
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload lbs > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload lbs > $OUT/pmc_write.json 2> $OUT/pmc_write.err
[ -f $OUT/bench.json ] || cp $OUT/pmc_write.json $OUT/bench.json
python3 $REPO/tools/summarize_profiles.py $OUT > $OUT/summary.log 2>&1
find $OUT -name "*counter_collection.csv" -size +2M -delete; find $OUT -name "*_kernel_trace.csv" -delete
cat $OUT/lbs_traffic.json
