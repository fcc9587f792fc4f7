#!/bin/bash
# GPU box: rocprofv3 kernel stats of tools/separation_bench.py (one crowd size), top kernels printed
n=${1:-8192}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_sep_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o sep --output-format csv -- python3 $REPO/tools/separation_bench.py $n > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-64s calls %6s total_ms %9.2f avg_us %9.1f" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
grep "^footprint" $OUT/run.log | cut -c1-120
