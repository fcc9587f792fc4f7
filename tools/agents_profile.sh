#!/bin/bash
# GPU box: rocprofv3 kernel stats of `bench.py --workload agents` (one GPU's share of configs[4] at 10k characters)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_agents_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o ag --output-format csv -- python3 $REPO/bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline --workload agents "$@" > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-70s calls %6s total_ms %9.2f avg_us %9.1f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], d.get("kernels_ms_per_step"))
PY
