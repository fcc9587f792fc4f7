#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the refit row's evidence. rocprofv3 kernel-trace stats of `bench.py --refit`, then
# FETCH_SIZE / WRITE_SIZE passes over tools/refit_bench.py (synthetic and real mesh).  Summaries -> gpurun_out/refit_profiles/.
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/refit_profiles
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run --output-format csv -- python3 $REPO/bench.py --refit --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_refit.json 2> $OUT/bench_refit.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for mesh in syn real; do
  flag=""; [ $mesh = real ] && flag="--real"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${mesh}_$c -o run --output-format csv -- python3 $REPO/tools/refit_bench.py $flag > $OUT/pmc_${mesh}_$c.log 2> $OUT/pmc_${mesh}_$c.err
  done
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for mesh in ("syn", "real"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for path in glob.glob("$OUT/pmc_%s_%s/**/*counter_collection.csv" % (mesh, c), recursive=True):
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "blas_refit" in r["Kernel_Name"] and r["Counter_Name"] == c]
            out.setdefault(mesh, {})[c + "_KiB_mean"] = sum(vals) / len(vals)
            out[mesh][c + "_launches"] = len(vals)
    log = open("$OUT/pmc_%s_FETCH_SIZE.log" % mesh).read()
    out[mesh]["refit_bench"] = [l for l in log.splitlines() if l.startswith(("mesh", "refit", "skin"))]
json.dump(out, open("$OUT/refit_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; rm -rf $OUT/trace
cat $OUT/bench_refit.json | cut -c1-400
