#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): PMC passes over tools/refit_bench.py (blas_refit_kernel), one counter group per pass.
# usage: refit_counters.sh [--real]
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/refit_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters
  rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o run --output-format csv -- python3 $REPO/tools/refit_bench.py "${@:3}" > $OUT/$1.log 2> $OUT/$1.err || tail -5 $OUT/$1.err
}
run fetch "FETCH_SIZE" "$@"
run write "WRITE_SIZE" "$@"
run tcc "TCC_HIT_sum TCC_MISS_sum" "$@"
run sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@"
python3 - <<PY
import csv, glob, collections
for g in ("fetch","write","tcc","sq"):
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % g, recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "blas_refit" in k or "skin_kernel" in k:
                per[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in per.items():
            print(g, k, {c: "%.4g" % (sum(v)/len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
find $OUT -name "*.csv" -size +4M -delete
