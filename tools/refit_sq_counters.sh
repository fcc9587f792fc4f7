#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): SQ / LDS counter passes over tools/refit_bench.py (the blas_refit kernels), one group per
# pass, printed as per-launch means. usage: refit_sq_counters.sh [refit_bench.py arguments]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/refit_sq_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
k=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_FLAT"; do
  k=$((k+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$k -o run --output-format csv -- python3 $REPO/tools/refit_bench.py "$@" > $OUT/g$k.log 2> $OUT/g$k.err || tail -3 $OUT/g$k.err
done
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(list)
for path in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "blas_refit" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(per.items()):
    print(k, "%.5g" % (sum(v) / len(v)), "launches", len(v))
PY
find $OUT -name "*.csv" -size +2M -delete
