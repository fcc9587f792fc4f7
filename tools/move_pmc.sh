#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes over the move stage (move_kernel<0>, <1>, <1,HEAVY>) of bench.py.
# Usage: tools/move_pmc.sh <tag> [bench flags]   -> gpurun_out/<tag>/pmc_summary.json
# Each pass is its own run with --kernel-trace only (8 SQ slots per pass); the program comes right after `--`.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CU_CYCLES"
P4="GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS_ATOMIC"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  echo "== pass $i: $P"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P -d $OUT/p$i -o run --output-format csv -- python3 $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline "$@" > $OUT/p$i.json 2> $OUT/p$i.err || { echo "pass $i failed"; tail -5 $OUT/p$i.err; }
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
res = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "move_" not in k and "pose_kernel" not in k and "skin_" not in k and "order_" not in k and "classify" not in k:
            continue
        res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in res.items()}
json.dump(summ, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for k, cs in summ.items():
    print(k[:70])
    print("   ", {c: round(v["mean"]) for c, v in cs.items()})
PY
find $OUT -name "*counter_collection.csv" -size +4M -delete; find $OUT -name "*_kernel_trace.csv" -size +4M -delete
