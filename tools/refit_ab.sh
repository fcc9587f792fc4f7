#!/bin/bash
# Runs ON THE GPU BOX: interleaved A/B of two library builds on the refit lines (bench.py --refit, serial and overlapped; fused).
# Usage: tools/refit_ab.sh <libA> <libB> <out dir under gpurun_out> [repeats]
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; OUT=$REPO/gpurun_out/$3; N=${4:-3}
mkdir -p $OUT
cd $REPO
for i in $(seq 1 $N); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    for cfg in "ser_unfused:--refit --no-overlap" "ser_fused:--refit --fuse --no-overlap" "ov_unfused:--refit"; do
      name=${cfg%%:*}; flags=${cfg#*:}
      SGE_AMD_LIB=$lib timeout -k 10 300 python bench.py $flags --no-cpu-baseline > $OUT/${name}_${v}_$i.json 2> $OUT/${name}_${v}_$i.err || { echo "bench $name $v failed"; tail -n 5 $OUT/${name}_${v}_$i.err; exit 1; }
      python - $OUT/${name}_${v}_$i.json $name $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels_ms_per_step"]
print("%-12s %s step %.4f lbs %.4f refit %.4f move %.4f" % (sys.argv[2], sys.argv[3], d["ms_per_step"], k["lbs"], k.get("blas_refit", 0.0), k["move_ccd"]), flush=True)
PY
    done
  done
done
