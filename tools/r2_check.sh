#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the GPU test suite, then the default bench line and the other single-GPU workloads.
# Usage: tools/r2_check.sh <tag>   -> gpurun_out/<tag>/
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
if [ $rc -ne 0 ]; then echo "GPU tests failed ($rc)"; exit $rc; fi
for cfg in "default:" "synthetic:--scene synthetic" "lbs:--workload lbs" "merged:--workload mixed" "ybot_cheese:--mesh ybot"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  echo "== bench $name ($flags)"
  timeout -k 10 400 python bench.py $flags $BENCH_EXTRA > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { echo "bench $name failed"; tail -5 $OUT/bench_$name.err; exit 1; }
  python - "$OUT/bench_$name.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "whole_path_hbm_frac", "kernels_ms_per_step")}, d["roofline"]["frac"], d["ccd"], d.get("cpu_baseline", {}).get("value"))
PY
done
