#!/bin/bash
# Runs ON THE GPU BOX: two settings (env assignments, "-" for none) over the single-GPU workloads of tools/r3_lines.sh, interleaved.
#   tools/workload_ab.sh "SGE_SKIN_CPW=4" "SGE_SKIN_CPW=8" [reps]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
A="$1"; B="$2"; REPS=${3:-2}
[ "$A" = "-" ] && A=""; [ "$B" = "-" ] && B=""
for w in "--workload lbs" "--workload mixed" "--mesh ybot" "--scene synthetic" "--workload agents" "--chars 2500" "--chars 20000" "--mesh ybot --workload mixed --chars 31250" "--workload agents --chars 31250"; do
  for r in $(seq $REPS); do
    for s in "$A" "$B"; do
      env $s python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline $w 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-48s %-22s %.4f ms  %.3f M chars/s  lbs %.3f move %.3f pose %.3f' % ('$w', '$s' or '-', d['ms_per_step'], d['value']/1e6, d['kernels_ms_per_step']['lbs'], d['kernels_ms_per_step']['move_ccd'], d['kernels_ms_per_step']['pose']))"
    done
  done
done
