#!/bin/bash
# Runs ON THE GPU BOX: tools/lbs_sustained.py under rocprofv3 --kernel-trace; prints the skin_kernel durations in launch order.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/lbs_sustained
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/trace -o run --output-format csv -- python3 $REPO/tools/lbs_sustained.py > $OUT/run.log 2>&1
python3 - $(find $OUT/trace -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1])) if "skin_kernel" in r["Kernel_Name"]]
rows.sort()
d = [(b - a) / 1e6 for a, b in rows][-130:]
gap = [(rows[i][0] - rows[i - 1][1]) / 1e6 for i in range(len(rows) - 129, len(rows))]
print("back to back (100 launches), ms:", " ".join("%.3f" % x for x in d[:100]))
print("mean of launches 1-10 %.3f, 11-50 %.3f, 51-100 %.3f" % (sum(d[:10]) / 10, sum(d[10:50]) / 40, sum(d[50:100]) / 50))
print("with a pause before each (30 launches), ms:", " ".join("%.3f" % x for x in d[100:]))
print("mean %.3f; mean idle gap before these launches %.3f ms" % (sum(d[100:]) / 30, sum(gap[100:]) / max(len(gap[100:]), 1)))
PY
find $OUT -name "*_kernel_trace.csv" -delete
