#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): tools/placement_pmc under rocprofv3 --pmc, one counter group per pass; per-launch means
# of the fast (three<0>) and the slow (three<1>) placement side by side -> gpurun_out/r4_placement/
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4_placement
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 $REPO/tools/placement_pmc ${1:-96} ${2:-4} > $OUT/plain_run.txt 2>&1; tail -8 $OUT/plain_run.txt
k=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_WRREQ_64B_sum TCC_NORMAL_WRITEBACK_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "WRITE_SIZE"; do
  k=$((k+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$k -o run --output-format csv -- $REPO/tools/placement_pmc ${1:-96} ${2:-4} > $OUT/g$k.log 2> $OUT/g$k.err || tail -3 $OUT/g$k.err
  grep "fast placement\|three<" $OUT/g$k.log | tail -3
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        which = "fast" if "three<0>" in name else "slow" if "three<1>" in name else None
        if which:
            per[r["Counter_Name"]][which].append(float(r["Counter_Value"]))
for path in glob.glob("$OUT/g*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        which = "fast" if "three<0>" in name else "slow" if "three<1>" in name else None
        if which:
            dur[which].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("kernel duration under the profiler, ms: fast %.4f slow %.4f" % (sum(dur["fast"]) / max(len(dur["fast"]), 1), sum(dur["slow"]) / max(len(dur["slow"]), 1)))
print("%-50s %16s %16s %8s" % ("counter (mean per launch)", "fast placement", "slow placement", "slow/fast"))
for c, d in sorted(per.items()):
    f = sum(d["fast"]) / max(len(d["fast"]), 1); s = sum(d["slow"]) / max(len(d["slow"]), 1)
    print("%-50s %16.6g %16.6g %8.3f" % (c, f, s, s / f if f else float("nan")))
PY
find $OUT -name "*.csv" -size +1M -delete
