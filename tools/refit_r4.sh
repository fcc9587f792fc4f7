#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round-4 evidence of the refit kernel, written under gpurun_out/r4_refit/:
#   bench_*.txt         tools/refit_bench.py: HIP-event time per launch, both meshes, both layouts, the fused form
#   phases_*.txt        in-kernel phase stamps (needs tools/build_variant.sh blasexp4 -DSGE_BLAS_EXPERIMENT=4)
#   kernel_stats_*.csv  rocprofv3 --kernel-trace --stats of the same command
#   pmc_*.txt           FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes, MI355X_MICROARCH.md's HBM recipe)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r4_refit
rm -rf $OUT; mkdir -p $OUT
cd $REPO
for args in "" "--real" "--padded" "--fuse" "--real --fuse"; do
  name=$(echo "syn $args" | tr -d '-' | tr ' ' '_')
  timeout -k 10 200 python tools/refit_bench.py $args 2>&1 | grep -v amdgpu.ids > $OUT/bench_$name.txt
  echo "== refit_bench.py $args"; cat $OUT/bench_$name.txt
done
if [ -f swift-game-engine_amd/libsge_amd_blasexp4.so ]; then
  for args in "" "--real"; do
    name=$(echo "syn $args" | tr -d '-' | tr ' ' '_')
    SGE_AMD_LIB=libsge_amd_blasexp4.so timeout -k 10 200 python tools/refit_phases.py $args 2>&1 | grep -v amdgpu.ids > $OUT/phases_$name.txt
    echo "== refit_phases.py $args"; cat $OUT/phases_$name.txt
  done
fi
cd /tmp && export TMPDIR=/tmp
for args in "" "--real"; do
  name=$(echo "syn $args" | tr -d '-' | tr ' ' '_')
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/trace_$name -o run --output-format csv -- python3 $REPO/tools/refit_bench.py $args > /dev/null 2> $OUT/trace_$name.err
  f=$(find $OUT/trace_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$name.csv
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_${ctr}_$name -o run --output-format csv -- python3 $REPO/tools/refit_bench.py $args > /dev/null 2> $OUT/pmc_${ctr}_$name.err
  done
  python3 - <<PY > $OUT/pmc_$name.txt
import csv, glob, collections
per = collections.defaultdict(list)
for path in glob.glob("$OUT/pmc_*_$name/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "blas_refit_kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(per.items()):
    print("%s per launch of blas_refit_kernel: mean %.6g (KiB as rocprofv3 reports them; FETCH_SIZE x 2 on gfx950 for wide streaming reads, MI355X_MICROARCH.md, HBM3E section), launches %d" % (k, sum(v) / len(v), len(v)))
PY
  echo "== counters $args"; cat $OUT/pmc_$name.txt
done
find $OUT -name "*.csv" -size +1M -delete
rm -rf $OUT/trace_* $OUT/pmc_FETCH* $OUT/pmc_WRITE*
