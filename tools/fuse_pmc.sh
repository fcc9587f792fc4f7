set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fuse_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o run --output-format csv -- python3 $REPO/tools/refit_bench.py --fuse > $OUT/$c.log 2> $OUT/$c.err
done
python3 - <<PY
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "skin_refit" in r["Kernel_Name"] and r["Counter_Name"] == c]
        out[c + "_KiB_mean"] = sum(vals) / len(vals); out[c + "_launches"] = len(vals)
json.dump(out, open("$OUT/fused_pmc.json", "w"), indent=1); print(json.dumps(out))
PY
find $OUT -name "*.csv" -delete
