"""Summarises the rocprofv3 outputs of tools/collect_profiles.sh: per-kernel duration stats and PMC byte counters."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    return hits[0] if hits else None


summary = {}
stats = find("trace/**/*kernel_stats.csv")
if stats:
    summary["kernel_stats_csv"] = os.path.relpath(stats, out)
    rows = list(csv.DictReader(open(stats)))
    summary["kernels"] = {r["Name"]: {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                      "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6,
                                      "pct": float(r["Percentage"])} for r in rows}
    import shutil
    shutil.copy(stats, os.path.join(out, "kernel_stats.csv"))
counters = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    path = find("pmc_%s/**/*counter_collection.csv" % ("fetch" if name == "FETCH_SIZE" else "write"))
    if not path:
        continue
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == name:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    counters[name] = {k: {"launches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)} for k, v in per.items()}
summary["counters"] = counters
try:
    summary["bench"] = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa
    summary["bench_error"] = str(e)
# the LBS kernel of the profiled schedule: skin_kernel (serial order / hand-over), skin_ticket_kernel or skin_ticket_multi_kernel (resident)
skin = sorted([k for k in counters.get("WRITE_SIZE", {}) if "skin_" in k and "refit" not in k and "jobs" not in k],
              key=lambda k: -counters["WRITE_SIZE"][k]["launches"])
per_kernel = {}
for k in skin:
    if k in counters.get("FETCH_SIZE", {}):
        per_kernel[k.split("(")[0].replace("void ", "")] = counters["WRITE_SIZE"][k]["mean_KiB"] * 1024 + 2 * counters["FETCH_SIZE"][k]["mean_KiB"] * 1024
if skin and skin[0] in counters.get("FETCH_SIZE", {}):
    w = counters["WRITE_SIZE"][skin[0]]["mean_KiB"] * 1024
    f = counters["FETCH_SIZE"][skin[0]]["mean_KiB"] * 1024
    summary["skin_kernel"] = {"kernel": skin[0], "WRITE_SIZE_bytes": w, "FETCH_SIZE_bytes_raw": f, "FETCH_SIZE_bytes_corrected_x2": 2 * f,
                              "hbm_bytes_per_launch": w + 2 * f,
                              "note": "gfx950: FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section) -> doubled; "
                                      "WRITE_SIZE is exact for 16-B-per-lane streaming stores"}
    # the record bench.py reports as roofline.traffic, valid for this kernel source and this shape only
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cfg = summary.get("bench", {}).get("config", {})
    json.dump({"characters": cfg.get("characters_per_gpu"), "vertices": cfg.get("vertices_per_character"),
               "hbm_bytes_per_launch": w + 2 * f, "skin_source_hash": bench.skin_source_hash(), "kernel": skin[0].split("(")[0],
               "hbm_bytes_per_launch_by_kernel": per_kernel,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of tools/collect_profiles.sh (FETCH_SIZE doubled: gfx950 correction)"},
              open(os.path.join(out, "lbs_traffic.json"), "w"))
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({k: summary[k] for k in summary if k != "bench"}, indent=1)[:3000])
