"""Experiment driver (GPU box): bench.py over library variants (SGE_AMD_LIB) and env settings.
Usage: variant_sweep.py lib1,lib2[,...][xN] [bench flags]   (a trailing xN repeats the whole list N times, interleaved; a summary
with mean / std / min per variant follows the per-run lines — boxes and processes differ by several percent, single runs do not)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = sys.argv[1]
reps = 1
if "x" in spec.rsplit(",", 1)[-1] and spec.rsplit("x", 1)[-1].isdigit():
    spec, r = spec.rsplit("x", 1)
    reps = int(r)
libs = spec.split(",") * reps
flags = sys.argv[2:]
results = {}
for lib in libs:
    env = dict(os.environ)
    name = lib
    if "@" in lib:  # lib@VAR=val@VAR=val
        parts = lib.split("@")
        lib = parts[0]
        for kv in parts[1:]:
            k, v = kv.split("=")
            env[k] = v
    if lib:
        env["SGE_AMD_LIB"] = lib
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "10", "--no-cpu-baseline"] + flags, env=env, capture_output=True, text=True)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        k = d["kernels_ms_per_step"]
        results.setdefault(name, []).append((d["ms_per_step"], k["lbs"], d.get("lbs_alone", {}).get("ms_per_launch", 0.0), k["move_ccd"], k["pose"]))
        print("%-40s step %.3f ms frac %.3f | lbs %.3f (alone %.3f) move %.3f pose %.3f | trips/q %.2f" % (name, d["ms_per_step"], d["whole_path_hbm_frac"], k["lbs"], d.get("lbs_alone", {}).get("ms_per_launch", 0.0), k["move_ccd"], k["pose"], d["ccd"]["sweep_trips_per_query"]), flush=True)
    except Exception as ex:
        print(name, "FAILED", ex, p.stderr[-300:], flush=True)
if reps > 1:
    import statistics as st
    print("---- summary over %d runs each: step mean +- std (min) | lbs | alone | move | pose" % reps)
    for name, rows in results.items():
        cols = list(zip(*rows))
        print("%-40s %.3f +- %.3f (%.3f) | %.3f | %.3f | %.3f | %.3f" % (name[-40:], st.mean(cols[0]), st.pstdev(cols[0]), min(cols[0]), st.mean(cols[1]), st.mean(cols[2]), st.mean(cols[3]), st.mean(cols[4])), flush=True)
