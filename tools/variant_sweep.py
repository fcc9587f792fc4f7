"""Experiment driver (GPU box): bench.py over library variants (SGE_AMD_LIB) and env settings. Usage: variant_sweep.py lib1,lib2 [bench flags]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1].split(",")
flags = sys.argv[2:]
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
        print("%-40s step %.3f ms frac %.3f | lbs %.3f (alone %.3f) move %.3f pose %.3f | trips/q %.2f" % (name, d["ms_per_step"], d["whole_path_hbm_frac"], k["lbs"], d.get("lbs_alone", {}).get("ms_per_launch", 0.0), k["move_ccd"], k["pose"], d["ccd"]["sweep_trips_per_query"]), flush=True)
    except Exception as ex:
        print(name, "FAILED", ex, p.stderr[-300:], flush=True)
