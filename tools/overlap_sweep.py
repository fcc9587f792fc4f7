"""Experiment driver (GPU box): bench.py under occupancy caps (SGE_SKIN_LDS_PAD / SGE_MOVE_LDS_PAD, dynamic-LDS padding that
limits resident workgroups per CU) with and without --overlap. Prints one line per configuration."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "10", "--no-cpu-baseline"]


def run(tag, flags, env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    p = subprocess.run(base + flags, env=e, capture_output=True, text=True)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        k = d["kernels_ms_per_step"]
        print("%-34s step %.3f ms  frac %.3f | lbs %.3f move %.3f pose %.3f" % (tag, d["ms_per_step"], d["whole_path_hbm_frac"], k["lbs"], k["move_ccd"], k["pose"]), flush=True)
    except Exception as ex:
        print(tag, "FAILED", ex, p.stderr[-400:], flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "lbs"):
    for pad in (0, 27000, 40000, 66000):
        run("lbs-only skinpad=%d" % pad, ["--workload", "lbs"], {"SGE_SKIN_LDS_PAD": pad})
if which == "prio":
    for sp in (40000, 66000):
        for env in ({}, {"SGE_SKIN_SETPRIO": 3}, {"SGE_SKIN_STREAM_HIGH": 1}, {"SGE_SKIN_SETPRIO": 3, "SGE_SKIN_STREAM_HIGH": 1}):
            run("overlap skinpad=%d %s" % (sp, " ".join(env)), ["--overlap"], dict(env, SGE_SKIN_LDS_PAD=sp))
    run("overlap nopad prio3+high", ["--overlap"], {"SGE_SKIN_SETPRIO": 3, "SGE_SKIN_STREAM_HIGH": 1})
if which in ("all", "overlap"):
    run("serial", [], {})
    for sp, mp in ((0, 0), (27000, 0), (40000, 0), (66000, 0), (40000, 12000), (66000, 12000), (66000, 30000), (40000, 30000)):
        run("overlap skinpad=%d movepad=%d" % (sp, mp), ["--overlap"], {"SGE_SKIN_LDS_PAD": sp, "SGE_MOVE_LDS_PAD": mp})
