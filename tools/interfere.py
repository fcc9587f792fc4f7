"""Interference probe (GPU box): the product's LBS kernel (10k characters x 14,080 vertices, SKIN stage only) timed alone and beside
co-runner kernels of tools/interfere.hip that each load ONE resource. Usage: python tools/interfere.py [cap]   (cap = LBS workgroups/CU, 0 = no cap)
Prints LBS ms per launch (HIP events on the engine's skin stream) per co-runner configuration."""
import ctypes as C
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "libinterfere.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "tools", "interfere.hip"), "-o", so])
cap = sys.argv[1] if len(sys.argv) > 1 else "3"
os.environ["SGE_OVERLAP_SKIN_WORKGROUPS"] = cap

import torch  # noqa: E402  (device init)
import bench  # noqa: E402
import __graft_entry__  # noqa: E402

torch.cuda.set_device(0)
sge = __graft_entry__.build()
abi = sge.abi
itf = C.CDLL(so)
itf.itf_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long]
itf.itf_time.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long]
itf.itf_time.restype = C.c_float

args = types.SimpleNamespace(mesh="synthetic", scene="cheese")
eng = sge.CharacterEngine(0)
eng.set_option(abi.OPT_SKIN_LAYOUT, abi.LAYOUT_PACKED)
eng.set_option(abi.OPT_OVERLAP_SKIN, 1)  # the capped launch on the skin stream, as in the default bench line
ybot = sge.assets.YBotAssets()
terrain = bench._build_world(sge, eng, ybot, args)
N = 10000
eng.resize(N)
bench._spawn_block(sge, eng, ybot, N, 0, N, terrain, "lbs", agents=False)
stages = abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_POSE | abi.STAGE_WRITEBACK | abi.STAGE_SKIN
for _ in range(5):
    eng.tick(stages=stages)
eng.synchronize()
assert itf.itf_init() == 0
eng.set_option(abi.OPT_PROFILE, 1)


def lbs(launches=12):
    eng.profile_read(reset=True)
    for _ in range(launches):
        eng.tick(dt=0.0, stages=abi.STAGE_SKIN)
    eng.synchronize()
    p = eng.profile_read(reset=True)
    return p.skin_ms / max(p.skin_launches, 1)


NAMES = {0: "valu dependent chain", 1: "valu 8 chains", 2: "lds", 3: "l2 pointer chase", 4: "salu", 5: "hbm read", 6: "valu chain, 3 waves/SIMD registers"}
print("LBS cap %s workgroups/CU; alone: %.3f ms" % (cap, lbs()), flush=True)
# (kind, workgroups, threads): 1024 one-wave workgroups = one wave per SIMD, the residency the grouped move kernel gets beside the LBS kernel
cases = [(0, 1024, 64), (0, 2048, 64), (0, 4096, 64), (1, 1024, 64), (1, 2048, 64), (6, 1024, 64), (6, 3072, 64), (2, 1024, 64), (2, 4096, 64), (3, 1024, 64), (3, 4096, 64),
         (4, 1024, 64), (4, 4096, 64), (5, 512, 256), (5, 2048, 256)]
for kind, wgs, threads in cases:
    iters = 2000
    t = itf.itf_time(kind, wgs, threads, iters)
    # size the co-runner to ~40 ms so that it covers all LBS launches of one measurement
    iters = max(int(iters * 40.0 / max(t, 1e-3)), 1)
    t40 = itf.itf_time(kind, wgs, threads, iters)
    assert itf.itf_launch(kind, wgs, threads, iters) == 0
    ms = lbs(12)
    itf.itf_sync()
    # and the other way round: how much longer the co-runner takes beside 12 LBS launches is not measured; its alone time is printed
    print("%-38s %5d x %3d  co-runner alone %.1f ms | LBS %.3f ms" % (NAMES[kind], wgs, threads, t40, ms), flush=True)
print("alone again: %.3f ms" % lbs(), flush=True)
eng.close()
