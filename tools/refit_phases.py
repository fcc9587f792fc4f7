"""Diagnostic (GPU box; needs the variant library built by `tools/build_variant.sh blasexp4 -DSGE_BLAS_EXPERIMENT=4`): where a
workgroup of blas_refit_kernel spends a step, in shader-clock cycles summed over the launch, seen by wave 0 of every workgroup.
usage: SGE_AMD_LIB=libsge_amd_blasexp4.so SGE_BLAS_RAW=0 python tools/refit_phases.py [--real]"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
n = 10000
eng = sge.CharacterEngine(0)
ybot = sge.assets.YBotAssets()
if "--real" in sys.argv:
    sge.crowd.upload_ybot_mesh(eng, ybot)
else:
    sge.crowd.upload_character_assets(eng, ybot)
sge.crowd.spawn_crowd(eng, ybot, n, None, mode="lbs")
info = eng.blas_build(eng.mesh["indices"])
eng.set_option(abi.OPT_FUSE_BLAS_REFIT, 0)
st = abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_POSE | abi.STAGE_WRITEBACK | abi.STAGE_SKIN | abi.STAGE_BLAS_REFIT
for _ in range(5):
    eng.tick(stages=st)
eng.synchronize()
out = np.zeros((1024, 12), np.uint64)
fn = eng.t.lib.sge_experiment_blas_phases
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(out.ctypes.data, 1) == 0
launches = 20
for _ in range(launches):
    eng.tick(stages=st)
eng.synchronize()
assert fn(out.ctypes.data, 0) == 0
used = out[out[:, :8].sum(1) > 0].astype(np.float64) / launches
names = ["barrier 1 (others' walks)", "wait for the tile's loads", "LDS write + barrier 2", "issue round words + next tile", "round words' latency", "walk", "end of character", "-"]
tot = used[:, :8].sum(1).mean()
print("workgroups %d, cycles per workgroup per launch %.0f" % (len(used), tot))
for k, nm in enumerate(names[:7]):
    print("  %-32s %9.0f cycles  %5.1f %%" % (nm, used[:, k].mean(), 100 * used[:, k].mean() / tot))
for k, nm in enumerate(["first barrier (slowest walk)", "level reductions", "later barriers", "write-out + re-initialisation"]):
    print("    end of character: %-28s %9.0f cycles  %5.1f %%" % (nm, used[:, 8 + k].mean(), 100 * used[:, 8 + k].mean() / tot))
