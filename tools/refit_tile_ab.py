"""Diagnostic: blas_refit_kernel at tile capacities 3072 / 2048 (/ 4096) in ONE process over the same position stream (the
schedule is rebuilt per capacity, the streams stay where they are): ms per launch and HBM rate, alternating.
usage: refit_tile_ab.py [--real] [--chars N] [caps ...]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
args = [a for a in sys.argv[1:]]
real = "--real" in args
n = int(args[args.index("--chars") + 1]) if "--chars" in args else 10000
caps = [int(a) for a in args if a.isdigit() and int(a) in (1024, 1536, 2048, 3072, 4096)] or [3072, 2048]
eng = sge.CharacterEngine(0)
ybot = sge.assets.YBotAssets()
if real:
    sge.crowd.upload_ybot_mesh(eng, ybot)
else:
    sge.crowd.upload_character_assets(eng, ybot)
sge.crowd.spawn_crowd(eng, ybot, n, None, mode="lbs")
eng.set_option(abi.OPT_FUSE_BLAS_REFIT, 0)
st = abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_POSE | abi.STAGE_WRITEBACK | abi.STAGE_SKIN
for _ in range(5):
    eng.tick(stages=st)
eng.synchronize()
eng.set_option(abi.OPT_PROFILE, 1)
V = eng.vertex_count
res = {c: [] for c in caps}
ref = None
for rnd in range(3):
    for cap in caps:
        os.environ["SGE_BLAS_TILE_CAP"] = str(cap)
        info = eng.blas_build(eng.mesh["indices"])
        for _ in range(5):
            eng.blas_refit()
        eng.synchronize()
        eng.blas_profile(reset=True)
        for _ in range(40):
            eng.blas_refit()
        eng.synchronize()
        ms, k = eng.blas_profile(reset=True)
        res[cap].append(ms / max(k, 1))
        b = eng.blas_bounds(0, 64)
        root = b[:, -1, :].copy()  # the root box does not depend on the tiling
        if ref is None:
            ref = root
        assert ref.tobytes() == root.tobytes()
nbytes = n * (V * 12 + (info.entryCount + 1) * 24)
for cap in caps:
    t = np.median(res[cap])
    print("tile cap %4d: ms/launch %s -> median %.4f = %.0f GB/s = %.3f of 8 TB/s" % (cap, " ".join("%.4f" % x for x in res[cap]), t, nbytes / t / 1e6, nbytes / t / 8e9))
