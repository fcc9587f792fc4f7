"""Diagnostic: the separation stage on a crowd with an environment switch of the library toggled EVERY step inside one process (both
settings give the same results, so both see the same crowd): ms per step of each setting.
usage: separation_ab.py ENV_NAME [n]   e.g. separation_ab.py SGE_SEPARATION_NO_DEFER 8192"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
ybot = sge.assets.YBotAssets()
eng = sge.CharacterEngine(0)
sge.crowd.upload_character_assets(eng, ybot, rings=3, segments=3)
scene = sge.crowd.upload_asset_scene(eng, ("cheese", "mirror"), footprint=200.0)
sge.crowd.spawn_crowd(eng, ybot, n, scene, seed=43, agents=True, mixed=True)
st = (abi.STAGE_ALL & ~abi.STAGE_SKIN) | abi.STAGE_SEPARATION
for _ in range(30):
    eng.tick(stages=st)
eng.synchronize()
t = {"0": [], "1": []}
for s in range(60):
    mode = "01"[s & 1]
    os.environ[name] = mode
    eng.synchronize()
    t0 = time.perf_counter()
    eng.tick(stages=st)
    eng.synchronize()
    t[mode].append((time.perf_counter() - t0) * 1e3)
for mode in ("0", "1"):
    a = np.array(t[mode])
    print("%s=%s: ms per step mean %.2f, first ten %.2f, last ten %.2f" % (name, mode, a.mean(), a[:10].mean(), a[-10:].mean()))
