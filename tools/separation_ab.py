"""Diagnostic: the separation stage on a crowd with an environment switch of the library off / on: two engines with the same crowd in
one process, stepped in lockstep (the settings give the same results, so both see the same crowd at every step), every step
synchronised; ms per step of each.
usage: separation_ab.py ENV_NAME [n]   e.g. separation_ab.py SGE_SEPARATION_NO_DEFER 8192"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
ybot = sge.assets.YBotAssets()
engines = []
for _ in range(2):
    eng = sge.CharacterEngine(0)
    sge.crowd.upload_character_assets(eng, ybot, rings=3, segments=3)
    scene = sge.crowd.upload_asset_scene(eng, ("cheese", "mirror"), footprint=200.0)
    sge.crowd.spawn_crowd(eng, ybot, n, scene, seed=43, agents=True, mixed=True)
    engines.append(eng)
st = (abi.STAGE_ALL & ~abi.STAGE_SKIN) | abi.STAGE_SEPARATION
t = [[], []]
for s in range(70):
    for k, eng in enumerate(engines):
        os.environ[name] = str(k)
        t0 = time.perf_counter()
        eng.tick(stages=st)
        eng.synchronize()
        t[k].append((time.perf_counter() - t0) * 1e3)
a, b = (engines[k].download(what=("bodies",))["bodies"].tobytes() for k in range(2))
assert a == b
for k in range(2):
    x = np.array(t[k])
    print("%s=%d: ms per step, steps 30-49: %.2f, steps 50-69: %.2f" % (name, k, x[30:50].mean(), x[50:70].mean()))
