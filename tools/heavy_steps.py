"""Diagnostic: per-step move-stage time on the real 17-Cheese scene (how spiky is it, and who is heavy?)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
eng = sge.CharacterEngine(0)
ybot = sge.assets.YBotAssets()
sge.crowd.upload_character_assets(eng, ybot, rings=3, segments=3)
scene = sge.crowd.upload_asset_scene(eng, ("cheese",))
n = 10000
state = sge.crowd.spawn_crowd(eng, ybot, n, scene)
st = abi.STAGE_INTENT | abi.STAGE_GRAVITY | abi.STAGE_MOVE
for thr in (4000, -1):
    eng.upload(**state)
    eng.set_option(abi.OPT_HEAVY_THRESHOLD, thr)
    for _ in range(140):
        eng.tick(stages=st)
    eng.synchronize()
    eng.set_option(abi.OPT_PROFILE, 1)
    times = []
    for _ in range(80):
        eng.profile_read(reset=True)
        eng.tick(stages=st)
        eng.synchronize()
        times.append(eng.profile_read(reset=True).move_ms)
    eng.set_option(abi.OPT_PROFILE, 0)
    t = np.array(times)
    print("threshold %d: per-step move ms  min %.3f  p25 %.3f  median %.3f  p75 %.3f  max %.3f  mean %.3f" % (
        thr, t.min(), np.percentile(t, 25), np.median(t), np.percentile(t, 75), t.max(), t.mean()))
    print("   ", np.round(t[:40], 2).tolist())
eng.close()
