"""Diagnostic: move-stage time on the real 17-Cheese scene for heavy-kernel variants (SGE_AMD_LIB) and thresholds."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
eng = sge.CharacterEngine(0)
ybot = sge.assets.YBotAssets()
sge.crowd.upload_character_assets(eng, ybot, rings=3, segments=3)
synthetic = "--synthetic" in sys.argv
scene = sge.crowd.upload_terrain(eng) if synthetic else sge.crowd.upload_asset_scene(eng, ("cheese",))
n = 10000
state = sge.crowd.spawn_crowd(eng, ybot, n, scene)
st = abi.STAGE_INTENT | abi.STAGE_GRAVITY | abi.STAGE_MOVE
for thr in [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [-1, 8000, 4000, 2000, 1000, 500]:
    eng.upload(**state)
    eng.set_option(abi.OPT_HEAVY_THRESHOLD, thr)
    for _ in range(140):
        eng.tick(stages=st)
    eng.synchronize()
    eng.set_option(abi.OPT_PROFILE, 1)
    eng.profile_read(reset=True)
    for _ in range(60):
        eng.tick(stages=st)
    eng.synchronize()
    p = eng.profile_read(reset=True)
    eng.set_option(abi.OPT_PROFILE, 0)
    print("%s threshold %6d: move %.3f ms/step" % (os.environ.get("SGE_AMD_LIB", "default"), thr, p.move_ms / 60), flush=True)
eng.close()
