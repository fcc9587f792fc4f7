"""Diagnostic: where does a move_kernel wave spend its cycles? Uses the -DSGE_CCD_TIMING build."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
sge.abi.LIB_NAME = "libsge_amd_timing.so"
eng = sge.CharacterEngine(0)
ybot = sge.assets.YBotAssets()
sge.crowd.upload_character_assets(eng, ybot, rings=4, segments=4)
which = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
terrain = sge.crowd.upload_terrain(eng) if which == "synthetic" else sge.crowd.upload_asset_scene(eng, tuple(which.split(",")))
n = 10000
sge.crowd.spawn_crowd(eng, ybot, n, terrain)
st = sge.abi.STAGE_ALL_FIXED
for _ in range(140):
    eng.tick(stages=st)
eng.move_stats(reset=True)
steps = 20
for _ in range(steps):
    eng.tick(stages=st)
s = eng.move_stats(reset=True)
waves = n * steps
print("cycles per wave (character step): total %.0f  traversal %.0f  sweep %.0f  other %.0f" % (
    s.sweepIterations / waves, s.traversalSteps / waves, s.sweepTrips / waves,
    (s.sweepIterations - s.traversalSteps - s.sweepTrips) / waves))
print("queries/char-step %.2f" % (s.queries / waves))
