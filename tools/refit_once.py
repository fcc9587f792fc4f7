"""Diagnostic: a few refit launches (for instrumented builds that print from the kernel). usage: refit_once.py [--real]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
eng = sge.CharacterEngine(0)
ybot = sge.assets.YBotAssets()
(sge.crowd.upload_ybot_mesh if "--real" in sys.argv else sge.crowd.upload_character_assets)(eng, ybot)
sge.crowd.spawn_crowd(eng, ybot, 10000, None, mode="lbs")
eng.blas_build(eng.mesh["indices"])
st = abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_POSE | abi.STAGE_WRITEBACK | abi.STAGE_SKIN
for _ in range(3):
    eng.tick(stages=st)
eng.synchronize()
for _ in range(3):
    eng.blas_refit()
    eng.synchronize()
