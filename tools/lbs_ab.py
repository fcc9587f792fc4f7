"""Diagnostic: A/B several builds of the library (different skin_kernel variants) inside ONE process, interleaved, so that
box-to-box and run-to-run noise cancels.  usage: lbs_ab.py libA.so libB.so ..."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
real = "--real" in sys.argv
libs = [a for a in sys.argv[1:] if a.endswith(".so")] or ["libsge_amd.so"]
pkgdir = os.path.dirname(sge.__file__)
engines = []
ybot = sge.assets.YBotAssets()
n = 10000
st = abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_POSE | abi.STAGE_WRITEBACK | abi.STAGE_SKIN
for lib in libs:
    table = sge.engine._ProductTable(0, lib_path=os.path.join(pkgdir, lib))
    eng = sge.CharacterEngine(table=table)
    if real:
        sge.crowd.upload_ybot_mesh(eng, ybot)
    else:
        sge.crowd.upload_character_assets(eng, ybot)
    sge.crowd.spawn_crowd(eng, ybot, n, None, mode="lbs")
    for _ in range(10):
        eng.tick(stages=st)
    eng.synchronize()
    eng.set_option(abi.OPT_PROFILE, 1)
    engines.append(eng)
res = {lib: [] for lib in libs}
for rnd in range(8):
    for lib, eng in zip(libs, engines):
        eng.profile_read(reset=True)
        for _ in range(40):
            eng.tick(stages=st)
        eng.synchronize()
        p = eng.profile_read(reset=True)
        res[lib].append(p.skin_ms / p.skin_launches)
for lib in libs:
    t = np.array(res[lib])
    print("%-22s skin ms: min %.4f median %.4f mean %.4f" % (lib, t.min(), np.median(t), t.mean()))
