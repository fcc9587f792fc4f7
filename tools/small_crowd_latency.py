"""Diagnostic: wall-clock per fixed step for small crowds (launch-bound regime): tick + synchronize, and tick only."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
for n in (1, 16, 256, 2048):
    eng = sge.CharacterEngine(0)
    ybot = sge.assets.YBotAssets()
    sge.crowd.upload_character_assets(eng, ybot)
    terrain = sge.crowd.upload_terrain(eng)
    sge.crowd.spawn_crowd(eng, ybot, n, terrain)
    for opt in (4000, 0):
        eng.set_option(abi.OPT_HEAVY_THRESHOLD, opt)
        for _ in range(150):
            eng.tick()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            eng.tick()
            eng.synchronize()
        t1 = time.perf_counter()
        for _ in range(300):
            eng.tick()
        eng.synchronize()
        t2 = time.perf_counter()
        print("n %5d heavy-threshold %d: tick+sync %.1f us/step, pipelined %.1f us/step" % (n, opt, (t1 - t0) / 300 * 1e6, (t2 - t1) / 300 * 1e6), flush=True)
    eng.close()
