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
    for opt in (0, 1):
        if opt and not hasattr(abi, "OPT_GRAPH"):
            continue
        if hasattr(abi, "OPT_GRAPH"):
            eng.set_option(abi.OPT_GRAPH, opt)
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
        print("n %5d graph %d: tick+sync %.1f us/step, pipelined %.1f us/step" % (n, opt, (t1 - t0) / 300 * 1e6, (t2 - t1) / 300 * 1e6), flush=True)
    eng.close()
