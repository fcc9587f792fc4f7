"""Diagnostic: the character-vs-character step (one GPU's share of configs[4]: 10k characters, pose + LBS + CCD with agent sweeps,
overlap schedule) with an environment switch toggled between blocks of steps inside ONE process (same allocations).
usage: agents_ab.py ENV_NAME [chars]   e.g. agents_ab.py SGE_AGENT_GRID_INLINE"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
ybot = sge.assets.YBotAssets()
eng = sge.CharacterEngine(0)
sge.crowd.upload_character_assets(eng, ybot)
scene = sge.crowd.upload_asset_scene(eng, ("cheese",), footprint=200.0)
sge.crowd.spawn_crowd(eng, ybot, n, scene, seed=43, agents=True)
eng.set_option(abi.OPT_OVERLAP_SKIN, 2)
ex = sge.parallel.AgentExchange(eng, n, 0, 1, torch.device("cuda", 0), None)
for _ in range(40):
    ex.step()
eng.synchronize()
res = {"0": [], "1": []}
for rnd in range(5):
    for mode in ("1", "0"):
        os.environ[name] = mode
        for _ in range(10):
            ex.step()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            ex.step()
        eng.synchronize()
        res[mode].append((time.perf_counter() - t0) / 100 * 1e3)
for mode in ("0", "1"):
    print("%s=%s: ms per step %s -> median %.4f" % (name, mode, " ".join("%.4f" % x for x in res[mode]), np.median(res[mode])))
