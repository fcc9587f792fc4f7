"""Debug: the soak-test scene (agents, real assets) stepped on GPU and oracle, compared after every step."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sge = importlib.import_module("swift-game-engine_amd")
import oracle_binding as ob
from scenes import build_scene
import torch
gpu = sge.CharacterEngine(0)
cpu = ob.oracle_engine()
n = 160
for e in (gpu, cpu):
    build_scene(sge, e, n, seed=77, mixed=True, agents=True, rings=3, segments=3, asset_scene=("cheese", "semla"), footprint=120.0)
ex = sge.parallel.AgentExchange(gpu, n, 0, 1, torch.device("cuda", 0), None)
st = sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN
for s in range(400):
    before = gpu.download()
    cost_before = gpu.move_cost()
    ex.step(stages=st)
    ob.tick_mt(cpu, 8, stages=st | sge.abi.STAGE_AGENTS)
    gpu.synchronize()
    g, c = gpu.download(), cpu.download()
    lists = np.zeros(2 * n, np.int32); counts = np.zeros(2, np.int32)
    assert gpu.t.lib.sge_debug_move_lists(gpu.h, sge.abi.ptr(lists), sge.abi.ptr(counts)) == 0
    members = np.concatenate([lists[:counts[0]], lists[n:n + counts[1]]])
    if len(members) != n or len(np.unique(members)) != n:
        missing = sorted(set(range(n)) - set(members.tolist()))
        u, cnt = np.unique(members, return_counts=True)
        print("step", s, "lists are not a permutation: listed", counts[0], "heavy", counts[1], "missing", missing, "duplicates", u[cnt > 1].tolist())
        print(" costs before of missing", cost_before[missing], "now", gpu.move_cost()[missing])
        order = lists[:counts[0]]
        print(" order costs(before):", cost_before[order].tolist())
        hb = np.bincount(np.clip(cost_before >> 7, 0, 31), minlength=32)
        print(" host hist by bucket (desc):", hb[::-1].tolist())
        break
    fields = []
    for key in ("bodies", "controllers"):
        for f in g[key].dtype.names:
            if f == "_pad":
                continue
            d = (g[key][f] != c[key][f]).reshape(n, -1).any(1)
            if d.any():
                fields.append((key, f, np.nonzero(d)[0].tolist()))
    if fields:
        print("step", s)
        for key, f, who in fields:
            k0 = who[0]
            print("  %s.%s differs for %s: gpu %s cpu %s (before %s)" % (key, f, who, g[key][f][k0], c[key][f][k0], before[key][f][k0]))
        print("cost", gpu.move_cost()[who[0]])
        break
else:
    print("no difference")
