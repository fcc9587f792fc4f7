"""AgentSeparationSystem (SGE_STAGE_SEPARATION) on crowds: ms per fixed step of the stage (a step with the stage minus a step without
it, both synchronised) for several crowd sizes on the cheese + mirror scene, whether the dataflow passes ran to the end or were redone
serially, and the depth of the dependency graph the dataflow had to respect (longest chain of loops that share an agent, computed on the
host from the positions at the head of the step: what bounds the stage however many wavefronts the chip offers).
usage: separation_bench.py [--sweeps] [--footprint F] [n ...]   (--sweeps: character-vs-character sweeps in the move stage as in configs[4], so that the
stage meets the shallow overlaps it is meant for instead of a crowd spawned on top of itself)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
sweeps = "--sweeps" in sys.argv
footprint = 200.0   # side of every prop of the scene: the crowd is spawned over the props' bounds
args = [a for a in sys.argv[1:] if a != "--sweeps"]
if "--footprint" in args:
    k = args.index("--footprint")
    footprint = float(args[k + 1])
    del args[k:k + 2]
sizes = [int(a) for a in args] or [192, 1024, 4096, 8192, 31250]
ybot = sge.assets.YBotAssets()


def dag_depth(pos, radius=1.5, margin=0.2):
    """Longest chain in the dependency graph of one pass: loop k passes every agent c > k in the 5 x 5 cells around it."""
    cell = 2 * radius + margin
    n = len(pos)
    cx = np.floor(pos[:, 0] / cell).astype(np.int64)
    cz = np.floor(pos[:, 2] / cell).astype(np.int64)
    from collections import defaultdict
    cells = defaultdict(list)
    for i in range(n):
        cells[(cx[i], cz[i])].append(i)
    level = np.zeros(n, np.int32)   # level of loop i = 1 + max level of the earlier loops it must wait for
    for i in range(n):
        best = 0
        for dz in range(-2, 3):
            for dx in range(-2, 3):
                for k in cells.get((cx[i] + dx, cz[i] + dz), ()):
                    if k < i and level[k] > best:
                        best = level[k]
        level[i] = best + 1
    return int(level.max()), np.bincount(level)[1:]


for n in sizes:
    eng = sge.CharacterEngine(0)
    sge.crowd.upload_character_assets(eng, ybot, rings=3, segments=3)
    scene = sge.crowd.upload_asset_scene(eng, ("cheese", "mirror"), footprint=footprint)
    sge.crowd.spawn_crowd(eng, ybot, n, scene, seed=43, agents=True, mixed=True)
    base = abi.STAGE_ALL & ~abi.STAGE_SKIN
    if sweeps:
        import torch
        ex = sge.parallel.AgentExchange(eng, n, 0, 1, torch.device("cuda", 0), None)
        tick = lambda stages: ex.step(stages=stages)
    else:
        tick = lambda stages: eng.tick(stages=stages)
    for _ in range(60 if sweeps else 30):
        tick(base | abi.STAGE_SEPARATION)
        eng.synchronize()  # as a host that keeps its World in step does: the stage's reach for the next step follows from this step's flag
    eng.synchronize()
    pos = eng.download(what=("bodies",))["bodies"]["position"].astype(np.float32)
    depth, hist = dag_depth(pos)
    info = np.zeros(4, np.int32)
    redo = 0
    flags = 0
    t = {}
    for label, st in (("with", base | abi.STAGE_SEPARATION), ("without", base)):
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            tick(st)
            if label == "with":
                eng.t.lib.sge_debug_separation(eng.h, abi.ptr(info))  # (synchronises: every step is timed with its own drain, both arms alike)
                redo += int(info[2] != 0)
                flags |= int(info[2])
            else:
                eng.synchronize()
        eng.synchronize()
        t[label] = (time.perf_counter() - t0) / 20 * 1e3
    print(("sweeps on, " if sweeps else "") + ("footprint %.0f, " % footprint) + "n %6d: step with the stage %.3f ms, without %.3f ms -> stage %.3f ms; agents listed %d; steps redone serially %d / 20; "
          "(flags %d: 1 = too many neighbours to track, 2 = pushed further than a cell); dependency depth %d levels of %d loops (widest level %d loops)" % (n, t["with"], t["without"], t["with"] - t["without"], info[0], redo, flags, depth, n, hist.max()), flush=True)
    eng.close()
