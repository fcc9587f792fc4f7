"""Diagnostic: the most expensive character of a settled crowd, its step's casts replayed one by one as single queries
(slide cast, snap cast, fall probe, the four offset casts) with the query counters of each."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
from importlib import import_module
E = import_module("swift-game-engine_amd.engine")
eng = sge.CharacterEngine(0)
ybot = sge.assets.YBotAssets()
sge.crowd.upload_character_assets(eng, ybot, rings=3, segments=3)
which = sys.argv[1] if len(sys.argv) > 1 else "cheese"
scene = sge.crowd.upload_terrain(eng) if which == "synthetic" else sge.crowd.upload_asset_scene(eng, tuple(which.split(",")))
n = 10000
sge.crowd.spawn_crowd(eng, ybot, n, scene)
st = abi.STAGE_INTENT | abi.STAGE_GRAVITY | abi.STAGE_MOVE | abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_WRITEBACK
for _ in range(141):
    eng.tick(stages=st)
eng.synchronize()
before = eng.download()
eng.tick(stages=st)
eng.synchronize()
cost = eng.move_cost()
order = np.argsort(-cost)
print("cost percentiles: median %d p90 %d p99 %d max %d" % tuple(np.percentile(cost, [50, 90, 99, 100])))
for k in list(order[:2]) + [order[len(order) // 2], order[len(order) // 10]]:
    b, p, c = before["bodies"][k], before["params"][k], before["controllers"][k]
    pos = np.array(b["position"], np.float32)
    vel = np.array(b["linearVelocity"], np.float32)
    print("char %d cost %d pos %s vel %s radius %.3f halfHeight %.3f snap %.3f fall %.1f minGroundDot %.3f flags %x" % (
        k, cost[k], np.round(pos, 3), np.round(vel, 3), p["radius"], p["halfHeight"], p["snapDistance"], p["fallProbeDistance"], p["minGroundDot"], c["flags"]))
    off = p["radius"] * 0.6
    casts = [("slide (velocity*dt, blocking)", pos, vel / 60.0, abi.CAST_BLOCKING),
             ("snap", pos, (0, -p["snapDistance"], 0), abi.CAST_GROUND), ("fall probe", pos, (0, -p["fallProbeDistance"], 0), abi.CAST_GROUND)]
    for name, d in (("+x", (off, 0, 0)), ("-x", (-off, 0, 0)), ("+z", (0, 0, off)), ("-z", (0, 0, -off))):
        casts.append(("offset " + name, pos + np.array(d, np.float32), (0, -p["snapDistance"], 0), abi.CAST_GROUND))
    for name, o, d, mode in casts:
        q = E.make_queries([o], [d], radius=p["radius"], half_height=p["halfHeight"], mode=mode, min_normal_y=p["minGroundDot"], mask=int(p["collisionMask"]))
        eng.move_stats(reset=True)
        h = eng.capsule_cast(q)[0]
        s = eng.move_stats(reset=True)
        print("   %-30s hit %d toi %.4f tri %6d n.y %.3f | candidates %5d evals %6d pruned %5d steps %4d trips %4d" % (
            name, h["hit"], h["toi"], h["triangleIndex"], h["triangleNormal"][1], s.candidates, s.sweepIterations, s.prunedPairs, s.traversalSteps, s.sweepTrips))
eng.close()
