"""Diagnostic: per-character cost of the move stage (distance evaluations of the last step, sge_move_cost_read) on a
settled 10k crowd, the most expensive characters' state, and each one's step replayed alone with the query counters."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
which = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("cheese",)
eng = sge.CharacterEngine(0)
ybot = sge.assets.YBotAssets()
sge.crowd.upload_character_assets(eng, ybot, rings=3, segments=3)
scene = sge.crowd.upload_terrain(eng) if which == ("synthetic",) else sge.crowd.upload_asset_scene(eng, which)
n = 10000
sge.crowd.spawn_crowd(eng, ybot, n, scene)
st = abi.STAGE_INTENT | abi.STAGE_GRAVITY | abi.STAGE_MOVE | abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_WRITEBACK
for _ in range(140):
    eng.tick(stages=st)
eng.synchronize()
for step in range(6):
    before = eng.download()
    eng.move_stats(reset=True)
    eng.set_option(abi.OPT_PROFILE, 1)
    eng.profile_read(reset=True)
    eng.tick(stages=st)
    eng.synchronize()
    ms = eng.profile_read(reset=True).move_ms
    s = eng.move_stats(reset=True)
    cost = eng.move_cost()
    q = np.percentile(cost, [50, 90, 99, 99.9])
    print("step %d: move %.3f ms; evals/char median %d p90 %d p99 %d p99.9 %d max %d; > 4000: %d; sum %d; pruned pairs %d; trips %d steps %d" % (
        step, ms, q[0], q[1], q[2], q[3], cost.max(), (cost > 4000).sum(), cost.sum(), s.prunedPairs, s.sweepTrips, s.traversalSteps))
top = np.argsort(-cost)[:8]
after = eng.download()
for k in top:
    # replay character k's last step alone
    eng.upload(first=int(k), **{key: v[k:k + 1] for key, v in before.items()})
    eng.move_stats(reset=True)
    eng.profile_read(reset=True)
    eng.tick(stages=st, first=int(k), count=1)
    eng.synchronize()
    ms = eng.profile_read(reset=True).move_ms
    s = eng.move_stats(reset=True)
    b, c = before["bodies"][k], before["controllers"][k]
    print("char %5d cost %6d alone %.3f ms: queries %d cand %d evals %d pruned %d steps %d trips %d | pos %s vel %s flags %x sideFrames %d gdist %.2f" % (
        k, cost[k], ms, s.queries, s.candidates, s.sweepIterations, s.prunedPairs, s.traversalSteps, s.sweepTrips,
        np.round(b["position"], 2), np.round(b["linearVelocity"], 2), c["flags"], c["sideContactFrames"], c["groundDistance"]))
eng.close()
