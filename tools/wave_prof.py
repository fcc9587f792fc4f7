"""Diagnostic (SGE_WAVE_PROF=1): per-wavefront cycles of move_group_kernel on a settled 10k crowd: distribution, phase split,
and what the slowest wavefronts hold."""
import importlib, os, sys
import numpy as np
os.environ["SGE_WAVE_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
which = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("cheese",)
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
overlap = len(sys.argv) > 3 and sys.argv[3] == "overlap"  # the whole default step: the grouped kernel beside the LBS launch
eng = sge.CharacterEngine(0)
ybot = sge.assets.YBotAssets()
if overlap:
    eng.set_option(abi.OPT_OVERLAP_SKIN, 1)
    sge.crowd.upload_character_assets(eng, ybot)
else:
    sge.crowd.upload_character_assets(eng, ybot, rings=3, segments=3)
scene = sge.crowd.upload_terrain(eng) if which == ("synthetic",) else sge.crowd.upload_asset_scene(eng, which)
n = 10000
sge.crowd.spawn_crowd(eng, ybot, n, scene)
st = abi.STAGE_ALL if overlap else (abi.STAGE_INTENT | abi.STAGE_GRAVITY | abi.STAGE_MOVE | abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_WRITEBACK)
for _ in range(150):
    eng.tick(stages=st)
eng.synchronize()
waves = (n + G - 1) // G
prof = np.zeros((waves, 8), np.uint64)
import ctypes as C
rc = eng.t.lib.sge_debug_wave_profile(eng.h, abi.ptr(prof), waves)
assert rc == 0, rc
cost = eng.move_cost()
p = prof.astype(np.float64)
tot = p[:, 0]
print("wavefronts %d; cycles total: median %.0f p90 %.0f p99 %.0f max %.0f (100 MHz memtime ticks? compare with kernel ms)" % (waves, np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max()))
print("share of summed cycles: gather %.2f sweep %.2f consume %.2f" % (p[:, 1].sum() / tot.sum(), p[:, 2].sum() / tot.sum(), p[:, 3].sum() / tot.sum()))
print("rounds median %.0f max %.0f; trips median %.0f max %.0f; steps median %.0f max %.0f" % (np.median(p[:, 4]), p[:, 4].max(), np.median(p[:, 5]), p[:, 5].max(), np.median(p[:, 6]), p[:, 6].max()))
t0 = p[:, 7].min()
end = (p[:, 7] - t0 + tot)
print("launch span %.0f ticks; last wave starts at %.0f" % (end.max(), (p[:, 7] - t0).max()))
hist = np.percentile(end, [50, 90, 99, 100])
print("wave END times after the first start: median %.0f p90 %.0f p99 %.0f max %.0f" % tuple(hist))
for w in np.argsort(-end)[:8]:
    print("wave %5d end %.0f total %.0f gather %.0f sweep %.0f consume %.0f rounds %d trips %d steps %d start %.0f costs(consecutive, not members) %s" % (
        w, end[w], tot[w], p[w, 1], p[w, 2], p[w, 3], p[w, 4], p[w, 5], p[w, 6], p[w, 7] - t0, cost[w * G:(w + 1) * G].tolist()))
eng.close()
