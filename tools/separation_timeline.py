"""Diagnostic (GPU box; variant library built by `tools/build_variant.sh septime -DSGE_SEP_TIMING=1`): the timeline of the last dataflow
pass of a step — when every loop drew its ticket, got its turn, had seen all its candidates pass, and ended (100-MHz clock).
usage: SGE_AMD_LIB=libsge_amd_septime.so python tools/separation_timeline.py [n]"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ybot = sge.assets.YBotAssets()
eng = sge.CharacterEngine(0)
sge.crowd.upload_character_assets(eng, ybot, rings=3, segments=3)
scene = sge.crowd.upload_asset_scene(eng, ("cheese", "mirror"), footprint=200.0)
sge.crowd.spawn_crowd(eng, ybot, n, scene, seed=43, agents=True, mixed=True)
st = (abi.STAGE_ALL & ~abi.STAGE_SKIN) | abi.STAGE_SEPARATION
for _ in range(32):
    eng.tick(stages=st)
eng.synchronize()
fn = eng.t.lib.sge_experiment_sep_timeline
fn.argtypes = [C.c_void_p, C.c_int]
ts = np.zeros((n, 8), np.uint64)
assert fn(ts.ctypes.data, n) == 0
extra = ts[:, 4:8].astype(np.int64)
ts = ts[:, :4]
t = (ts.astype(np.int64) - int(ts[:, 0].min())) * 0.01  # us
wall = t[:, 3].max()
print("pass: %.1f us for %d loops" % (wall, n))
act = t[:, 3] - t[:, 1]
p1 = t[:, 2] - t[:, 1]
p2 = t[:, 3] - t[:, 2]
print("own turn -> end: mean %.2f us (sum %.0f us = %.1f x the pass); candidates' turns: mean %.2f, median %.2f us; pairs: mean %.2f, median %.2f us"
      % (act.mean(), act.sum(), act.sum() / wall, p1.mean(), np.median(p1), p2.mean(), np.median(p2)))
# does loop i + 1 start when loop i ends? gap between a loop's end and the next loop's turn
order = np.argsort(t[:, 1])
print("turn order follows index order: %.1f %% of consecutive turns have a higher index" % (100 * np.mean(np.diff(order) > 0)))
gap = t[1:, 1] - t[:-1, 3]
print("turn(i+1) - end(i): mean %.2f us, median %.2f us; fraction within 3 us after: %.2f" % (gap.mean(), np.median(gap), np.mean((gap > -0.5) & (gap < 3))))
# how many loops are between turn and end at a time
ev = np.concatenate([np.stack([t[:, 1], np.ones(n)], 1), np.stack([t[:, 3], -np.ones(n)], 1)])
ev = ev[np.argsort(ev[:, 0])]
live = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0])
print("loops between their turn and their end, time-weighted mean: %.2f" % ((live[:-1] * dt).sum() / dt.sum()))
nh, rounds, fall, trips = extra.T
contacts, crawl, march = (rounds >> 8) & 255, rounds >> 16, fall >> 8
rounds, fall = rounds & 255, fall & 255
sw = trips > 0
print('sweeps: %d loops; per such loop: %.1f trips, %.1f march evaluations over all items (%.1f of them advancing by exactly minAdvance), %.2f contacts' % (sw.sum(), trips[sw].mean(), march[sw].mean(), crawl[sw].mean(), contacts[sw].mean()))
long = trips > 8
if long.any(): print('  loops of > 8 trips: %.1f trips, %.1f march evaluations (%.1f crawling), %.2f contacts' % (trips[long].mean(), march[long].mean(), crawl[long].mean(), contacts[long].mean()))
print("pairs phase by the loop's work (mean us, share of the summed time):")
tot = p2.sum()
for name, sel in (("no changing pair", nh == 0), ("pairs, no cast needed", (nh > 0) & (rounds == 0)), ("rounds, no item", (rounds > 0) & (trips == 0) & (fall == 0)), ("items swept, <= 2 trips", (trips > 0) & (trips <= 2) & (fall == 0)),
                  ("items swept, 3-8 trips", (trips > 2) & (trips <= 8) & (fall == 0)), ("items swept, > 8 trips", (trips > 8) & (fall == 0)), ("a pair through the BVH casts", fall > 0)):
    if sel.any():
        print("  %-30s %5d loops  %7.2f us  %5.1f %%" % (name, sel.sum(), p2[sel].mean(), 100 * p2[sel].sum() / tot))
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "sep_timeline_%d.npy" % n), np.concatenate([t, extra], 1).astype(np.float32))
