"""Diagnostic (SGE_WAVE_PROF=1): in-kernel cycle stamps (s_memtime, shader cycles) of move_group_kernel (per wavefront),
move_kernel<0> and pose_kernel (per character) on a settled 10k crowd, in one of three settings:
  alone     the collision + pose kernels with nothing beside them (no skin stage)
  overlap   the shipped default step: skin(n) on its own stream beside move(n+1) + pose(n+1), LBS workgroups handing places over
  resident  the same with the LBS launch as resident workgroups (SGE_SKIN_PERSISTENT quarters of a workgroup per CU, default 8)
usage: wave_prof.py [scene[,scene]] [alone|overlap|resident] [quarters]"""
import importlib, os, sys
import numpy as np
os.environ["SGE_WAVE_PROF"] = "1"
which = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("cheese",)
mode = sys.argv[2] if len(sys.argv) > 2 else "alone"
if mode == "resident":
    os.environ["SGE_SKIN_PERSISTENT"] = sys.argv[3] if len(sys.argv) > 3 else "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
G = 4
overlap = mode != "alone"
eng = sge.CharacterEngine(0)
ybot = sge.assets.YBotAssets()
eng.set_option(abi.OPT_PROFILE, 1)
if overlap:
    eng.set_option(abi.OPT_OVERLAP_SKIN, 1)
sge.crowd.upload_character_assets(eng, ybot)  # the bench's 14,080-vertex crowd mesh in every setting: same LDS / register footprints
scene = sge.crowd.upload_terrain(eng) if which == ("synthetic",) else sge.crowd.upload_asset_scene(eng, which)
n = 10000
sge.crowd.spawn_crowd(eng, ybot, n, scene)
st = abi.STAGE_ALL if overlap else (abi.STAGE_ALL & ~abi.STAGE_SKIN)
for _ in range(150):
    eng.tick(stages=st)
eng.synchronize()
eng.profile_read(reset=True)
for _ in range(20):
    eng.tick(stages=st)
eng.synchronize()
tm = eng.profile_read(reset=True)
times = {k: round(getattr(tm, k) / max(getattr(tm, k.replace('_ms', '_launches')), 1), 4) for k, _ in tm._fields_ if k.endswith('_ms')}
prof = np.zeros((3 * n, 8), np.uint64)
rc = eng.t.lib.sge_debug_wave_profile(eng.h, abi.ptr(prof), 3 * n)
assert rc == 0, rc
p = prof.astype(np.float64)
waves = (n + G - 1) // G
grp, mv0, pose = p[:n], p[n:2 * n], p[2 * n:3 * n]  # (the grouped launch has fewer wavefronts than characters: unused rows stay zero)
grp = grp[grp[:, 0] > 0]


def pct(a):
    return "median %7.0f  p90 %7.0f  p99 %7.0f  max %8.0f  sum %.4g" % (np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max(), a.sum())


print("== setting: %s, scene %s, %d characters; stage times per step (HIP events): %s" % (mode, "+".join(which), n, times))
tot = grp[:, 0]
print("move_group_kernel: %d wavefronts (the first ones one character each, then 4 each)" % len(grp))
print("  total cycles     %s" % pct(tot))
print("  traverse (gather)%s  share %.3f" % (pct(grp[:, 1]), grp[:, 1].sum() / tot.sum()))
print("  sweep            %s  share %.3f" % (pct(grp[:, 2]), grp[:, 2].sum() / tot.sum()))
print("  consume          %s  share %.3f" % (pct(grp[:, 3]), grp[:, 3].sum() / tot.sum()))
print("  rounds median %.0f max %.0f; sweep trips median %.0f max %.0f; traversal steps median %.0f max %.0f" % (
    np.median(grp[:, 4]), grp[:, 4].max(), np.median(grp[:, 5]), grp[:, 5].max(), np.median(grp[:, 6]), grp[:, 6].max()))
print("  cycles per traversal step %.0f; cycles per sweep trip %.0f" % (grp[:, 1].sum() / max(grp[:, 6].sum(), 1), grp[:, 2].sum() / max(grp[:, 5].sum(), 1)))
t0 = grp[:, 7].min()
end = grp[:, 7] - t0 + tot
print("  launch span %.0f cycles; last wavefront starts at %.0f" % (end.max(), (grp[:, 7] - t0).max()))
m = mv0[mv0[:, 0] > 0]
if len(m):
    print("move_kernel<0>: %d characters" % len(m))
    print("  total cycles     %s" % pct(m[:, 0]))
    print("  overlap queries  %s  share %.3f" % (pct(m[:, 1]), m[:, 1].sum() / m[:, 0].sum()))
    print("  cycles per traversal step %.0f (steps median %.0f)" % (m[:, 1].sum() / max(m[:, 2].sum(), 1), np.median(m[:, 2])))
    print("  launch span %.0f cycles" % ((m[:, 7] - m[:, 7].min() + m[:, 0]).max()))
q = pose[pose[:, 0] > 0]
if len(q):
    print("pose_kernel: %d characters" % len(q))
    print("  total cycles     %s" % pct(q[:, 0]))
    for k, name in enumerate(["state machines  ", "bone locals     ", "action/align/lean", "model products  ", "palette         "], 1):
        print("  %s %s  share %.3f" % (name, pct(q[:, k]), q[:, k].sum() / q[:, 0].sum()))
    print("  launch span %.0f cycles" % ((q[:, 7] - q[:, 7].min() + q[:, 0]).max()))
eng.close()
