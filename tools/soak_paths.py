"""Soak (GPU box): the three schedules of the move stage (default / everybody in the multi-wave launch / nobody in it) over many
steps of the full 10k crowd on one scene; every state array must stay identical bit for bit. Usage: soak_paths.py [scene] [steps]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
which = sys.argv[1] if len(sys.argv) > 1 else "cheese"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
ybot = sge.assets.YBotAssets()
st = abi.STAGE_ALL & ~abi.STAGE_SKIN


def run(threshold, mixed):
    eng = sge.CharacterEngine(0)
    sge.crowd.upload_character_assets(eng, ybot, rings=3, segments=3)
    scene = sge.crowd.upload_terrain(eng) if which == "synthetic" else sge.crowd.upload_asset_scene(eng, tuple(which.split(",")))
    eng.set_option(abi.OPT_HEAVY_THRESHOLD, threshold)
    sge.crowd.spawn_crowd(eng, ybot, 10000, scene, mixed=mixed)
    snaps = []
    for s in range(steps):
        eng.tick(stages=st)
        if s in (steps // 3, 2 * steps // 3, steps - 1):
            eng.synchronize()
            snaps.append(eng.download())
    ov = eng.move_stats().overflow
    eng.close()
    return snaps, ov


for mixed in (False, True):
    base, ov = run(4000, mixed)
    assert ov == 0
    for threshold in (0, -1):
        other, ov2 = run(threshold, mixed)
        assert ov2 == 0
        for k, (a, b) in enumerate(zip(base, other)):
            for name in a:
                if not np.array_equal(a[name].view(np.uint8), b[name].view(np.uint8)):
                    bad = np.flatnonzero((a[name].view(np.uint8).reshape(len(a[name]), -1) != b[name].view(np.uint8).reshape(len(b[name]), -1)).any(axis=1))
                    raise SystemExit("MISMATCH scene %s mixed %s threshold %d snapshot %d array %s: characters %s" % (which, mixed, threshold, k, name, bad[:10]))
        print("scene %s mixed=%s: threshold %d agrees with the default over %d steps" % (which, mixed, threshold, steps), flush=True)
print("soak ok")
