"""Soak (GPU box): N steps (argv[1], default 400) of the full 10k mixed crowd on the cheese scene with SGE_OPT_OVERLAP_SKIN off and on: the state, every
palette and sampled skinned output must agree bit for bit (the two-palette-buffer schedule changes no result)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
ybot = sge.assets.YBotAssets()
def run(overlap, steps=400):
    eng = sge.CharacterEngine(0)
    eng.set_option(abi.OPT_OVERLAP_SKIN, overlap)
    sge.crowd.upload_character_assets(eng, ybot)
    scene = sge.crowd.upload_asset_scene(eng, ("cheese",))
    sge.crowd.spawn_crowd(eng, ybot, 10000, scene, mixed=True)
    for s in range(steps):
        eng.tick(stages=abi.STAGE_ALL)
    eng.synchronize()
    d = eng.download()
    pal = eng.palettes(0, 10000)[0]
    sk = [eng.skinned(i * eng.vertex_count, 512) for i in range(0, 10000, 53)]
    eng.close()
    return d, pal, sk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
a = run(0, N); b = run(1, N)
for k in a[0]:
    assert np.array_equal(a[0][k].view(np.uint8), b[0][k].view(np.uint8)), k
assert np.array_equal(a[1], b[1])
for x, y in zip(a[2], b[2]):
    for u, v in zip(x, y):
        assert np.array_equal(u, v)
print("overlap on/off agree bit for bit after %d steps: state, all palettes, sampled skinned output" % N)
