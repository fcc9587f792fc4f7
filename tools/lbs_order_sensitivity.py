"""Diagnostic: how sensitive is the LBS kernel to vertex order (bone-set coherence inside a wavefront)?"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
ybot = sge.assets.YBotAssets()
for mode in ("coherent", "shuffled-64", "shuffled-all"):
    eng = sge.CharacterEngine(0)
    built, mesh = sge.crowd.upload_character_assets(eng, ybot)
    m = {k: v for k, v in eng.mesh.items()}
    V = m["positions"].shape[0]
    rng = np.random.default_rng(1)
    if mode == "shuffled-all":
        perm = rng.permutation(V)
    elif mode == "shuffled-64":  # shuffle blocks of 64 vertices (wave-sized runs stay coherent)
        perm = (rng.permutation(V // 64)[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
    else:
        perm = np.arange(V)
    m2 = {k: (np.ascontiguousarray(v[perm]) if k in ("positions", "normals", "tangents", "uvs", "boneIndices", "boneWeights") else v) for k, v in m.items()}
    eng.upload_skinned_mesh(m2)
    n = 10000
    sge.crowd.spawn_crowd(eng, ybot, n, mode="lbs")
    st = abi.STAGE_POSE | abi.STAGE_SKIN
    for _ in range(5):
        eng.tick(stages=st)
    eng.set_option(abi.OPT_PROFILE, 1)
    eng.profile_read(reset=True)
    for _ in range(30):
        eng.tick(stages=st)
    p = eng.profile_read(reset=True)
    print(mode, "lbs ms:", p.skin_ms / 30)
    eng.close()
