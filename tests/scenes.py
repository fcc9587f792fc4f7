"""TEST INFRASTRUCTURE: identical scene construction for the HIP product and the CPU oracle,
and the comparison rules (bit-exact for the CCD state, 1e-5 relative for float pose/skin output)."""
import importlib

import numpy as np


def build_scene(pkg, eng, n, terrain_cells=(48, 32), seed=1234, mode="ccd", agents=False, mixed=False,
                rings=6, segments=6, pose_debug=True, mesh_inv_bind=False, real_mesh=False, asset_scene=None,
                footprint=200.0):
    """real_mesh: the FBX-derived Y-Bot (tests/golden/ybot_skinned.npz) instead of the synthetic tube mesh;
    asset_scene: tuple of static asset names ("cheese", "mirror", "semla") instead of the synthetic terrain."""
    ybot = pkg.assets.YBotAssets()
    if pose_debug:
        eng.set_option(pkg.abi.OPT_STORE_POSE_DEBUG, 1)
    if real_mesh:
        pkg.crowd.upload_ybot_mesh(eng, ybot)
    else:
        pkg.crowd.upload_character_assets(eng, ybot, rings=rings, segments=segments, mesh_inv_bind=mesh_inv_bind)
    if asset_scene:
        terrain = pkg.crowd.upload_asset_scene(eng, asset_scene, footprint=footprint)
        state = pkg.crowd.spawn_crowd(eng, ybot, n, terrain, seed=seed, mode=mode, agents=agents, mixed=mixed)
        return ybot, terrain, state
    terrain = pkg.crowd.upload_terrain(eng, cells=terrain_cells) if terrain_cells else None
    if terrain is None:
        pkg.crowd.upload_ground_plane(eng)
        terrain = {"half": (40.0, 40.0)}
    state = pkg.crowd.spawn_crowd(eng, ybot, n, terrain, seed=seed, mode=mode, agents=agents, mixed=mixed) if terrain_cells or mode == "lbs" else None
    return ybot, terrain, state


def assert_struct_equal(a, b, name, skip=("_pad",)):
    for f in a.dtype.names:
        if f in skip:
            continue
        x, y = a[f], b[f]
        if x.dtype.kind == "f":
            same = (x.view(np.dtype("u%d" % x.dtype.itemsize)) == y.view(np.dtype("u%d" % y.dtype.itemsize))) | ((x == 0) & (y == 0))
        else:
            same = x == y
        if not np.all(same):
            bad = np.argwhere(~same)[:5]
            raise AssertionError(f"{name}.{f} differs at {bad.tolist()}: gpu={x[tuple(bad[0])]!r} cpu={y[tuple(bad[0])]!r}")


def compare_states(pkg, gpu, cpu, n, float_tol_fields=("posePhase", "time", "motionTime", "blendT", "idleInertia", "weight")):
    g, c = gpu.download(), cpu.download()
    # CCD state: bit-exact (integer/branch decisions and IEEE float32/float64 arithmetic in the oracle's order)
    assert_struct_equal(g["bodies"], c["bodies"], "bodies")
    assert_struct_equal(g["controllers"], c["controllers"], "controllers")
    # locomotion/action clocks go through powf/fmodf: discrete fields exact, float fields to 1e-6
    for key in ("locomotion", "actions"):
        for f in g[key].dtype.names:
            x, y = g[key][f], c[key][f]
            if f in float_tol_fields:
                assert np.allclose(x, y, rtol=1e-6, atol=1e-7), (key, f)
            elif f != "_pad":
                assert np.array_equal(x, y), (key, f, x[:4], y[:4])
    gp, _, _ = gpu.palettes()
    cp, _, _ = cpu.palettes()
    scale = np.abs(cp).max()
    assert np.abs(gp - cp).max() <= 1e-5 * scale, ("palette", np.abs(gp - cp).max(), scale)
