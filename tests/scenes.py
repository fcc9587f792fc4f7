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


EPS32 = float(np.finfo(np.float32).eps)


def assert_close(got, ref, name, rel=1e-5, floor=1e-2, group=None, ulps=16):
    """The float bar of the path (BASELINE.json north_star: 1e-5 relative), in two forms that must both hold:
      max-norm      max|got - ref| <= rel * max|ref|
      per element   |got - ref| <= rel * max(|ref|, floor * max|ref|) + ulps * eps32 * mag      for EVERY element
    The second keeps a small component of a large buffer honest: a vertex coordinate 0.05 units from the origin on a crowd that
    spans 100 units may be off by 1e-5 * 1 unit, not by 1e-5 * 100. The last term is what float32 itself takes: a palette entry is
    the end of a chain of up to a dozen 4x4 products whose terms have the magnitude `mag` of the character (a rotation error of one
    rounding times the lever arm, per bone of the chain; OCML sinf / cosf against libm at its head), a skinned coordinate adds four
    weighted transforms with fused multiply-adds where the oracle has none — a component that cancels to ~0 carries those
    roundings whatever its own size. Measured on the GPU over every parity test: at most 7 eps32 * mag; the bar allows 16.
    `group` = elements per character (the buffer is [characters][group]): mag is then that character's own max|ref|, not the
    crowd's; without it mag = max|ref| of the whole buffer."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    if ref.size == 0:
        return
    a = np.abs(ref).astype(np.float64)
    scale = float(a.max())
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    assert err.max() <= rel * scale, (name, "max-norm", float(err.max()), scale)
    if group and ref.size % group == 0:
        mag = np.broadcast_to(a.reshape(-1, group).max(axis=1)[:, None], (ref.size // group, group)).reshape(ref.shape)
    else:
        mag = scale
    tol = rel * np.maximum(a, floor * scale) + ulps * EPS32 * mag
    bad = err > tol
    if bad.any():
        i = np.unravel_index(int(np.argmax(err / tol)), err.shape)
        raise AssertionError(f"{name}: {int(bad.sum())} of {ref.size} elements beyond 1e-5 * max(|ref|, {floor} * {scale:.4g}) + {ulps} ulp; worst at {i}: "
                             f"got {got[i]!r} ref {ref[i]!r} |d| {err[i]:.3g} tol {tol[i]:.3g}")


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
    assert_close(gp, cp, "palette", group=gp.shape[1] * 16)  # mag = the character's own largest palette entry


# ---- kinematic platform scene (dynamic triangle set + PlatformCarry) ---------------------------------------------

def box_mesh(hx, hy, hz):
    """12-triangle box centred on the origin (outward winding)."""
    p = np.array([[-hx, -hy, -hz], [hx, -hy, -hz], [hx, hy, -hz], [-hx, hy, -hz],
                  [-hx, -hy, hz], [hx, -hy, hz], [hx, hy, hz], [-hx, hy, hz]], np.float32)
    i = np.array([0, 2, 1, 0, 3, 2, 4, 5, 6, 4, 6, 7, 0, 1, 5, 0, 5, 4, 3, 6, 2, 3, 7, 6, 0, 4, 7, 0, 7, 3, 1, 2, 6, 1, 6, 5], np.uint32)
    return p, i


def translation_matrix(t):
    m = np.eye(4, dtype=np.float32)
    m[3, :3] = t
    return m.reshape(16)


class PlatformScene:
    """Ground quad (static set) + kinematic box platforms (dynamic set) that move along a velocity each step, driven the
    way the reference orders its systems: KinematicPlatformMotionSystem (Systems.swift:122-155) moves the bodies,
    CollisionQueryRefreshSystem (:157-180) re-poses the dynamic set, then the character systems run."""

    def __init__(self, pkg, eng, starts, velocities, half=(6.0, 0.5, 6.0), dt=1.0 / 60.0):
        self.pkg, self.eng, self.dt = pkg, eng, dt
        self.pos = np.asarray(starts, np.float64).reshape(-1, 3).copy()
        self.vel = np.asarray(velocities, np.float64).reshape(-1, 3)
        self.box = box_mesh(*half)
        gp, gi, gm = pkg.assets.ground_plane()
        eng.rebuild_static([{"positions": gp, "indices": gi, "modelMatrix": gm, "material": (0.9, 0.8, 0)}])
        eng.rebuild_dynamic([{"positions": self.box[0], "indices": self.box[1], "modelMatrix": translation_matrix(p.astype(np.float32)),
                              "material": (0.6, 0.5, 0), "layer": 2} for p in self.pos])
        eng.upload_platforms(None)

    def step(self, stages=None):
        prev = self.pos.copy()
        self.pos = self.pos + self.vel * self.dt
        mats = np.stack([translation_matrix(p.astype(np.float32)) for p in self.pos])
        self.eng.update_transforms(self.pkg.abi.SET_DYNAMIC, np.arange(len(self.pos)), mats)
        pf = np.zeros(len(self.pos), self.pkg.abi.platform_dtype)
        for k in range(len(self.pos)):
            mn, mx = self.eng.mesh_world_aabb(self.box[0], mats[k])
            pf[k]["aabbMin"], pf[k]["aabbMax"] = mn, mx
            pf[k]["delta"] = self.pos[k].astype(np.float32) - prev[k].astype(np.float32)  # positionF - prevPositionF
            pf[k]["kinematic"], pf[k]["hasAABB"] = 1, 1
        self.eng.upload_platforms(pf)
        A = self.pkg.abi
        physics = A.STAGE_INTENT | A.STAGE_GRAVITY | A.STAGE_MOVE | A.STAGE_LOCOMOTION | A.STAGE_ACTION | A.STAGE_WRITEBACK
        self.eng.tick(dt=self.dt, stages=physics if stages is None else stages)


def spawn_on_platforms(pkg, eng, ybot, positions):
    n = len(positions)
    eng.resize(n)
    state = {"bodies": pkg.assets.default_bodies(n, np.asarray(positions, np.float64)), "params": pkg.assets.default_controller_params(n),
             "controllers": pkg.assets.default_controller_state(n), "intents": pkg.assets.default_intents(n),
             "locomotion": pkg.assets.default_locomotion(n, ybot), "actions": pkg.assets.default_actions(n, ybot, present=True)}
    eng.upload(**state)
    return state
