"""CPU tests: the oracle against the committed golden vectors and analytic known answers.

pose_chain_f64.npz was produced by functions IMPORTED from the reference's own
Tools/FitMotion/fit_motion.py (tests/golden/make_golden.py); the reference has no tests or
golden vectors of its own for this path (GameTests/GameTests.swift:12-16 is empty)."""
import os

import numpy as np
import pytest

import oracle_binding as ob
from scenes import build_scene


@pytest.fixture(scope="module")
def golden(sge):
    return np.load(os.path.join(sge.assets.GOLDEN_DIR, "pose_chain_f64.npz"))


@pytest.fixture()
def cpu():
    e = ob.oracle_engine()
    yield e
    e.close()


def test_skeleton_build_matches_reference_tool(sge, ybot, cpu, golden):
    built = cpu.upload_skeleton(ybot)
    assert ybot.bone_count == 65 and ybot.pelvis_index == 0 and ybot.lean_index == ybot.names.index("mixamorig:Spine2")
    assert ybot.zero_root and abs(ybot.unit_scale - 0.026) < 1e-9
    assert np.abs(built["bindLocal"] - golden["bind.local"]).max() < 2e-6
    B = ybot.bone_count
    model = golden["bind.model"].reshape(B, 4, 4)
    inv = built["invBindModel"].reshape(B, 4, 4).astype(np.float64)
    for i in range(B):  # stored [col][row]
        assert np.abs(model[i].T @ inv[i].T - np.eye(4)).max() < 5e-6


@pytest.mark.parametrize("profile", ["Idle", "Walking", "Running", "FallingIdle", "StandingDodgeBackward"])
def test_single_profile_pose_matches_reference_tool(sge, ybot, cpu, golden, profile):
    abi = sge.abi
    cpu.upload_skeleton(ybot)
    cpu.upload_profiles(ybot.profiles)
    phases = golden["phases"]
    n = len(phases)
    cpu.resize(n)
    k = ybot.profile_index(profile)
    L = sge.assets.default_locomotion(n, ybot)
    L["flags"] = abi.MOTION_PRESENT | abi.MOTION_LOOP | abi.MOTION_IN_PLACE
    L["motionProfile"] = k
    L["motionTime"] = phases.astype(np.float32) * np.float32(ybot.profiles[k]["cycleDuration"])
    cpu.upload(bodies=sge.assets.default_bodies(n, np.zeros((n, 3))), params=sge.assets.default_controller_params(n),
               controllers=sge.assets.default_controller_state(n), intents=sge.assets.default_intents(n),
               locomotion=L, actions=sge.assets.default_actions(n))
    cpu.tick(dt=0.0, stages=abi.STAGE_POSE)
    pal, mod, loc = cpu.palettes(model=True, local=True)
    scale = np.abs(golden[f"{profile}.model"]).max()
    assert np.abs(loc - golden[f"{profile}.local"]).max() <= 1e-5 * scale
    assert np.abs(mod - golden[f"{profile}.model"]).max() <= 1e-5 * scale
    # foot heights come out of the reference's compute_foot_contacts itself
    lf, rf = ybot.names.index("mixamorig:LeftFoot"), ybot.names.index("mixamorig:RightFoot")
    assert np.abs(mod[:, lf, 13] - golden[f"{profile}.footAuxLeft"]).max() <= 1e-5 * scale
    assert np.abs(mod[:, rf, 13] - golden[f"{profile}.footAuxRight"]).max() <= 1e-5 * scale
    # palette = model * invBindModel
    inv = cpu.skeleton["invBindModel"].reshape(-1, 4, 4).astype(np.float64)
    m = mod.reshape(n, -1, 4, 4).astype(np.float64)
    ref = np.einsum("nbkr,bck->nbcr", m, inv).reshape(n, -1, 16)
    assert np.abs(pal - ref).max() < 5e-6


def test_bind_pose_skinning_is_identity(sge, ybot, cpu):
    build_scene(sge, cpu, 2, terrain_cells=(12, 10), rings=5, segments=5)
    L = sge.assets.default_locomotion(2, ybot)
    L["flags"] = 0
    cpu.upload(locomotion=L, actions=sge.assets.default_actions(2))
    cpu.tick(dt=0.0, stages=sge.abi.STAGE_POSE | sge.abi.STAGE_SKIN)
    pal, _, _ = cpu.palettes()
    assert np.abs(pal - np.eye(4, dtype=np.float32).reshape(16)).max() < 2e-6
    p, nrm, tan = cpu.skinned()
    V = cpu.vertex_count
    assert np.abs(p[:V] - cpu.mesh["positions"]).max() < 5e-6
    assert np.abs(nrm[:V] - cpu.mesh["normals"] / np.linalg.norm(cpu.mesh["normals"], axis=1, keepdims=True)).max() < 5e-6
    assert np.array_equal(tan[:V, 3], cpu.mesh["tangents"][:, 3])


def test_skinning_zero_weight_influences_are_skipped(sge, ybot, cpu):
    """skinningKernel skips influences with weight <= 0 (RayTracing.metalinc:758-761): garbage indices there are harmless."""
    built, mesh = sge.crowd.upload_character_assets(cpu, ybot, rings=4, segments=4)
    m2 = {k: v.copy() for k, v in cpu.mesh.items()}
    zero = m2["boneWeights"] <= 0
    m2["boneIndices"][zero] = 64
    cpu.resize(1)
    sge.crowd.upload_terrain(cpu, cells=(8, 8))
    st = sge.crowd.spawn_crowd(cpu, ybot, 1, mode="lbs")
    cpu.tick(stages=sge.abi.STAGE_POSE | sge.abi.STAGE_SKIN)
    a = cpu.skinned()
    cpu.upload_skinned_mesh(m2)
    cpu.resize(1)
    sge.crowd.spawn_crowd(cpu, ybot, 1, mode="lbs")
    cpu.tick(stages=sge.abi.STAGE_POSE | sge.abi.STAGE_SKIN)
    b = cpu.skinned()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_capsule_vs_plane_analytic(sge, cpu):
    sge.crowd.upload_ground_plane(cpu)
    hs = np.array([0.0, 0.5, 3.0, 7.5, 150.0], np.float32)
    origin = np.stack([np.zeros_like(hs), -3 + 2.5 + hs, np.zeros_like(hs)], -1)
    q = sge.make_queries(origin, np.tile([[0, -200.0, 0]], (len(hs), 1)), mode=sge.abi.CAST_GROUND)
    h = cpu.capsule_cast(q)
    assert h["hit"].all()
    # TOI = gap, up to the 1e-5 contact epsilon / minAdvance bracket and 10 bisection steps
    assert np.abs(h["toi"] - hs).max() < 0.03 / 1024 + 2e-5 * 200
    assert np.allclose(h["normal"], [0, 1, 0], atol=1e-6) and np.allclose(h["triangleNormal"], [0, 1, 0], atol=1e-6)
    assert np.allclose(h["position"][:, 1], -3, atol=1e-5)
    assert h["material"]["muS"][0] == np.float32(0.9)
    # moving up never hits; zero-length cast returns nil (CollisionQuery.swift:987-988)
    q2 = sge.make_queries(origin, np.tile([[0, 5.0, 0]], (len(hs), 1)))
    assert not cpu.capsule_cast(q2)["hit"][1:].any()
    assert not cpu.capsule_cast(sge.make_queries(origin, np.zeros((len(hs), 3))))["hit"].any()
    # blocking filter rejects a hit whose normal does not oppose the motion
    q3 = sge.make_queries([[0, -1.0, 0]], [[0, 3.0, 0]], mode=sge.abi.CAST_BLOCKING)
    assert cpu.capsule_cast(q3)["hit"][0] == 0
    # overlap depth = r - dist
    o, c = cpu.capsule_overlap_all(sge.make_queries([[0, -3 + 2.5 - 0.4, 0]]), 8)
    assert c[0] == 2 and np.allclose(o["depth"][0, :2], 0.4, atol=1e-5)
    d, f = cpu.capsule_overlap(sge.make_queries([[0, -3 + 2.5 - 0.4, 0], [0, 5, 0]]))
    assert f.tolist() == [1, 0] and abs(d["depth"][0] - 0.4) < 1e-5 and d["triangleIndex"][0] == o["triangleIndex"][0, 0]
    # layer mask
    qm = sge.make_queries(origin, np.tile([[0, -200.0, 0]], (len(hs), 1)), mask=2)
    assert not cpu.capsule_cast(qm)["hit"].any()


def test_bvh_invariants_and_bruteforce(sge, cpu):
    terrain = sge.crowd.upload_terrain(cpu, cells=(9, 7))
    col = cpu.collision_copy()
    nodes, order, leaf = col["nodes"], col["triOrder"], col["triLeaf"]
    T = len(order)
    assert T == 9 * 7 * 2 and sorted(order.tolist()) == list(range(T))
    leaves = np.flatnonzero(nodes["left"] < 0)
    assert nodes["count"][leaves].max() <= 4 and nodes["count"][leaves].sum() == T
    for li in leaves:
        s, c = nodes["start"][li], nodes["count"][li]
        assert (leaf[order[s:s + c]] == li).all()
        bb = col["aabbs"][order[s:s + c]]
        assert np.array_equal(nodes["boundsMin"][li], bb[:, 0].min(0)) and np.array_equal(nodes["boundsMax"][li], bb[:, 1].max(0))
    inner = np.flatnonzero(nodes["left"] >= 0)
    for ni in inner:
        l, r = nodes["left"][ni], nodes["right"][ni]
        assert nodes["parent"][l] == ni and nodes["parent"][r] == ni and l == ni + 1  # pre-order
        assert np.array_equal(nodes["boundsMin"][ni], np.minimum(nodes["boundsMin"][l], nodes["boundsMin"][r]))
    # brute force: the BVH answer equals the minimum over single-triangle worlds
    rng = np.random.default_rng(2)
    nq = 24
    x, z = rng.uniform(-3.5, 3.5, nq), rng.uniform(-2.5, 2.5, nq)
    y = sge.assets.terrain_height(x, z, *terrain["half"]) + rng.uniform(2.6, 5, nq)
    q = sge.make_queries(np.stack([x, y, z], -1), rng.normal(0, 2.0, (nq, 3)) + [0, -2, 0], radius=0.8, half_height=0.5)
    full = cpu.capsule_cast(q)
    best = np.full(nq, np.inf)
    single = ob.oracle_engine()
    pos, idx = terrain["positions"], terrain["indices"].reshape(-1, 3)
    for t in range(len(idx)):
        single.rebuild_static([{"positions": pos, "indices": idx[t]}])
        h = single.capsule_cast(q)
        best = np.where(h["hit"] == 1, np.minimum(best, h["toi"]), best)
    single.close()
    assert np.array_equal(full["hit"] == 1, np.isfinite(best))
    assert np.array_equal(full["toi"][full["hit"] == 1], best[np.isfinite(best)].astype(np.float32))


def test_degenerate_and_ragged_static_input(sge, cpu):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1], [2, 0, 0], [1e-7, 0, 0]], np.float32)
    idx = np.array([0, 2, 1, 0, 1, 3, 0, 4, 0, 1, 2], np.uint32)  # collinear, repeated vertex, 2 trailing indices
    cpu.rebuild_static([{"positions": pos, "indices": idx}])
    v, t, n = cpu.collision_counts()
    assert (v, t, n) == (5, 1, 1)
    cpu.rebuild_static([])
    assert cpu.collision_counts() == (0, 0, 0)
    assert cpu.capsule_cast(sge.make_queries([[0, 5, 0]], [[0, -10, 0]]))["hit"][0] == 0


def test_config1_settles_on_ground_quad(sge, ybot, cpu):
    """config 1: spawn (0,7.5,0), g=(0,-98,0), dt=1/60: bottom ends groundSnapSkin above y=-3."""
    build_scene(sge, cpu, 1, terrain_cells=None)
    cpu.resize(1)
    cpu.upload(bodies=sge.assets.default_bodies(1, np.array([[0, 7.5, 0]])), params=sge.assets.default_controller_params(1),
               controllers=sge.assets.default_controller_state(1), intents=sge.assets.default_intents(1),
               locomotion=sge.assets.default_locomotion(1, ybot), actions=sge.assets.default_actions(1, ybot, present=True))
    ys, states = [], []
    for s in range(600):
        cpu.tick()
        d = cpu.download(what=("bodies", "controllers", "locomotion"))
        ys.append(d["bodies"]["position"][0, 1])
        states.append(int(d["locomotion"]["state"][0]))
    assert abs(ys[-1] - (-3 + 2.5 + 0.05)) < 2e-3
    assert max(np.abs(np.diff(ys[-100:]))) < 1e-6  # at rest
    assert d["controllers"]["flags"][0] & 3 == 3 and states[-1] == sge.abi.LOCO_IDLE
    assert d["bodies"]["linearVelocity"][0, 1] == 0.0
    pal, _, _ = cpu.palettes()
    assert np.isfinite(pal).all()


def test_locomotion_state_machine_hysteresis(sge, ybot, cpu):
    """Systems.swift:297-324 with CharacterFactory's thresholds (0.15 / 0.3 / 6 / 5)."""
    abi = sge.abi
    build_scene(sge, cpu, 1, terrain_cells=None)
    cpu.resize(1)
    ctrl = sge.assets.default_controller_state(1)
    ctrl["flags"] = abi.CTRL_GROUNDED | abi.CTRL_GROUNDED_NEAR
    bodies = sge.assets.default_bodies(1, np.zeros((1, 3)))
    L = sge.assets.default_locomotion(1, ybot)

    def run(speed, state):
        bodies["linearVelocity"][0] = (speed, 0, 0)
        L2 = L.copy()
        L2["state"] = state
        cpu.upload(bodies=bodies, params=sge.assets.default_controller_params(1), controllers=ctrl,
                   intents=sge.assets.default_intents(1), locomotion=L2, actions=sge.assets.default_actions(1))
        cpu.tick(stages=abi.STAGE_LOCOMOTION)
        return int(cpu.download(what=("locomotion",))["locomotion"]["state"][0])

    assert run(0.2, abi.LOCO_IDLE) == abi.LOCO_IDLE and run(0.3, abi.LOCO_IDLE) == abi.LOCO_WALK
    assert run(6.0, abi.LOCO_IDLE) == abi.LOCO_RUN and run(5.5, abi.LOCO_WALK) == abi.LOCO_WALK
    assert run(5.5, abi.LOCO_RUN) == abi.LOCO_RUN and run(4.9, abi.LOCO_RUN) == abi.LOCO_WALK
    assert run(0.1, abi.LOCO_RUN) == abi.LOCO_IDLE and run(0.14, abi.LOCO_WALK) == abi.LOCO_IDLE
    assert run(9.0, abi.LOCO_FALLING) == abi.LOCO_RUN
    ctrl["flags"] = 0
    ctrl["groundDistance"] = 60.0
    assert run(0.0, abi.LOCO_IDLE) == abi.LOCO_FALLING
    ctrl["groundDistance"] = 10.0
    assert run(0.0, abi.LOCO_IDLE) == abi.LOCO_IDLE and run(0.0, abi.LOCO_FALLING) == abi.LOCO_FALLING


def test_tangents_helper(sge, cpu):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1], [5, 5, 5]], np.float32)
    nrm = np.tile([[0, 2, 0]], (4, 1)).astype(np.float32)
    uv = np.array([[0, 0], [1, 0], [0, 1], [0, 0]], np.float32)
    t = cpu.compute_tangents(pos, nrm, uv, np.array([0, 1, 2], np.uint32))
    assert np.allclose(t[:3], [[1, 0, 0, -1]] * 3) and np.array_equal(t[3], [1, 0, 0, 1])
    t16 = cpu.compute_tangents(pos, nrm, uv, np.array([0, 1, 2], np.uint16))
    assert np.array_equal(t, t16)


def _blend_expected(local_from, local_to, w, parent, lean_index=-1, run_lean_weight=0.0):
    """float64 restatement of ProceduralPoseSystem.swift:185-220 on reference-tool locals: lerp the translations, slerp the
    rotations (scipy, shortest arc), yaw-stable root (:206-215), run lean (:369-393); then the parent-before-child model
    product."""
    from scipy.spatial.transform import Rotation, Slerp

    B = local_from.shape[0]
    A = local_from.reshape(B, 4, 4).transpose(0, 2, 1)  # column-major 16 -> math matrices
    Bm = local_to.reshape(B, 4, 4).transpose(0, 2, 1)
    out = np.zeros((B, 4, 4))
    for i in range(B):
        ra, rb = A[i, :3, :3], Bm[i, :3, :3]
        t = A[i, :3, 3] + (Bm[i, :3, 3] - A[i, :3, 3]) * w
        if i == 0:
            yaw = np.arctan2(ra[0, 2], ra[2, 2])
            yq = Rotation.from_euler("y", yaw).as_matrix()
            pa, pb = yq.T @ ra, yq.T @ rb
            r = yq @ Slerp([0, 1], Rotation.from_matrix([pa, pb]))(w).as_matrix()
        else:
            r = Slerp([0, 1], Rotation.from_matrix([ra, rb]))(w).as_matrix()
        out[i] = np.eye(4)
        out[i, :3, :3], out[i, :3, 3] = r, t
    def chain(loc):
        model = np.zeros_like(loc)
        for i in range(B):
            model[i] = loc[i] if parent[i] < 0 else model[parent[i]] @ loc[i]
        return model

    model = chain(out)
    if run_lean_weight > 0.001 and lean_index >= 0:
        # the chest's right axis taken into its parent's space, a 10-degree * weight turn about it applied on the left
        right_world = model[lean_index][:3, 0] / np.linalg.norm(model[lean_index][:3, 0])
        p = parent[lean_index]
        right_local = Rotation.from_matrix(model[p][:3, :3]).inv().apply(right_world) if p >= 0 else right_world
        lean = np.eye(4)
        lean[:3, :3] = Rotation.from_rotvec(right_local * np.radians(10.0) * run_lean_weight).as_matrix()
        out[lean_index] = lean @ out[lean_index]
        model = chain(out)
    to16 = lambda m: m.transpose(0, 2, 1).reshape(B, 16)  # noqa: E731
    return to16(out), to16(model)


@pytest.mark.parametrize("case", [("Walking", 2, "Running", 4, 0.3), ("Running", 1, "Walking", 3, 0.65), ("Walking", 5, "Idle", 2, 0.5),
                                  ("Running", 4, "FallingIdle", 0, 0.85), ("Idle", 3, "Running", 1, 0.1)])
def test_locomotion_blend_matches_reference_tool_plus_slerp(sge, ybot, cpu, golden, case):
    """Pins the blending branch (P6): from/to locals come from the reference's own fit_motion.py (golden, float64); the blend
    between them is restated in float64 with scipy's slerp and compared with the oracle's float32 simd restatement."""
    abi = sge.abi
    from_name, ia, to_name, ib, blend_t = case
    state_of = {"Idle": abi.LOCO_IDLE, "Walking": abi.LOCO_WALK, "Running": abi.LOCO_RUN, "FallingIdle": abi.LOCO_FALLING}
    cpu.upload_skeleton(ybot)
    cpu.upload_profiles(ybot.profiles)
    cpu.resize(1)
    L = sge.assets.default_locomotion(1, ybot)
    fs, ts = state_of[from_name], state_of[to_name]
    L["fromState"], L["state"] = fs, ts
    L["flags"] |= abi.LOCO_IS_BLENDING
    phases = golden["phases"]
    for s, name, k in ((fs, from_name, ia), (ts, to_name, ib)):
        L["time"][0, s] = np.float32(phases[k]) * np.float32(ybot.profiles[ybot.profile_index(name)]["cycleDuration"])
    if ts == abi.LOCO_IDLE:           # weightTo = 1 - clamp(idleInertia) (:93-96)
        L["idleInertia"] = 1.0 - blend_t
        w = np.float64(np.float32(1.0) - np.float32(L["idleInertia"][0]))
    else:                             # smootherstep of blendT (:98-99)
        L["blendT"] = blend_t
        t = np.float64(np.float32(blend_t))
        w = t * t * t * (t * (t * 6 - 15) + 10)
    cpu.upload(bodies=sge.assets.default_bodies(1, np.zeros((1, 3))), params=sge.assets.default_controller_params(1),
               controllers=sge.assets.default_controller_state(1), intents=sge.assets.default_intents(1),
               locomotion=L, actions=sge.assets.default_actions(1))
    cpu.tick(dt=0.0, stages=abi.STAGE_POSE)
    _, mod, loc = cpu.palettes(model=True, local=True)
    run_w = w if ts == abi.LOCO_RUN else (1.0 - w if fs == abi.LOCO_RUN else 0.0)   # :103-113
    exp_local, exp_model = _blend_expected(golden[f"{from_name}.local"][ia], golden[f"{to_name}.local"][ib], w, ybot.parent,
                                           ybot.lean_index, run_w)
    scale = np.abs(exp_model).max()
    assert np.abs(loc[0] - exp_local).max() <= 2e-5 * scale, np.abs(loc[0] - exp_local).max()
    assert np.abs(mod[0] - exp_model).max() <= 2e-5 * scale, np.abs(mod[0] - exp_model).max()
    after = cpu.download(what=("locomotion",))["locomotion"]
    assert after["flags"][0] & abi.LOCO_IS_BLENDING                    # dt = 0: still mid-blend


@pytest.mark.parametrize("case", [("Walking", 1, 3, 0.4), ("Idle", 4, 2, 1.0), ("Running", 2, 5, 0.25)])
def test_action_layer_matches_reference_tool_plus_slerp(sge, ybot, cpu, golden, case):
    """Pins the action layer (P8, ProceduralPoseSystem.swift:286-338): base locomotion pose and the StandingDodgeBackward pose
    come from the reference's fit_motion.py (golden); their per-bone lerp / slerp by the action weight is restated in float64."""
    abi = sge.abi
    base_name, ia, ib, weight = case
    state_of = {"Idle": abi.LOCO_IDLE, "Walking": abi.LOCO_WALK, "Running": abi.LOCO_RUN}
    cpu.upload_skeleton(ybot)
    cpu.upload_profiles(ybot.profiles)
    cpu.resize(1)
    phases = golden["phases"]
    L = sge.assets.default_locomotion(1, ybot)
    st = state_of[base_name]
    L["state"], L["fromState"] = st, st
    L["time"][0, st] = np.float32(phases[ia]) * np.float32(ybot.profiles[ybot.profile_index(base_name)]["cycleDuration"])
    A = sge.assets.default_actions(1, ybot, present=True)
    k = ybot.profile_index("StandingDodgeBackward")
    A["profile"] = k
    A["flags"] = abi.ACTION_PRESENT | abi.ACTION_ACTIVE | abi.ACTION_IN_PLACE
    A["weight"] = weight
    A["time"] = np.float32(phases[ib]) * np.float32(ybot.profiles[k]["cycleDuration"])
    cpu.upload(bodies=sge.assets.default_bodies(1, np.zeros((1, 3))), params=sge.assets.default_controller_params(1),
               controllers=sge.assets.default_controller_state(1), intents=sge.assets.default_intents(1), locomotion=L, actions=A)
    cpu.tick(dt=0.0, stages=abi.STAGE_POSE)
    _, mod, loc = cpu.palettes(model=True, local=True)
    w = float(np.float32(weight))
    run_w = (1.0 if st == abi.LOCO_RUN else 0.0) * (1.0 - w)                      # runLeanWeight *= (1 - w), :297
    base, act = golden[f"{base_name}.local"][ia].copy(), golden["StandingDodgeBackward.local"][ib]
    B = base.shape[0]
    from scipy.spatial.transform import Rotation, Slerp
    exp = np.zeros((B, 4, 4))
    for i in range(B):
        a, b = base[i].reshape(4, 4).T, act[i].reshape(4, 4).T
        exp[i] = np.eye(4)
        exp[i, :3, 3] = a[:3, 3] + (b[:3, 3] - a[:3, 3]) * w
        exp[i, :3, :3] = Slerp([0, 1], Rotation.from_matrix([a[:3, :3], b[:3, :3]]))(w).as_matrix()
    exp16 = exp.transpose(0, 2, 1).reshape(B, 16)
    el, em = _blend_expected(exp16, exp16, 0.0, ybot.parent, ybot.lean_index, run_w)   # model chain (+ the scaled run lean)
    scale = np.abs(em).max()
    assert np.abs(loc[0] - el).max() <= 2e-5 * scale, np.abs(loc[0] - el).max()
    assert np.abs(mod[0] - em).max() <= 2e-5 * scale, np.abs(mod[0] - em).max()


def test_side_contact_only_cache_policy_known_answer(sge):
    """SideContactOnlyCachePolicy (Systems.swift:1136-1157) against DefaultContactCachePolicy (:1102-1134) on the oracle: a capsule
    resting slightly inside the ground quad is depenetrated by a GROUND contact (normal.y = 1 >= minGroundDot). The default policy
    records it in the contact manifold (ContactManifoldCache.update: one entry per overlapped triangle of the quad, 8 frames); the side-only policy returns at its
    `guard isSideContact` and leaves the cache empty. Position and velocity are the same either way."""
    import oracle_binding as ob
    A = sge.abi
    outs = []
    for flag in (0, A.STAGE_SIDE_CONTACT_CACHE):
        cpu = ob.oracle_engine()
        quad = np.array([[-40, 0, -40], [40, 0, -40], [40, 0, 40], [-40, 0, 40]], np.float32)
        cpu.rebuild_static([dict(positions=quad, indices=np.array([0, 1, 2, 0, 2, 3], np.uint32), modelMatrix=np.eye(4, dtype=np.float32).reshape(-1))])
        cpu.resize(1)
        params = sge.assets.default_controller_params(1)
        bodies = sge.assets.default_bodies(1, np.array([[0.0, 2.5 - 0.2, 0.0]]))  # bottom of the capsule 0.2 under the quad
        cpu.upload(bodies=bodies, params=params, controllers=sge.assets.default_controller_state(1),
                   intents=sge.assets.default_intents(1), locomotion=None, actions=None)
        cpu.tick(stages=A.STAGE_MOVE | flag)
        d = cpu.download(what=("bodies", "controllers"))
        outs.append((d["bodies"]["position"][0].copy(), int(d["controllers"]["manifoldCount"][0]), int(d["controllers"]["manifoldFrames"][0])))
        cpu.close()
    (p0, count0, frames0), (p1, count1, frames1) = outs
    assert np.array_equal(p0, p1)
    assert count0 == 2 and frames0 == 8  # both triangles of the quad overlap: a ground contact uses up to two hits (:764)
    assert count1 == 0 and frames1 == 0
