"""GPU parity tests: the HIP product (through the C ABI) against the CPU oracle on identical inputs.

Bars (BASELINE.json north_star: 1e-5 relative):
  - BVH structure, query TOIs/normals/triangle ids, and the whole CCD state: BIT-EXACT
    (both sides run IEEE float32 in the same order with contraction off);
  - palettes and skinned vertices: |gpu - cpu| <= 1e-5 * max|cpu| (libm vs OCML trig, FMA in the LBS kernel).
"""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from scenes import PlatformScene, assert_close, assert_struct_equal, box_mesh, build_scene, compare_states, spawn_on_platforms, translation_matrix

pytestmark = pytest.mark.gpu
REL = 1e-5


@pytest.fixture(scope="module")
def engines(sge):
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    yield gpu, cpu
    gpu.close()
    cpu.close()


def test_library_refuses_without_fallback(sge):
    # the product path is the HIP library; the loader has no alternative
    assert sge.abi.load_library().sge_abi_version() == sge.abi.SGE_ABI_VERSION == 2


def test_bvh_build_bit_exact(sge, engines):
    gpu, cpu = engines
    for cells in ((7, 5), (40, 28)):
        for e in engines:
            sge.crowd.upload_terrain(e, cells=cells)
        g, c = gpu.collision_copy(), cpu.collision_copy()
        for k in ("positions", "indices", "aabbs", "triOrder", "triLeaf"):
            assert np.array_equal(g[k], c[k]), k
        assert_struct_equal(g["nodes"], c["nodes"], "nodes")
        # invariants: every triangle in exactly one leaf, leaves hold <= 4
        nodes = g["nodes"]
        leaves = nodes[nodes["left"] < 0]
        assert leaves["count"].max() <= 4 and leaves["count"].sum() == len(g["triOrder"])
        assert sorted(g["triOrder"].tolist()) == list(range(len(g["triOrder"])))


def test_bvh_ornate_mirror(sge, engines):
    gpu, cpu = engines
    z = np.load(sge.assets.GOLDEN_DIR + "/ornate_mirror_static.npz")
    # row-major JSON -> simd columns (StaticMeshLoader.swift:127-134), then the demo's x8 scale and offset
    # (DemoScene.swift:328-335; without the scale most triangles fall under the 1e-10 area filter)
    m = z["transformRowMajor"].reshape(4, 4).astype(np.float32)
    m = (np.diag([1, 1, 1, 1]).astype(np.float32) @ m @ np.diag([8, 8, 8, 1]).astype(np.float32))
    m[:3, 3] += (-10, 1, 4)
    ents = [{"positions": z["positions"], "indices": z["indices"], "modelMatrix": m.T.reshape(16)}]
    for e in engines:
        e.rebuild_static(ents)
    g, c = gpu.collision_copy(), cpu.collision_copy()
    assert 14000 < g["triOrder"].shape[0] <= 14246
    for k in ("positions", "indices", "aabbs", "triOrder", "triLeaf"):
        assert np.array_equal(g[k], c[k]), k
    assert_struct_equal(g["nodes"], c["nodes"], "nodes")
    # queries against a real asset
    rng = np.random.default_rng(5)
    lo, hi = g["positions"].min(0), g["positions"].max(0)
    n = 512
    origin = rng.uniform(lo - 0.5, hi + 0.5, (n, 3)).astype(np.float32)
    delta = rng.normal(0, 1.0, (n, 3)).astype(np.float32)
    q = sge.make_queries(origin, delta, radius=0.4, half_height=0.3, mode=sge.abi.CAST)
    q["mode"] = rng.integers(0, 3, n)
    assert_struct_equal(gpu.capsule_cast(q), cpu.capsule_cast(q), "cast")
    go, gc = gpu.capsule_overlap_all(q, 8)
    co, cc = cpu.capsule_overlap_all(q, 8)
    assert np.array_equal(gc, cc)
    assert_struct_equal(go, co, "overlap")
    assert gc.max() > 0 and gpu.capsule_cast(q)["hit"].sum() > 10


def test_queries_bit_exact_on_terrain(sge, engines):
    gpu, cpu = engines
    for e in engines:
        terrain = sge.crowd.upload_terrain(e, cells=(64, 48))
    rng = np.random.default_rng(11)
    n = 2000
    x = rng.uniform(-30, 30, n)
    z = rng.uniform(-22, 22, n)
    y = sge.assets.terrain_height(x, z, *terrain["half"]) + rng.uniform(1.0, 6.0, n)
    origin = np.stack([x, y, z], -1).astype(np.float32)
    delta = np.zeros((n, 3), np.float32)
    kind = rng.integers(0, 4, n)
    delta[kind == 0] = (0, -0.8, 0)            # snap probe
    delta[kind == 1] = (0, -200.0, 0)          # fall probe
    delta[kind >= 2] = rng.normal(0, 1.5, ((kind >= 2).sum(), 3))
    q = sge.make_queries(origin, delta)
    q["mode"] = np.where(kind <= 1, sge.abi.CAST_GROUND, np.where(kind == 2, sge.abi.CAST_BLOCKING, sge.abi.CAST))
    gh, ch = gpu.capsule_cast(q), cpu.capsule_cast(q)
    assert ch["hit"].sum() > n // 3
    assert_struct_equal(gh, ch, "cast")
    # overlaps: push the capsules into the ground
    q2 = sge.make_queries(origin - np.array([0, 2.2, 0], np.float32))
    go, gc = gpu.capsule_overlap_all(q2, 8)
    co, cc = cpu.capsule_overlap_all(q2, 8)
    assert cc.max() == 8 and np.array_equal(gc, cc)
    assert_struct_equal(go, co, "overlap")
    gd, gf = gpu.capsule_overlap(q2)
    cd, cf = cpu.capsule_overlap(q2)
    assert cf.sum() > n // 2 and np.array_equal(gf, cf)
    assert_struct_equal(gd, cd, "overlap-deepest")
    for mh in (1, 3):
        go, gc = gpu.capsule_overlap_all(q2[:200], mh)
        co, cc = cpu.capsule_overlap_all(q2[:200], mh)
        assert np.array_equal(gc, cc)
        assert_struct_equal(go, co, "overlap%d" % mh)
    assert gpu.move_stats().overflow == 0


def test_cast_edge_cases(sge, engines):
    gpu, cpu = engines
    for e in engines:
        sge.crowd.upload_ground_plane(e)
    origin = np.array([[0, 0.5, 0], [0, 0.5, 0], [100, 5, 0], [0, -0.6, 0], [39.9, 2, 39.9]], np.float32)
    delta = np.array([[0, 0, 0], [0, -1e-7, 0], [0, -10, 0], [0, -1, 0], [0, -20, 0]], np.float32)
    q = sge.make_queries(origin, delta, mode=sge.abi.CAST_GROUND)
    gh, ch = gpu.capsule_cast(q), cpu.capsule_cast(q)
    assert_struct_equal(gh, ch, "cast")
    assert gh["hit"].tolist() == ch["hit"].tolist() and gh["hit"][0] == 0 and gh["hit"][1] == 0 and gh["hit"][2] == 0
    # analytic: vertical capsule at height h above y=-3 plane: toi ~= h - r - hh
    q = sge.make_queries([[0, 7.5, 0]], [[0, -200, 0]], mode=sge.abi.CAST_GROUND)
    h = gpu.capsule_cast(q)[0]
    assert h["hit"] == 1 and abs(h["toi"] - (7.5 + 3 - 2.5)) < 2e-3 and h["normal"][1] > 0.999
    # empty world
    for e in engines:
        e.rebuild_static([])
    assert gpu.capsule_cast(q)["hit"][0] == 0 and cpu.capsule_cast(q)["hit"][0] == 0


def test_skinning_kernel_vs_oracle(sge, engines):
    gpu, cpu = engines
    n = 5
    for e in engines:
        build_scene(sge, e, n, terrain_cells=(16, 12), rings=9, segments=7)
    for e in engines:
        e.tick(stages=sge.abi.STAGE_ALL)
    gpu.synchronize()
    gp, gn, gt = gpu.skinned()
    cp, cn, ct = cpu.skinned()
    assert gp.shape[0] == n * gpu.vertex_count
    assert_close(gp, cp, "skinned positions vs oracle", group=3 * gpu.vertex_count)
    assert np.abs(gn - cn).max() <= 2e-6 and np.abs(gt - ct).max() <= 2e-6
    assert np.allclose(np.linalg.norm(gn, axis=1), 1, atol=1e-5)
    # padded (Metal float3 stride) layout gives the same numbers
    gpu.set_option(sge.abi.OPT_SKIN_LAYOUT, sge.abi.LAYOUT_PADDED16)
    gpu.tick(dt=0.0, stages=sge.abi.STAGE_SKIN)
    pp, pn, pt = gpu.skinned()
    gpu.set_option(sge.abi.OPT_SKIN_LAYOUT, sge.abi.LAYOUT_PACKED)
    gpu.tick(dt=0.0, stages=sge.abi.STAGE_SKIN)
    qp, qn, qt = gpu.skinned()
    # the two layouts are separate template instantiations (FMA contraction may differ by an ulp)
    assert np.abs(pp - qp).max() <= 1e-6 * np.abs(qp).max() and np.abs(pn - qn).max() <= 3e-7 and np.abs(pt - qt).max() <= 3e-7


def test_bind_pose_identity_palette(sge, engines):
    """Known answer: palette = model(bindLocal) * invBindModel = I, so skinned == source."""
    gpu, _ = engines
    ybot, _, _ = build_scene(sge, gpu, 2, terrain_cells=(16, 12), rings=5, segments=5)
    L = sge.assets.default_locomotion(2, ybot)
    L["flags"] = 0  # no locomotion, no motion profile -> bind pose branch
    a = sge.assets.default_actions(2)
    gpu.upload(locomotion=L, actions=a)
    gpu.tick(dt=0.0, stages=sge.abi.STAGE_POSE | sge.abi.STAGE_SKIN)
    pal, _, _ = gpu.palettes()
    assert np.abs(pal - np.eye(4, dtype=np.float32).reshape(16)).max() < 2e-6
    p, nrm, _ = gpu.skinned()
    V = gpu.vertex_count
    assert np.abs(p[:V] - gpu.mesh["positions"]).max() < 5e-6
    assert np.abs(p[V:] - gpu.mesh["positions"]).max() < 5e-6


@pytest.mark.parametrize("mixed", [False, True])
def test_full_tick_parity(sge, engines, mixed):
    gpu, cpu = engines
    n = 96
    for e in engines:
        build_scene(sge, e, n, terrain_cells=(56, 40), seed=21 + mixed, mixed=mixed)
    steps = 150
    for s in range(steps):
        for e in engines:
            e.tick()
        if s in (0, 1, 5, 20, 60, steps - 1):
            gpu.synchronize()
            compare_states(sge, gpu, cpu, n)
    g = gpu.download()
    grounded = (g["controllers"]["flags"] & sge.abi.CTRL_GROUNDED_NEAR) != 0
    assert grounded.mean() > 0.8, "most characters should have landed"
    assert (g["locomotion"]["state"] != 0).any()
    gp, gn, gt = gpu.skinned()
    cp, cn, ct = cpu.skinned()
    assert_close(gp, cp, "skinned positions vs oracle", group=3 * gpu.vertex_count)
    assert gpu.move_stats().overflow == 0


def test_action_layer_and_mesh_rebind(sge, engines):
    gpu, cpu = engines
    n = 16
    for e in engines:
        ybot, _, st = build_scene(sge, e, n, terrain_cells=(32, 24), seed=4, mesh_inv_bind=True)
        a = st["actions"].copy()
        a["flags"] |= sge.abi.ACTION_ACTIVE
        a["weight"] = np.linspace(0.0, 1.0, n).astype(np.float32)
        a["time"] = np.linspace(0.0, 1.0, n).astype(np.float32)
        e.upload(actions=a)
    for s in range(40):
        for e in engines:
            e.tick()
    gpu.synchronize()
    compare_states(sge, gpu, cpu, n)


def test_config1_single_ybot_settles(sge, engines):
    """config 1: one Y-Bot dropped on the 80x80 quad at y=-3 settles with its bottom groundSnapSkin above it."""
    gpu, cpu = engines
    for e in engines:
        ybot, _, _ = build_scene(sge, e, 1, terrain_cells=None)
        e.resize(1)
        e.upload(bodies=sge.assets.default_bodies(1, np.array([[0, 7.5, 0]])),
                 params=sge.assets.default_controller_params(1), controllers=sge.assets.default_controller_state(1),
                 intents=sge.assets.default_intents(1), locomotion=sge.assets.default_locomotion(1, ybot),
                 actions=sge.assets.default_actions(1, ybot, present=True))
    ys = []
    for s in range(240):
        for e in engines:
            e.tick()
        if s % 30 == 29:
            gpu.synchronize()
            compare_states(sge, gpu, cpu, 1)
        ys.append(gpu.download(what=("bodies",))["bodies"]["position"][0, 1])
    assert abs(ys[-1] - (-3 + 2.5 + 0.05)) < 2e-3, ys[-1]
    assert abs(ys[-1] - ys[-20]) < 1e-6


def test_character_vs_character_sweeps(sge, engines):
    """config 5 on one GPU: the all-gathered capsule snapshot (world size 1 here) is binned into the XZ grid
    and swept; the oracle loops over all pairs like the reference (Systems.swift:1069)."""
    import torch

    gpu, cpu = engines
    n = 160

    class Cap:
        def resize(self, k):
            pass

        def upload(self, **kw):
            self.kw = kw

    for e in engines:
        ybot, terrain, _ = build_scene(sge, e, 1, terrain_cells=(40, 30), rings=3, segments=3)
        cap = Cap()
        sge.crowd.spawn_crowd(cap, ybot, n, terrain, seed=9, mode="ccd", agents=True)
        cap.kw["bodies"]["position"][:, 0] *= 0.35
        cap.kw["bodies"]["position"][:, 2] *= 0.35
        cap.kw["params"]["agentFlags"][::7] = sge.abi.AGENT_PRESENT  # some non-solid agents
        cap.kw["params"]["agentFlags"][3::11] |= sge.abi.AGENT_RADIUS_OVERRIDE
        cap.kw["params"]["agentRadiusOverride"][:] = 1.1
        e.resize(n)
        e.upload(**cap.kw)
    ex = sge.parallel.AgentExchange(gpu, n, 0, 1, torch.device("cuda", 0), None)
    stages = sge.abi.STAGE_ALL
    hits = 0
    for s in range(60):
        ex.step(stages=stages)
        cpu.tick(stages=stages | sge.abi.STAGE_AGENTS)
        if s % 10 == 9:
            gpu.synchronize()
            compare_states(sge, gpu, cpu, n)
    free = ob.oracle_engine()
    build_scene(sge, free, 1, terrain_cells=(40, 30), rings=3, segments=3)
    free.resize(n)
    free.upload(**cap.kw)
    for s in range(60):
        free.tick(stages=stages)
    a = free.download(what=("bodies",))["bodies"]["position"]
    b = cpu.download(what=("bodies",))["bodies"]["position"]
    assert np.abs(a - b).max() > 1e-3, "the scene must exercise capsule-capsule hits"
    free.close()


def test_agent_carried_beyond_the_snapshot_grid(sge, engines):
    """A solid agent that platform carry moves far outside the bounding box of the gathered snapshot (the XZ grid is built
    over the snapshot, the sweep starts from the carried position): it meets nobody, like in the oracle's all-pairs loop,
    and the grid lookup stays inside its cell table."""
    import torch

    gpu, cpu = engines
    n = 48
    ix, iz = np.meshgrid(np.arange(8), np.arange(6), indexing="ij")
    pos = np.stack([ix.reshape(-1) * 5.0 - 17.5, np.full(n, -0.45), iz.reshape(-1) * 5.0 - 12.5], -1)
    for e in engines:
        ybot, _, _ = build_scene(sge, e, 1, terrain_cells=None, rings=3, segments=3)
        e.resize(n)
        params = sge.assets.default_controller_params(n)
        params["agentFlags"] = sge.abi.AGENT_PRESENT | sge.abi.AGENT_SOLID
        ctrl = sge.assets.default_controller_state(n)
        ctrl["flags"] = sge.abi.CTRL_GROUNDED | sge.abi.CTRL_GROUNDED_NEAR
        vel = np.zeros((n, 3), np.float32)
        vel[:, 0] = 4.5
        e.upload(bodies=sge.assets.default_bodies(n, pos), params=params, controllers=ctrl, intents=sge.assets.default_intents(n, vel),
                 locomotion=sge.assets.default_locomotion(n, ybot), actions=sge.assets.default_actions(n, ybot, present=True))
    ex = sge.parallel.AgentExchange(gpu, n, 0, 1, torch.device("cuda", 0), None)
    # kinematic "platforms" whose top is flush with the ground quad under the outermost columns, each moving once, very far
    events = {3: ((15.0, -4.0, -40.0), (21.0, -3.0, 40.0), (400.0, 0.0, 250.0)),
              7: ((-21.0, -4.0, -40.0), (-15.0, -3.0, 40.0), (-400.0, 0.0, -250.0))}
    for s in range(12):
        pf = None
        if s in events:
            pf = np.zeros(1, sge.abi.platform_dtype)
            pf["aabbMin"], pf["aabbMax"], pf["delta"] = events[s]
            pf["kinematic"], pf["hasAABB"] = 1, 1
        for e in engines:
            e.upload_platforms(pf)
        ex.step()
        cpu.tick(stages=sge.abi.STAGE_ALL | sge.abi.STAGE_AGENTS)
        gpu.synchronize()
        compare_states(sge, gpu, cpu, n)
    x = gpu.download(what=("bodies",))["bodies"]["position"][:, 0]
    assert (x > 200).sum() >= 6 and (x < -200).sum() >= 6, "agents must have been carried away on both sides"
    for e in engines:
        e.upload_platforms(None)


def test_real_asset_scene_bvh_and_queries(sge, engines):
    """The engine's own static assets (17-Cheese + ornate mirror + Semla from the FBX sources, merged as in
    configs[3]: 135,928 triangles + the ground quad): BVH bit-exact, casts / overlaps bit-exact."""
    gpu, cpu = engines
    ents, bounds = sge.crowd.asset_scene_entities(("cheese", "mirror", "semla"))
    for e in engines:
        e.rebuild_static(ents)
    g, c = gpu.collision_copy(), cpu.collision_copy()
    assert g["triOrder"].shape[0] == 71680 + 14246 + 50002 + 2
    for k in ("positions", "indices", "aabbs", "triOrder", "triLeaf"):
        assert np.array_equal(g[k], c[k]), k
    assert_struct_equal(g["nodes"], c["nodes"], "nodes")
    rng = np.random.default_rng(11)
    n = 2048
    # probes start near the surfaces: a random mesh vertex plus a few units of noise
    origin = (g["positions"][rng.integers(0, len(g["positions"]), n)] + rng.normal(0, 2.5, (n, 3))).astype(np.float32)
    delta = (rng.normal(0, 1, (n, 3)) * rng.choice([0.3, 3.0, 30.0], (n, 1))).astype(np.float32)
    delta[: n // 4, 0] = 0
    delta[: n // 4, 2] = 0  # vertical probes, as the ground probe issues them
    q = sge.make_queries(origin, delta)
    q["mode"] = rng.integers(0, 3, n)
    q["mask"] = rng.choice([0xFFFFFFFF, 1, 2, 4, 1 << 30, 5], n)  # per-asset collision layers
    gh, ch = gpu.capsule_cast(q), cpu.capsule_cast(q)
    assert_struct_equal(gh, ch, "cast")
    assert gh["hit"].sum() > n // 16
    go, gc = gpu.capsule_overlap_all(q, 8)
    co, cc = cpu.capsule_overlap_all(q, 8)
    assert np.array_equal(gc, cc) and gc.max() == 8 and (gc > 0).sum() > n // 8
    assert_struct_equal(go, co, "overlap")


def test_real_assets_full_tick_parity(sge):
    """Real Y-Bot mesh (35,440 vertices, mesh inverse-bind re-bind) + merged real static scene, mixed motion states:
    the CCD state stays bit-exact with the oracle, palettes and every skinned vertex within 1e-5 relative."""
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    n = 128
    for e in (gpu, cpu):
        build_scene(sge, e, n, seed=5, mixed=True, real_mesh=True, asset_scene=("cheese", "mirror", "semla"))
    assert gpu.vertex_count == 35440
    no_skin = sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN
    steps = 200
    for s in range(steps):
        gpu.tick(stages=no_skin)
        ob.tick_mt(cpu, 8, stages=no_skin)
        if s in (0, 3, 30, 90, steps - 1):
            gpu.synchronize()
            compare_states(sge, gpu, cpu, n)
    d = gpu.download()
    assert ((d["controllers"]["flags"] & sge.abi.CTRL_GROUNDED_NEAR) != 0).mean() > 0.6
    assert gpu.move_stats().overflow == 0
    gpu.tick(dt=0.0, stages=sge.abi.STAGE_SKIN)
    ob.tick_mt(cpu, 8, dt=0.0, stages=sge.abi.STAGE_SKIN)
    gp, gn, gt = gpu.skinned()
    cp, cn, ct = cpu.skinned()
    assert_close(gp, cp, "skinned positions vs oracle", group=3 * gpu.vertex_count)
    assert np.abs(gn - cn).max() <= 2e-5 and np.abs(gt - ct).max() <= 2e-5
    gpu.close()
    cpu.close()


def test_dynamic_set_refit_and_raycast_bit_exact(sge, engines):
    """Rows C1/C3/C4/C10: static terrain + a dynamic set of boxes; updateTransforms on both sets (refit), then the
    BVH copies, capsule casts / overlaps and raycasts must match the oracle bit for bit."""
    gpu, cpu = engines
    pos, idx = sge.assets.make_synthetic_static_mesh(40, 28, 1.0)
    box = box_mesh(2.0, 0.6, 3.0)
    rng = np.random.default_rng(17)
    centres = np.c_[rng.uniform(-15, 15, 6), rng.uniform(3, 9, 6), rng.uniform(-10, 10, 6)].astype(np.float32)
    statics = [{"positions": pos, "indices": idx}, {"positions": box[0], "indices": box[1], "modelMatrix": translation_matrix((4, 8, 2)), "layer": 4}]
    dynamics = [{"positions": box[0], "indices": box[1], "modelMatrix": translation_matrix(c), "layer": 2, "material": (0.5, 0.4, 1)} for c in centres]

    def rot(deg, t):
        a = np.radians(deg)
        m = np.eye(4, dtype=np.float32)
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
        m[3, :3] = t
        return m.reshape(16)

    for e in engines:
        e.rebuild_static(statics)
        e.rebuild_dynamic(dynamics)
        e.update_transforms(sge.abi.SET_DYNAMIC, [0, 3, 5], np.stack([rot(20, centres[0] + 1), rot(-50, centres[3] - 2), rot(90, centres[5])]))
        e.update_transforms(sge.abi.SET_STATIC, [1], rot(33, (-6, 7, 1)).reshape(1, 16))
        e.update_transforms(sge.abi.SET_DYNAMIC, [9, 1], np.stack([rot(0, (0, 0, 0)), rot(10, centres[1] + (0, 1, 0))]))  # 9: unknown entity
    for which in (sge.abi.SET_STATIC, sge.abi.SET_DYNAMIC):
        g, c = gpu.collision_copy(which), cpu.collision_copy(which)
        for k in ("positions", "indices", "aabbs", "triOrder", "triLeaf"):
            assert np.array_equal(g[k], c[k]), (which, k)
        assert_struct_equal(g["nodes"], c["nodes"], "nodes")
    T_static = gpu.collision_counts()[1]
    n = 3000
    anchor = np.concatenate([centres, [[-6, 7, 1]], pos[rng.integers(0, len(pos), 8)]])
    origin = (anchor[rng.integers(0, len(anchor), n)] + rng.normal(0, 2.5, (n, 3))).astype(np.float32)
    delta = (rng.normal(0, 1, (n, 3)) * rng.choice([0.5, 4.0, 25.0], (n, 1))).astype(np.float32)
    q = sge.make_queries(origin, delta, radius=0.8, half_height=0.6)
    q["mode"] = rng.integers(0, 3, n)
    q["mask"] = rng.choice([0xFFFFFFFF, 1, 2, 4, 6], n)
    gh, ch = gpu.capsule_cast(q), cpu.capsule_cast(q)
    assert_struct_equal(gh, ch, "cast")
    assert (gh["triangleIndex"] >= T_static).sum() > 100 and ((gh["hit"] != 0) & (gh["triangleIndex"] < T_static)).sum() > 100
    go, gc = gpu.capsule_overlap_all(q, 8)
    co, cc = cpu.capsule_overlap_all(q, 8)
    assert np.array_equal(gc, cc) and (gc > 0).sum() > 200
    assert_struct_equal(go, co, "overlapAll")
    g1, gf = gpu.capsule_overlap(q)
    c1, cf = cpu.capsule_overlap(q)
    assert np.array_equal(gf, cf)
    assert_struct_equal(g1[gf != 0], c1[cf != 0], "overlap")
    direction = rng.normal(0, 1, (n, 3)).astype(np.float32)
    for mask in (0xFFFFFFFF, 2, 5):
        gr, cr = gpu.raycast(origin, direction, 40.0, mask=mask), cpu.raycast(origin, direction, 40.0, mask=mask)
        assert_struct_equal(gr, cr, "raycast")
    assert gr["hit"].sum() > 100
    # identical geometry in both sets: the static set wins every tie
    for e in engines:
        e.rebuild_static(statics)
        e.rebuild_dynamic(statics)
    assert_struct_equal(gpu.capsule_cast(q), cpu.capsule_cast(q), "cast(tie)")
    assert (gpu.capsule_cast(q)["triangleIndex"] < T_static).all()
    go, gc = gpu.capsule_overlap_all(q, 8)
    co, cc = cpu.capsule_overlap_all(q, 8)
    assert np.array_equal(gc, cc)
    assert_struct_equal(go, co, "overlapAll(tie)")
    assert_struct_equal(gpu.raycast(origin, direction, 40.0), cpu.raycast(origin, direction, 40.0), "raycast(tie)")
    for e in engines:
        e.rebuild_dynamic([])
        e.upload_platforms(None)


def test_kinematic_platforms_tick_parity(sge):
    """Rows C21/C23: characters riding, being pushed by and walking off moving platforms (dynamic set re-posed every step,
    PlatformCarry inputs uploaded every step) stay bit-exact with the oracle."""
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    ybot = sge.assets.YBotAssets()
    starts = [(0, 2, 0), (30, 1, 0), (-30, 3, 10), (0, 6, 40)]
    vels = [(3.0, 0, 1.5), (-4.0, 0, 0), (0, 1.2, 0), (0, -0.8, 2.0)]
    rng = np.random.default_rng(8)
    n = 96
    spots = np.array(starts)[rng.integers(0, 4, n)] + np.c_[rng.uniform(-9, 9, n), rng.uniform(4, 7, n), rng.uniform(-9, 9, n)]
    scenes = []
    for e in (gpu, cpu):
        sge.crowd.upload_character_assets(e, ybot, rings=3, segments=3)
        scenes.append(PlatformScene(sge, e, starts, vels))
        state = spawn_on_platforms(sge, e, ybot, spots)
        heading = rng.uniform(0, 2 * np.pi, n) if e is gpu else heading
        state["intents"] = sge.assets.default_intents(n, np.c_[np.cos(heading) * 3, np.zeros(n), np.sin(heading) * 3].astype(np.float32))
        e.upload(**state)
    rode = np.zeros(n, bool)
    for s in range(240):
        for sc in scenes:
            sc.step(stages=sge.abi.STAGE_ALL)
        if s % 20 == 0 or s == 239:
            gpu.synchronize()
            compare_states(sge, gpu, cpu, n)
            rode |= gpu.download(what=("controllers",))["controllers"]["groundTriangleIndex"] >= 2
    assert rode.sum() > n // 4, "a good part of the crowd must have stood on a platform"
    assert gpu.move_stats().overflow == 0
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("real", [False, True])
def test_heavy_four_wave_kernel_parity(sge, real):
    """SGE_OPT_HEAVY_THRESHOLD = 0 sends every character through the multi-wave kernel from its second step on (the
    default sends only the expensive ones): scheduling must not change a single bit of the result."""
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    gpu.set_option(sge.abi.OPT_HEAVY_THRESHOLD, 0)
    n = 96
    for e in (gpu, cpu):
        if real:
            build_scene(sge, e, n, seed=9, mixed=True, rings=3, segments=3, agents=True, asset_scene=("cheese", "mirror"))
        else:
            build_scene(sge, e, n, terrain_cells=(56, 40), seed=31, mixed=True, rings=3, segments=3, agents=True)
    import torch
    ex = sge.parallel.AgentExchange(gpu, n, 0, 1, torch.device("cuda", 0), None)   # character-vs-character sweeps on (world size 1)
    st = sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN
    for s in range(140):
        ex.step(stages=st)
        ob.tick_mt(cpu, 8, stages=st | sge.abi.STAGE_AGENTS)
        if s in (0, 1, 2, 10, 60, 139):
            gpu.synchronize()
            compare_states(sge, gpu, cpu, n)
    stats = gpu.move_stats()
    assert stats.overflow == 0 and stats.sweepTrips > 0
    # and back to the one-wave kernel mid-run
    gpu.set_option(sge.abi.OPT_HEAVY_THRESHOLD, -1)
    for s in range(20):
        ex.step(stages=st)
        ob.tick_mt(cpu, 8, stages=st | sge.abi.STAGE_AGENTS)
    gpu.synchronize()
    compare_states(sge, gpu, cpu, n)
    gpu.close()
    cpu.close()


def test_long_soak_parity(sge):
    """900 fixed steps (15 simulated seconds) of a mixed crowd with character-vs-character sweeps on the merged real scene,
    default scheduling (heavy characters in the multi-wave kernel, speculative ground samples): the CCD state must still
    equal the oracle's bit for bit at the end, with intents redirected on the way to keep everybody moving."""
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    n = 160
    states = []
    for e in (gpu, cpu):
        _, _, st0 = build_scene(sge, e, n, seed=77, mixed=True, agents=True, rings=3, segments=3, asset_scene=("cheese", "semla"), footprint=120.0)
        states.append(st0)
    import torch
    ex = sge.parallel.AgentExchange(gpu, n, 0, 1, torch.device("cuda", 0), None)
    st = sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN
    rng = np.random.default_rng(5)
    for s in range(900):
        if s % 150 == 149:  # turn everybody around: new headings, some stop
            heading = rng.uniform(0, 2 * np.pi, n)
            speed = rng.choice([0.0, 4.5, 12.5], n)
            intents = sge.assets.default_intents(n, np.c_[np.cos(heading) * speed, np.zeros(n), np.sin(heading) * speed].astype(np.float32))
            for e in (gpu, cpu):
                e.upload(intents=intents)
        ex.step(stages=st)  # snapshot exchange (world size 1) + tick with SGE_STAGE_AGENTS
        ob.tick_mt(cpu, 8, stages=st | sge.abi.STAGE_AGENTS)
        if s % 100 == 99:
            gpu.synchronize()
            compare_states(sge, gpu, cpu, n)
    d = gpu.download(what=("bodies", "controllers", "locomotion"))
    assert np.isfinite(d["bodies"]["position"]).all()
    assert len(np.unique(d["locomotion"]["state"])) >= 3
    assert gpu.move_stats().overflow == 0
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("out_layout", ["packed", "padded16"])
def test_skinning_encode_job_list(sge, out_layout):
    """RTSkinningEncoder.encode over a heterogeneous job list (B3): different meshes, bone counts, palettes, source layouts
    and destination offsets in one call — one launch on the GPU — against the oracle's per-job loop; then the one-job form."""
    import torch

    A = sge.abi
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(12)
    counts = [1, 255, 256, 257, 5000, 33, 12345]
    bones = [3, 65, 17, 256, 65, 1, 40]
    padded_src = [False, True, False, False, True, False, True]
    base, total = [], 0
    for v in counts:
        total += int(rng.integers(0, 5))            # gaps between the jobs' destination ranges
        base.append(total)
        total += v
    host_jobs, dev_jobs, keep = [], [], []
    for v, b, pad in zip(counts, bones, padded_src):
        pos = rng.normal(0, 1, (v, 3)).astype(np.float32)
        nrm = rng.normal(0, 1, (v, 3)).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        tan = np.c_[rng.normal(0, 1, (v, 3)), rng.choice([-1.0, 1.0], v)].astype(np.float32)
        idx = rng.integers(0, b, (v, 4)).astype(np.uint16)
        wgt = rng.uniform(0, 1, (v, 4)).astype(np.float32)
        wgt[rng.uniform(size=(v, 4)) < 0.4] = 0
        wgt[:, 0] = np.maximum(wgt[:, 0], 0.05)
        wgt[::17, 3] = -0.25                          # a negative weight is skipped like a zero one (`w > 0`)
        wgt /= np.maximum(wgt, 0).sum(1, keepdims=True)
        pal = np.tile(np.eye(4, dtype=np.float32).reshape(16), (b, 1))
        pal[:, :12] += rng.normal(0, 0.4, (b, 12)).astype(np.float32)
        pal[:, [3, 7, 11]] = 0
        host = {"sourcePositions": pos, "sourceNormals": nrm, "sourceTangents": tan, "sourceBoneIndices": idx, "sourceBoneWeights": wgt, "palette": pal}
        keep.append(host)
        host_jobs.append(dict({k: a.ctypes.data for k, a in host.items()}, paletteCount=b, vertexCount=v, dstBaseVertex=base[len(host_jobs)]))
        on_dev = {}
        for k, a in host.items():
            src = a
            if pad and k in ("sourcePositions", "sourceNormals"):
                src = np.zeros((v, 4), np.float32)
                src[:, :3] = a
            on_dev[k] = torch.from_numpy(np.ascontiguousarray(src)).to(dev)
        keep.append(on_dev)
        dev_jobs.append(dict({k: t.data_ptr() for k, t in on_dev.items()}, paletteCount=b, vertexCount=v, dstBaseVertex=base[len(dev_jobs)],
                             sourceLayout=A.LAYOUT_PADDED16 if pad else A.LAYOUT_PACKED))
    stride = 4 if out_layout == "padded16" else 3
    layout = A.LAYOUT_PADDED16 if out_layout == "padded16" else A.LAYOUT_PACKED
    ref = [np.full((total, 3), 7.0, np.float32), np.full((total, 3), 7.0, np.float32), np.full((total, 4), 7.0, np.float32)]
    cpu.skinning_encode(ref[0].ctypes.data, ref[1].ctypes.data, ref[2].ctypes.data, A.LAYOUT_PACKED, host_jobs)

    def run(jobs):
        out = [torch.full((total, stride), 7.0, dtype=torch.float32, device=dev), torch.full((total, stride), 7.0, dtype=torch.float32, device=dev),
               torch.full((total, 4), 7.0, dtype=torch.float32, device=dev)]
        gpu.skinning_encode(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), layout, jobs)
        gpu.synchronize()
        return [o.cpu().numpy() for o in out]

    got = run(dev_jobs)
    scale = np.abs(ref[0]).max()
    assert np.abs(got[0][:, :3] - ref[0]).max() <= REL * scale
    assert np.abs(got[1][:, :3] - ref[1]).max() <= 2e-5 and np.abs(got[2] - ref[2]).max() <= 2e-5
    gaps = np.ones(total, bool)
    for v, b0 in zip(counts, base):
        gaps[b0:b0 + v] = False
    assert (got[0][gaps] == 7.0).all() and (got[2][gaps] == 7.0).all()     # nothing written between the jobs
    # one job at a time (the single-launch-per-job path) gives the same numbers
    one = run(dev_jobs[4:5])
    sl = slice(base[4], base[4] + counts[4])
    assert np.abs(one[0][sl, :3] - ref[0][sl]).max() <= REL * scale and (one[0][: base[4]] == 7.0).all()
    # empty list and zero-vertex jobs are no-ops (RTSkinningEncoder.swift:32-35)
    untouched = run([])
    assert (untouched[0] == 7.0).all()
    untouched = run([dict(dev_jobs[1], vertexCount=0), dict(dev_jobs[2], vertexCount=0)])
    assert (untouched[0] == 7.0).all()
    gpu.close()
    cpu.close()


def test_context_buffers_and_caller_stream(sge):
    """sge_crowd_buffers / sge_skinned_mesh_buffers / sge_context_set_stream: a caller builds RTSkinningJobs over the
    context's own device buffers (the way RTGeometryCache.makeSkinningJob would) and runs everything on its own stream."""
    import torch

    A = sge.abi
    gpu = sge.CharacterEngine(0)
    lib, h = gpu.t.lib, gpu.h
    n = 6
    ybot, _, _ = build_scene(sge, gpu, n, terrain_cells=(16, 12), rings=5, segments=6)
    V, B = gpu.vertex_count, gpu.bone_count
    stream = torch.cuda.Stream(device=0)
    assert lib.sge_context_set_stream(h, C.c_void_p(stream.cuda_stream)) == 0
    for _ in range(5):
        gpu.tick()
    stream.synchronize()
    expect = [a.copy() for a in gpu.skinned()]
    pal, op, on, ot = (C.c_void_p() for _ in range(4))
    assert lib.sge_crowd_buffers(h, C.byref(pal), C.byref(op), C.byref(on), C.byref(ot)) == 0
    src = [C.c_void_p() for _ in range(5)]
    assert lib.sge_skinned_mesh_buffers(h, *[C.byref(p) for p in src]) == 0
    assert all(p.value for p in (pal, op, on, ot, *src))
    # wipe the outputs, then re-skin every character through the job-list entry point, in reverse order
    dev = torch.device("cuda", 0)
    with torch.cuda.stream(stream):
        scratch = [torch.zeros((n * V, 3), dtype=torch.float32, device=dev), torch.zeros((n * V, 3), dtype=torch.float32, device=dev),
                   torch.zeros((n * V, 4), dtype=torch.float32, device=dev)]
    jobs = [dict(sourcePositions=src[0].value, sourceNormals=src[1].value, sourceTangents=src[2].value, sourceBoneIndices=src[3].value,
                 sourceBoneWeights=src[4].value, palette=pal.value + c * B * 64, paletteCount=B, vertexCount=V, dstBaseVertex=c * V)
            for c in reversed(range(n))]
    gpu.skinning_encode(scratch[0].data_ptr(), scratch[1].data_ptr(), scratch[2].data_ptr(), A.LAYOUT_PACKED, jobs)
    stream.synchronize()
    got = [t.cpu().numpy() for t in scratch]
    for g, e in zip(got, expect):
        assert np.abs(g - e).max() <= 1e-6 * max(np.abs(e).max(), 1.0)
    # back to the context's own stream
    assert lib.sge_context_set_stream(h, None) == 0
    gpu.tick(dt=0.0, stages=A.STAGE_SKIN)
    gpu.synchronize()
    assert np.abs(gpu.skinned()[0] - expect[0]).max() <= 1e-6 * np.abs(expect[0]).max()
    gpu.close()


def test_demo_scene_example_parity(sge):
    """examples/demo_scene.py — the reference DemoScene's physical content (ground, mirror hulls, two kinematic platforms, one
    steered Y-Bot with the real skinned mesh) driven through CollisionQueryService every step — gives the same trace on the GPU
    and on the oracle: positions bit for bit, skinned vertices to 1e-5."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("demo_scene", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "demo_scene.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    tg, sg = demo.run(gpu, steps=480)
    tc, sc = demo.run(cpu, steps=480)
    assert len(tg) == len(tc) == 16
    for (s1, p1, f1, g1, l1), (s2, p2, f2, g2, l2) in zip(tg, tc):
        assert s1 == s2 and np.array_equal(p1, p2) and (f1, g1, l1) == (f2, g2, l2), (s1, p1, p2)
    assert_close(sg, sc, "sg vs sc")
    pushed = [p[0] for _, p, _, _, _ in tg]
    assert min(pushed) < -12.0            # it reached the mover and was pushed along with it
    # the renderer's next step: primary rays at the player's refitted acceleration structure (GPU: wide-BVH traversal over the
    # boxes of the last step's refit; oracle: scan over every triangle). The two sides skin independently (1e-5), so silhouette
    # pixels may differ; the image must agree almost everywhere and the depths where both hit.
    mg, dg, hg = demo.render_probe(gpu)
    mc, dc, hc = demo.render_probe(cpu)
    assert mg.sum() > 100 and (mg != mc).sum() <= 0.02 * mg.sum()
    both = mg & mc
    assert np.abs(dg[both] - dc[both]).max() < 2e-3 and np.median(np.abs(dg[both] - dc[both])) < 1e-5
    same = both.reshape(-1) & (hg["primitive"] == hc["primitive"])
    assert same.sum() > 0.9 * both.sum()
    assert np.abs(hg["normal"][same] - hc["normal"][same]).max() < 1e-3
    gpu.close()
    cpu.close()


def test_full_size_properties(sge):
    """BASELINE.json configs[2] at full size (10k clones x 14,080 vertices vs 71,680 triangles), checked through
    size-independent properties: characters are independent, so an oracle run over a RANDOM SUBSET of the crowd
    must reproduce the GPU's result for those indices bit for bit; clones that start identical stay identical;
    nobody tunnels through the terrain; two runs give identical checksums."""
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    ybot = sge.assets.YBotAssets()
    n = 10000
    steps = 40
    sge.crowd.upload_character_assets(gpu, ybot)
    terrain = sge.crowd.upload_terrain(gpu)
    assert gpu.vertex_count == 14080 and gpu.collision_counts()[1] == 71680
    state0 = sge.crowd.spawn_crowd(gpu, ybot, n, terrain)
    # make characters 1 and 2 exact clones of character 0
    for k in state0:
        state0[k][1] = state0[k][0]
        state0[k][2] = state0[k][0]
    gpu.upload(**state0)

    def run():
        gpu.upload(**state0)
        for _ in range(steps):
            gpu.tick()
        gpu.synchronize()
        d = gpu.download(what=("bodies", "controllers", "locomotion"))
        p, nrm, tan = gpu.skinned(0, 3 * gpu.vertex_count)
        # checksum of checksums over the whole skinned output, computed from strided samples to bound PCIe time
        samples = [gpu.skinned(i * gpu.vertex_count, 256)[0].view(np.uint32).sum(dtype=np.uint64) for i in range(0, n, 97)]
        return d, (p, nrm, tan), np.asarray(samples)

    d1, (p, nrm, tan), chk1 = run()
    d2, _, chk2 = run()
    assert np.array_equal(chk1, chk2)                                   # deterministic
    assert_struct_equal(d1["bodies"], d2["bodies"], "bodies(run1 vs run2)")
    V = gpu.vertex_count
    assert np.array_equal(p[:V], p[V:2 * V]) and np.array_equal(p[:V], p[2 * V:3 * V])   # clones stay identical
    assert np.array_equal(tan[:V], tan[V:2 * V])
    assert np.isfinite(p).all() and np.allclose(np.linalg.norm(nrm, axis=1), 1, atol=1e-5)
    # nobody fell through: capsule bottom above the terrain surface (minus contact skin)
    pos = d1["bodies"]["position"]
    ground = sge.assets.terrain_height(pos[:, 0], pos[:, 2], *terrain["half"])
    assert ((pos[:, 1] - 2.5) - ground > -0.35).all()
    assert gpu.move_stats().overflow == 0
    # subset parity against the oracle
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(n, 96, replace=False))
    sge.crowd.upload_character_assets(cpu, ybot)
    cpu.rebuild_static([{"positions": terrain["positions"], "indices": terrain["indices"]}])
    cpu.resize(len(pick))
    cpu.upload(**{k: v[pick] for k, v in state0.items()})
    for _ in range(steps):
        cpu.tick()
    c = cpu.download(what=("bodies", "controllers", "locomotion"))
    assert_struct_equal(d1["bodies"][pick], c["bodies"], "bodies(subset)")
    assert_struct_equal(d1["controllers"][pick], c["controllers"], "controllers(subset)")
    assert np.array_equal(d1["locomotion"]["state"][pick], c["locomotion"]["state"])
    gp = gpu.skinned(int(pick[5]) * V, V)[0]
    cp = cpu.skinned(5 * V, V)[0]
    assert_close(gp, cp, "skinned positions vs oracle", group=3 * gpu.vertex_count)
    gpu.close()
    cpu.close()


def _lbs_stages(sge):
    A = sge.abi
    return A.STAGE_LOCOMOTION | A.STAGE_ACTION | A.STAGE_POSE | A.STAGE_WRITEBACK | A.STAGE_SKIN


def test_config_lbs_only_parity(sge, engines):
    """BASELINE.json configs[1] at a size the oracle finishes: clones in steady Running state with
    per-clone phase offsets, pose + palette + LBS, no MOVE stage — the stage set `bench.py --workload lbs` times."""
    gpu, cpu = engines
    n = 64
    for e in engines:
        ybot, _, st = build_scene(sge, e, n, terrain_cells=None, mode="lbs", rings=9, segments=7)
    assert (st["locomotion"]["state"] == sge.abi.LOCO_RUN).all() and len(np.unique(st["locomotion"]["motionTime"])) == n
    stages = _lbs_stages(sge)
    before = gpu.download(what=("bodies",))["bodies"]
    for s in range(90):
        for e in engines:
            e.tick(stages=stages)
        if s in (0, 1, 30, 89):
            gpu.synchronize()
            compare_states(sge, gpu, cpu, n)
            gp, gn, gt = gpu.skinned()
            cp, cn, ct = cpu.skinned()
            assert_close(gp, cp, "skinned positions vs oracle", group=3 * gpu.vertex_count)
            assert np.abs(gn - cn).max() <= 2e-5 and np.abs(gt - ct).max() <= 2e-5
    # no collision stage ran: bodies are untouched, the pose moved
    assert_struct_equal(gpu.download(what=("bodies",))["bodies"], before, "bodies")
    pal, _, _ = gpu.palettes()
    assert np.abs(pal[0] - pal[1]).max() > 1e-3, "clones run at different phases"


def test_config_lbs_only_full_size(sge):
    """configs[1] at BASELINE's full size (10k clones x 14,080 vertices), through size-independent properties: a random
    subset re-run on the oracle reproduces the GPU's palettes / skinned vertices for those clones (1e-5), two runs give
    identical checksums, clones with equal clocks are identical, every normal is unit length."""
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    ybot = sge.assets.YBotAssets()
    n, steps = 10000, 25
    sge.crowd.upload_character_assets(gpu, ybot)
    assert gpu.vertex_count == 14080
    state0 = sge.crowd.spawn_crowd(gpu, ybot, n, None, mode="lbs")
    for k in state0:
        state0[k][1] = state0[k][0]
    stages = _lbs_stages(sge)
    V = gpu.vertex_count

    def run():
        gpu.upload(**state0)
        for _ in range(steps):
            gpu.tick(stages=stages)
        gpu.synchronize()
        chk = [gpu.skinned(i * V, 512)[0].view(np.uint32).sum(dtype=np.uint64) for i in range(0, n, 89)]
        return gpu.download(what=("locomotion",))["locomotion"], np.asarray(chk)

    l1, chk1 = run()
    l2, chk2 = run()
    assert np.array_equal(chk1, chk2)
    assert_struct_equal(l1, l2, "locomotion(run1 vs run2)")
    p, nrm, tan = gpu.skinned(0, 2 * V)
    assert np.array_equal(p[:V], p[V:]) and np.array_equal(tan[:V], tan[V:])
    assert np.isfinite(p).all() and np.allclose(np.linalg.norm(nrm, axis=1), 1, atol=1e-5)
    last = gpu.skinned((n - 1) * V, V)[0]
    assert np.isfinite(last).all() and np.abs(last - p[:V]).max() > 1e-3      # the last clone was written, at its own phase
    rng = np.random.default_rng(1)
    pick = np.sort(rng.choice(n, 48, replace=False))
    sge.crowd.upload_character_assets(cpu, ybot)
    cpu.resize(len(pick))
    cpu.upload(**{k: v[pick] for k, v in state0.items()})
    for _ in range(steps):
        cpu.tick(stages=stages)
    gpal = np.stack([gpu.palettes(int(i), 1)[0][0] for i in pick])
    cpal, _, _ = cpu.palettes()
    assert_close(gpal, cpal, "palettes vs oracle", group=gpal.shape[-2] * 16)
    for k in (0, 17, 47):
        gp = gpu.skinned(int(pick[k]) * V, V)[0]
        cp = cpu.skinned(k * V, V)[0]
        assert_close(gp, cp, "gp vs cp")
    gpu.close()
    cpu.close()


def test_overlap_mode_parity(sge):
    """SGE_OPT_OVERLAP_SKIN (the bench's default): skin(n) runs on a second stream beside move(n+1) + pose(n+1). Scheduling only:
    CCD state bit-exact with the oracle at every check, and the skinned vertices read after any step are that step's."""
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    gpu.set_option(sge.abi.OPT_OVERLAP_SKIN, 1)
    n = 160
    for e in (gpu, cpu):
        build_scene(sge, e, n, seed=13, mixed=True, rings=9, segments=7, asset_scene=("cheese",), footprint=120.0)
    for s in range(120):
        gpu.tick()
        ob.tick_mt(cpu, 8)
        if s in (0, 1, 2, 17, 60, 119):
            compare_states(sge, gpu, cpu, n)      # downloads join both streams
            gp, gn, gt = gpu.skinned()
            cp, cn, ct = cpu.skinned()
            assert_close(gp, cp, "skinned positions vs oracle", group=3 * gpu.vertex_count)
            assert np.abs(gn - cn).max() <= 2e-5 and np.abs(gt - ct).max() <= 2e-5
    # stage subsets and the non-overlapped path mid-run
    gpu.tick(stages=sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN)
    ob.tick_mt(cpu, 8, stages=sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN)
    gpu.set_option(sge.abi.OPT_OVERLAP_SKIN, 0)
    for s in range(5):
        gpu.tick()
        ob.tick_mt(cpu, 8)
    compare_states(sge, gpu, cpu, n)
    assert_close(gpu.skinned()[0], cpu.skinned()[0], "skinned positions", group=3 * gpu.vertex_count)
    assert gpu.move_stats().overflow == 0
    gpu.close()
    cpu.close()


def test_resident_lbs_form_writes_the_same_streams(sge, monkeypatch):
    """Overlap mode skins large crowds (from 6,000 characters on) with resident workgroups that draw their work units
    from a ticket counter; SGE_SKIN_PERSISTENT (read when a context is created) forces the form on (q/4 workgroups per CU) or off (0).
    Same arithmetic either way, so the three output streams must be bit-identical: a small crowd whose characters are split over
    several work units and a larger one with the form forced, then the automatic choice at 16,500 characters against the form
    switched off. All have more units than resident workgroups, so most units are drawn from the counter."""
    ybot = sge.assets.YBotAssets()
    for n, settings in ((37, ("0", "8")), (1500, ("0", "7")), (16500, ("0", None))):
        outs = []
        for setting in settings:
            if setting is None:
                monkeypatch.delenv("SGE_SKIN_PERSISTENT", raising=False)
            else:
                monkeypatch.setenv("SGE_SKIN_PERSISTENT", setting)
            gpu = sge.CharacterEngine(0)
            try:
                gpu.set_option(sge.abi.OPT_OVERLAP_SKIN, 1)
                sge.crowd.upload_character_assets(gpu, ybot)
                terrain = sge.crowd.upload_terrain(gpu)
                sge.crowd.spawn_crowd(gpu, ybot, n, terrain, seed=21, mode="ccd", mixed=True)
                for _ in range(4):
                    gpu.tick()
                V = gpu.vertex_count
                first = max(0, n - 3) * V  # the last characters: certainly drawn from the counter
                outs.append([a.copy() for a in gpu.skinned(first_vertex=first, vertex_count=3 * V)] +
                            [a.copy() for a in gpu.skinned(first_vertex=(n // 2) * V, vertex_count=2 * V)])
            finally:
                gpu.close()
        for a, b in zip(*outs):
            assert np.array_equal(a, b), n
    monkeypatch.delenv("SGE_SKIN_PERSISTENT", raising=False)


def test_api_edge_cases(sge):
    """Empty crowd, empty world, state errors and argument checks of the C ABI (status codes, no crashes)."""
    gpu = sge.CharacterEngine(0)
    ybot = sge.assets.YBotAssets()
    lib, h = gpu.t.lib, gpu.h
    # nothing uploaded yet: an empty tick is fine, stages that need assets report a state error
    gpu.resize(0)
    gpu.tick()
    gpu.resize(4)
    with pytest.raises(sge.SgeError):
        gpu.tick(stages=sge.abi.STAGE_POSE)
    with pytest.raises(sge.SgeError):
        gpu.tick(stages=sge.abi.STAGE_SKIN)
    # an empty collision world: characters simply fall
    sge.crowd.upload_character_assets(gpu, ybot, rings=3, segments=3)
    gpu.rebuild_static([])
    assert gpu.collision_counts() == (0, 0, 0)
    q = sge.make_queries(np.zeros((3, 3), np.float32), np.array([[0, -5, 0]] * 3, np.float32))
    assert gpu.capsule_cast(q)["hit"].sum() == 0 and gpu.capsule_overlap_all(q, 8)[1].sum() == 0
    assert gpu.raycast(np.zeros((2, 3), np.float32), np.array([[0, -1, 0]] * 2, np.float32), 10.0)["hit"].sum() == 0
    state = {"bodies": sge.assets.default_bodies(4, np.zeros((4, 3))), "params": sge.assets.default_controller_params(4),
             "controllers": sge.assets.default_controller_state(4), "intents": sge.assets.default_intents(4),
             "locomotion": sge.assets.default_locomotion(4, ybot), "actions": sge.assets.default_actions(4, ybot, present=True)}
    gpu.upload(**state)
    for _ in range(30):
        gpu.tick()
    d = gpu.download(what=("bodies", "controllers", "locomotion"))
    assert (d["bodies"]["position"][:, 1] < -5).all() and (d["controllers"]["flags"] & sge.abi.CTRL_GROUNDED).sum() == 0
    assert (d["locomotion"]["state"] == sge.abi.LOCO_FALLING).all()
    # static bodies are skipped by the move stage (Systems.swift:1845)
    state["bodies"]["bodyType"][:2] = sge.abi.BODY_STATIC
    gpu.upload(**state)
    gpu.tick()
    y = gpu.download(what=("bodies",))["bodies"]["position"][:, 1]
    assert (y[:2] == 0).all() and (y[2:] < 0).all()
    # argument checks
    assert lib.sge_characters_resize(h, -1) != 0
    assert lib.sge_capsule_overlap_all_batch(h, None, 1, 8, None, None) != 0
    qq = np.ascontiguousarray(q)
    out = np.zeros((3, 9), sge.abi.overlap_hit_dtype)
    cnt = np.zeros(3, np.int32)
    assert lib.sge_capsule_overlap_all_batch(h, sge.abi.ptr(qq), 3, 9, sge.abi.ptr(out), sge.abi.ptr(cnt)) != 0   # max_hits > 8
    assert lib.sge_capsule_overlap_all_batch(h, sge.abi.ptr(qq), 3, 0, sge.abi.ptr(out), sge.abi.ptr(cnt)) != 0
    assert lib.sge_collision_update_transforms(h, 2, None, None, 0) != 0                                         # unknown set
    too_many = np.zeros(sge.abi.SGE_MAX_PLATFORMS + 1, sge.abi.platform_dtype)
    assert lib.sge_platforms_upload(h, sge.abi.ptr(too_many), too_many.shape[0]) != 0
    bad = {"positions": np.zeros((3, 3), np.float32), "indices": np.array([0, 1, 7], np.uint32)}
    with pytest.raises(sge.SgeError):
        gpu.rebuild_static([bad])
    assert b"index out of range" in lib.sge_last_error()
    d = sge.abi.TickDesc()
    d.dt, d.stages, d.first, d.count = 1 / 60, sge.abi.STAGE_ALL, 3, 5                                          # range past the crowd
    assert lib.sge_tick(h, C.byref(d)) != 0
    assert lib.sge_context_set_option(h, 99, 1) != 0
    gpu.close()


def test_full_size_kernel_paths_agree_on_the_real_scene(sge):
    """BASELINE.json configs[2] on the mesh it names (17-Cheese) at full size, where the oracle is too slow to follow: the three
    schedules of the move stage must give the same state bit for bit — the default (four characters per wavefront, the expensive
    few in the multi-wave launch), everybody in the multi-wave launch (its near / far passes and crawl speculation on 10,000
    characters), and nobody in it (the densely tessellated rim of the cheese, ~900 candidates per cast, inside the grouped kernel:
    queue overflow sweeps, far passes, speculation at full lane count) — and an oracle run over a random subset must reproduce
    the default run for those characters."""
    abi = sge.abi
    ybot = sge.assets.YBotAssets()
    n, steps = 10000, 120
    st = abi.STAGE_ALL & ~abi.STAGE_SKIN  # the skinned output is a function of the palettes, which are compared

    def run(threshold):
        eng = sge.CharacterEngine(0)
        sge.crowd.upload_character_assets(eng, ybot, rings=3, segments=3)
        scene = sge.crowd.upload_asset_scene(eng, ("cheese",))
        eng.set_option(abi.OPT_HEAVY_THRESHOLD, threshold)
        state0 = sge.crowd.spawn_crowd(eng, ybot, n, scene)
        for _ in range(steps):
            eng.tick(stages=st)
        eng.synchronize()
        out = eng.download()
        pal = eng.palettes(0, 64)[0]
        stats = eng.move_stats()
        eng.close()
        return state0, scene, out, pal, stats

    state0, scene, base, pal0, stats0 = run(4000)
    assert stats0.overflow == 0
    for threshold in (0, -1):
        _, _, other, pal, stats = run(threshold)
        assert stats.overflow == 0
        for k in ("bodies", "controllers", "locomotion", "actions"):
            assert_struct_equal(base[k], other[k], "%s (heavy threshold %d vs default)" % (k, threshold))
        assert np.array_equal(pal0, pal)
    # subset against the oracle
    cpu = ob.oracle_engine()
    sge.crowd.upload_character_assets(cpu, ybot, rings=3, segments=3)
    sge.crowd.upload_asset_scene(cpu, ("cheese",))
    rng = np.random.default_rng(3)
    cost_rank = np.argsort(-np.abs(base["bodies"]["position"][:, 0]))  # characters near the rim are the expensive ones
    pick = np.sort(np.unique(np.concatenate([rng.choice(n, 40, replace=False), cost_rank[:24]])))
    cpu.resize(len(pick))
    cpu.upload(**{k: v[pick] for k, v in state0.items()})
    for _ in range(steps):
        cpu.tick(stages=st)
    c = cpu.download()
    assert_struct_equal(base["bodies"][pick], c["bodies"], "bodies(subset vs oracle)")
    assert_struct_equal(base["controllers"][pick], c["controllers"], "controllers(subset vs oracle)")
    cpu.close()



def _strided_checksums(eng, n, V, stride=97):
    """Order-independent integer checksums over every `stride`-th character's three skinned streams (bit patterns summed)."""
    sums = np.zeros(3, np.uint64)
    for c in range(0, n, stride):
        for k, a in enumerate(eng.skinned(first_vertex=c * V, vertex_count=V)):
            sums[k] += np.ascontiguousarray(a).view(np.uint32).astype(np.uint64).sum()
    return sums


@pytest.mark.parametrize("mesh", ["synthetic", "ybot"])
def test_bench_default_full_size(sge, mesh):
    """The combination bench.py times, as one test: SGE_OPT_OVERLAP_SKIN on, every stage including skin, 10,000 characters on the
    17-Cheese mesh (BASELINE.json configs[2]) — the three-stream, two-palette-buffer schedule at full size — with the synthetic
    14,080-vertex mesh of the headline and with the FBX-derived Y-Bot (35,440 welded vertices, the real weights and bone indices:
    bench.py's `real_mesh` object; SURVEY 8d "value distributions"). Checked against (1) an oracle run over a subset (random
    characters + the expensive ones at the rim): bodies / controllers bit-exact, palettes and the skinned vertices of three subset
    characters <= 1e-5; (2) the same crowd stepped in serial order (overlap off): every state array and the strided checksums of
    the three output streams identical."""
    abi = sge.abi
    ybot = sge.assets.YBotAssets()
    n, settle, steps = 10000, 40, 60
    upload_mesh = sge.crowd.upload_ybot_mesh if mesh == "ybot" else sge.crowd.upload_character_assets

    def run(overlap):
        eng = sge.CharacterEngine(0)
        eng.set_option(abi.OPT_OVERLAP_SKIN, 1 if overlap else 0)
        upload_mesh(eng, ybot)
        scene = sge.crowd.upload_asset_scene(eng, ("cheese",))
        state0 = sge.crowd.spawn_crowd(eng, ybot, n, scene)
        for _ in range(settle + steps):
            eng.tick()
        eng.synchronize()
        return eng, state0

    gpu, state0 = run(True)
    V, B = gpu.vertex_count, gpu.bone_count
    assert V == (35440 if mesh == "ybot" else 14080)
    out = gpu.download()
    assert gpu.move_stats().overflow == 0
    sums = _strided_checksums(gpu, n, V)
    rng = np.random.default_rng(5)
    rim = np.argsort(-np.abs(out["bodies"]["position"][:, 0]))  # characters near the rim of the cheese are the expensive ones
    pick = np.sort(np.unique(np.concatenate([rng.choice(n, 48, replace=False), rim[:24]])))
    assert len(pick) >= 64
    pal = np.stack([gpu.palettes(int(c), 1)[0][0] for c in pick])
    skinned = {int(c): [a.copy() for a in gpu.skinned(first_vertex=int(c) * V, vertex_count=V)] for c in (pick[0], pick[len(pick) // 2], pick[-1])}
    gpu.close()

    # (2) serial order
    ser, _ = run(False)
    other = ser.download()
    for k in ("bodies", "controllers", "locomotion", "actions"):
        assert_struct_equal(out[k], other[k], "%s (overlap vs serial order)" % k)
    assert np.array_equal(sums, _strided_checksums(ser, n, V))
    ser.close()

    # (1) oracle over the subset
    cpu = ob.oracle_engine()
    upload_mesh(cpu, ybot)
    sge.crowd.upload_asset_scene(cpu, ("cheese",))
    cpu.resize(len(pick))
    cpu.upload(**{k: v[pick] for k, v in state0.items()})
    for _ in range(settle + steps):
        ob.tick_mt(cpu, 8)
    c = cpu.download()
    assert_struct_equal(out["bodies"][pick], c["bodies"], "bodies(subset vs oracle)")
    assert_struct_equal(out["controllers"][pick], c["controllers"], "controllers(subset vs oracle)")
    cpal = cpu.palettes(0, len(pick))[0]
    assert_close(pal, cpal, "palettes vs oracle", group=pal.shape[-2] * 16)
    for j, cidx in ((0, int(pick[0])), (len(pick) // 2, int(pick[len(pick) // 2])), (len(pick) - 1, int(pick[-1]))):
        cp, cn, ct = cpu.skinned(first_vertex=j * V, vertex_count=V)
        gp, gn, gt = skinned[cidx]
        assert_close(gp, cp, "gp vs cp")
        assert np.abs(gn - cn).max() <= 2e-5 and np.abs(gt - ct).max() <= 2e-5
    cpu.close()


def test_pose_beside_the_next_move_stage(sge, monkeypatch):
    """Overlap mode runs pose(n) on a stream of its own, beside move(n+1): the move kernels leave a copy of what the animation stages
    read of bodies / controllers (PoseInput, two buffers) and no longer store transformRotation, which is the pose stage's. The
    schedule must not change a bit: 3,000 characters stepped back to back (no host synchronisation between the ticks, so the
    launches really overlap), with everything that leaves the pipelined form mixed in — a partial-range tick, a tick without the skin
    stage, a tick with the separation stage, states re-uploaded mid-run, a pose-only tick — against the same sequence with the
    pose launch kept on the main stream (SGE_POSE_PIPELINE=0, read when a context is created)."""
    abi = sge.abi
    ybot = sge.assets.YBotAssets()
    n, V_check = 3000, (0, 1499, 2999)

    def run(piped):
        monkeypatch.setenv("SGE_POSE_PIPELINE", "1" if piped else "0")
        eng = sge.CharacterEngine(0)
        eng.set_option(abi.OPT_OVERLAP_SKIN, 1)
        sge.crowd.upload_character_assets(eng, ybot, rings=8, segments=8)
        scene = sge.crowd.upload_asset_scene(eng, ("cheese",), footprint=120.0)
        sge.crowd.spawn_crowd(eng, ybot, n, scene, seed=77, mixed=True, agents=True)
        snaps = []
        for s in range(64):
            if s == 9:
                eng.tick(first=100, count=900)                                  # partial range: pose on the main stream
            elif s == 17:
                eng.tick(stages=abi.STAGE_ALL & ~abi.STAGE_SKIN)                # no skin stage
            elif s == 23:
                eng.tick(stages=abi.STAGE_ALL | abi.STAGE_SEPARATION)           # bodies move again behind the move kernels
            elif s == 31:
                eng.tick(stages=abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_POSE | abi.STAGE_WRITEBACK | abi.STAGE_SKIN)
            elif s == 40:                                                       # teleport a block of characters, reset their clocks
                st = eng.download(first=500, count=64)
                st["bodies"]["position"][:, 1] += 3.0
                st["bodies"]["linearVelocity"][:] = 0
                st["locomotion"]["motionTime"][:] = 0
                eng.upload(first=500, bodies=st["bodies"], locomotion=st["locomotion"])
                eng.tick()
            else:
                eng.tick()
            if s in (8, 9, 10, 17, 18, 23, 24, 31, 32, 40, 41, 63):
                d = eng.download()
                V = eng.vertex_count
                snaps.append((s, d, eng.palettes()[0].copy(), [[a.copy() for a in eng.skinned(first_vertex=c * V, vertex_count=V)] for c in V_check]))
        assert eng.move_stats().overflow == 0
        eng.close()
        return snaps

    a, b = run(True), run(False)
    monkeypatch.delenv("SGE_POSE_PIPELINE", raising=False)
    for (s, da, pa, va), (_, db, pb, vb) in zip(a, b):
        for k in ("bodies", "controllers", "locomotion", "actions"):
            assert_struct_equal(da[k], db[k], "%s after tick %d (pose beside move vs pose in front of it)" % (k, s))
        assert np.array_equal(pa, pb), s
        for ca, cb in zip(va, vb):
            for x, y in zip(ca, cb):
                assert np.array_equal(x, y), s


def test_schedule_transitions_leave_no_launch_behind(sge):
    """The three-stream schedule across everything that changes it in mid-run, with no host synchronisation of the caller's own in
    between: the overlap option switched off and on, the context moved to a caller's stream and back, the crowd resized (smaller,
    then larger again) and re-uploaded while launches of the old crowd are in flight. A context in serial order fed the same calls
    must end with the same states and the same skinned streams, bit for bit."""
    import torch

    A = sge.abi
    ybot = sge.assets.YBotAssets()
    n = 2000

    def run(scheduled):
        eng = sge.CharacterEngine(0)
        lib, h = eng.t.lib, eng.h
        sge.crowd.upload_character_assets(eng, ybot, rings=8, segments=8)
        scene = sge.crowd.upload_asset_scene(eng, ("cheese",), footprint=100.0)
        sge.crowd.spawn_crowd(eng, ybot, n, scene, seed=5, mixed=True)
        state0 = eng.download()
        caller = torch.cuda.Stream(device=0)
        if scheduled:
            eng.set_option(A.OPT_OVERLAP_SKIN, 1)
        for _ in range(12):
            eng.tick()
        if scheduled:
            eng.set_option(A.OPT_OVERLAP_SKIN, 0)                    # pending pose / skin launches are joined inside
        for _ in range(3):
            eng.tick()
        if scheduled:
            eng.set_option(A.OPT_OVERLAP_SKIN, 2)                    # 2: the overlap schedule on a caller's stream as well (ABI version 2)
            assert lib.sge_context_set_stream(h, C.c_void_p(caller.cuda_stream)) == 0
        for _ in range(5):
            eng.tick()
        if scheduled:
            assert lib.sge_context_set_stream(h, None) == 0          # back to the context's own stream
        for _ in range(4):
            eng.tick()
        mid = eng.download()
        eng.resize(n // 4)                                           # launches of the 2,000-character crowd may still be in flight
        eng.upload(**{k: v[: n // 4] for k, v in state0.items()})
        for _ in range(6):
            eng.tick()
        small = eng.download()
        small_skin = [a.copy() for a in eng.skinned()]
        eng.resize(n)
        eng.upload(**state0)
        for _ in range(6):
            eng.tick()
        out = eng.download()
        V = eng.vertex_count
        skin = [a.copy() for a in eng.skinned(first_vertex=(n - 2) * V, vertex_count=2 * V)]
        pal = eng.palettes()[0].copy()
        assert eng.move_stats().overflow == 0
        eng.close()
        return mid, small, small_skin, out, skin, pal

    a, b = run(True), run(False)
    for k in ("bodies", "controllers", "locomotion", "actions"):
        assert_struct_equal(a[0][k], b[0][k], "%s before the resize" % k)
        assert_struct_equal(a[1][k], b[1][k], "%s of the smaller crowd" % k)
        assert_struct_equal(a[3][k], b[3][k], "%s at the end" % k)
    for x, y in zip(a[2] + a[4], b[2] + b[4]):
        assert np.array_equal(x, y)
    assert np.array_equal(a[5], b[5])


def test_overlap_on_a_caller_stream_with_a_consumer(sge):
    """SGE_OPT_OVERLAP_SKIN behind the reference's calling convention: RTSkinningEncoder.encode enqueues on the CALLER's command
    buffer and the consumer enqueued right behind it sees the skinned vertices (RTSkinningEncoder.swift:27-56,
    RayTracingScene.swift:35-43). Here the context runs on a caller-provided stream with the overlap option on; after every tick a
    consumer on a SECOND caller stream orders itself with sge_skin_wait, copies the position stream (device to device) and hands
    the streams back with sge_skin_consumed — no host synchronisation anywhere in the loop. Every copy must be exactly that step's positions as a serial-order context produces
    them. Also: the palette pointer alternates between the two buffers sge_crowd_palette_buffers reports, and jobs built from
    the freshly queried pointer after two more ticks re-skin the crowd to the same streams (advisor finding, round 2)."""
    import torch

    A = sge.abi
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    ybot = sge.assets.YBotAssets()
    n, steps = 1500, 6

    def make(overlap):
        eng = sge.CharacterEngine(0)
        eng.set_option(A.OPT_OVERLAP_SKIN, 2 if overlap else 0)   # 2 = the explicit opt-in for a caller-provided stream
        sge.crowd.upload_character_assets(eng, ybot)
        terrain = sge.crowd.upload_terrain(eng)
        sge.crowd.spawn_crowd(eng, ybot, n, terrain, seed=31, mode="ccd", mixed=True)
        return eng

    ser = make(False)
    V, B = ser.vertex_count, ser.bone_count
    expect = []
    for _ in range(steps):
        ser.tick()
        expect.append(ser.skinned()[0].copy())
    ser.close()

    gpu = make(True)
    lib, h = gpu.t.lib, gpu.h
    dev = torch.device("cuda", 0)
    main, consumer = torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)
    assert lib.sge_context_set_stream(h, C.c_void_p(main.cuda_stream)) == 0
    snaps = [torch.zeros((n * V, 3), dtype=torch.float32, device=dev) for _ in range(steps)]
    torch.cuda.synchronize()
    bufs = (C.c_void_p * 2)()
    latest = C.c_int32(-1)
    seen = []
    for k in range(steps):
        gpu.tick()
        pal, op = C.c_void_p(), C.c_void_p()
        assert lib.sge_crowd_buffers(h, C.byref(pal), C.byref(op), None, None) == 0
        assert lib.sge_crowd_palette_buffers(h, bufs, C.byref(latest)) == 0
        assert pal.value == bufs[latest.value]
        seen.append(latest.value)
        assert lib.sge_skin_wait(h, C.c_void_p(consumer.cuda_stream)) == 0
        assert hip.hipMemcpyAsync(C.c_void_p(snaps[k].data_ptr()), op, n * V * 12, 3, C.c_void_p(consumer.cuda_stream)) == 0
        assert lib.sge_skin_consumed(h, C.c_void_p(consumer.cuda_stream)) == 0  # ... and the next skin launch behind the copy
    assert seen == [(seen[0] + k) % 2 for k in range(steps)], seen  # a whole-crowd pose stage flips the buffer every tick
    consumer.synchronize()
    for k in range(steps):
        assert np.array_equal(snaps[k].cpu().numpy(), expect[k]), "consumer copy of step %d" % k
    # jobs from the freshly queried palette pointer: re-skin into scratch buffers on the caller's stream
    src = [C.c_void_p() for _ in range(5)]
    assert lib.sge_skinned_mesh_buffers(h, *[C.byref(p) for p in src]) == 0
    pal = C.c_void_p()
    assert lib.sge_crowd_buffers(h, C.byref(pal), None, None, None) == 0
    with torch.cuda.stream(main):
        scratch = [torch.zeros((n * V, 3), dtype=torch.float32, device=dev), torch.zeros((n * V, 3), dtype=torch.float32, device=dev),
                   torch.zeros((n * V, 4), dtype=torch.float32, device=dev)]
    jobs = [dict(sourcePositions=src[0].value, sourceNormals=src[1].value, sourceTangents=src[2].value, sourceBoneIndices=src[3].value,
                 sourceBoneWeights=src[4].value, palette=pal.value + c * B * 64, paletteCount=B, vertexCount=V, dstBaseVertex=c * V)
            for c in range(n)]
    gpu.skinning_encode(scratch[0].data_ptr(), scratch[1].data_ptr(), scratch[2].data_ptr(), A.LAYOUT_PACKED, jobs)
    main.synchronize()
    assert np.abs(scratch[0].cpu().numpy() - expect[-1]).max() <= 1e-6 * np.abs(expect[-1]).max()
    assert lib.sge_context_set_stream(h, None) == 0
    gpu.close()


def test_side_contact_cache_policy_parity(sge, engines):
    """KinematicMoveStopSystem.init(gravity:contactCachePolicy:) with the reference's second policy, SideContactOnlyCachePolicy
    (Systems.swift:1136-1157: the depenetration pass records only side contacts) = SGE_STAGE_SIDE_CONTACT_CACHE: bit-exact against
    the oracle over 150 steps on the hilly terrain, and not the same as the default policy's trace (the flag does something)."""
    gpu, cpu = engines
    A = sge.abi
    n = 96
    for e in engines:
        build_scene(sge, e, n, terrain_cells=(56, 40), seed=23, mixed=True)
    st = A.STAGE_ALL | A.STAGE_SIDE_CONTACT_CACHE
    for s in range(150):
        for e in engines:
            e.tick(stages=st)
        if s in (0, 5, 60, 149):
            gpu.synchronize()
            compare_states(sge, gpu, cpu, n)
    side_only = gpu.download()["controllers"].copy()
    assert gpu.move_stats().overflow == 0
    ref = sge.CharacterEngine(0)
    build_scene(sge, ref, n, terrain_cells=(56, 40), seed=23, mixed=True)
    for s in range(150):
        ref.tick()
    default = ref.download()["controllers"]
    ref.close()
    assert side_only.tobytes() != default.tobytes()


class _SyntheticRig:
    """A rig shaped like assets.YBotAssets (same attributes), with a random hierarchy and random Fourier profiles: what the pose
    kernel's generic paths need that the Y-Bot never shows — bone counts other than 65, orders above 4, translation channels on
    bones other than the root, a last pass whose bones ARE animated, bones whose parent sits in the same pass."""

    def __init__(self, bones, order, seed, animate_all, translate_some, chain=False):
        rng = np.random.default_rng(seed)
        self.names = ["mixamorig:Hips"] + ["bone%03d" % i for i in range(1, bones)]
        if bones > 12:
            self.names[5] = "mixamorig:Spine2"
        self.parent = np.full(bones, -1, np.int32)
        for i in range(1, bones):
            self.parent[i] = i - 1 if (chain and i % 3) else int(rng.integers(max(0, i - 9), i))
        self.translations = rng.uniform(-8, 8, (bones, 3)).astype(np.float32)
        self.pre_rotation_degrees = rng.uniform(-40, 40, (bones, 3)).astype(np.float32)
        self.unit_scale = 0.026
        self.root_fix_degrees = np.array([0, 180, 0], np.float32)
        self.zero_root = True
        self.pelvis_index = 0
        self.lean_index = 5 if bones > 12 else -1
        self.profile_names = ["Idle", "Walking", "Running", "FallingIdle", "StandingDodgeBackward"]
        self.profiles = []
        ncoef = 1 + 2 * order
        for k, name in enumerate(self.profile_names):
            present = np.ones(bones, np.uint8) if animate_all else (rng.uniform(size=bones) < 0.8).astype(np.uint8)
            present[0] = 1
            count = np.full((bones, 6), 255, np.uint8)
            coeffs = np.zeros((bones, 6, 17), np.float32)
            for b in range(bones):
                if not present[b]:
                    continue
                axes = [3, 4, 5] + ([0, 1, 2] if (b == 0 or (translate_some and b % 7 == 3)) else [])
                for a in axes:
                    if rng.uniform() < 0.1:
                        continue                                        # a nil axis
                    c = ncoef if rng.uniform() < 0.8 else int(rng.integers(0, ncoef + 1))   # ragged and empty rows too
                    count[b, a] = c
                    amp = 25.0 if a >= 3 else 3.0
                    coeffs[b, a, :c] = rng.normal(0, amp, c) / (1 + np.arange(c))
            self.profiles.append({"name": name, "order": order if k != 1 else max(1, order - 2), "cycleDuration": float(rng.uniform(0.6, 1.6)),
                                  "sampleFps": 30, "bonePresent": present, "coeffCount": count, "coeffs": coeffs})

    @property
    def bone_count(self):
        return len(self.names)

    def profile_index(self, name):
        return self.profile_names.index(name)


@pytest.mark.parametrize("bones,order,animate_all,translate_some,chain", [(64, 8, True, True, False), (130, 6, True, True, False),
                                                                        (100, 4, False, False, True), (33, 8, False, True, False)])
def test_pose_kernel_generic_rigs(sge, engines, bones, order, animate_all, translate_some, chain):
    """pose_kernel beyond the Y-Bot (round 3's kernel visits bones by slots, has a static-last-pass shortcut, a one-product model
    path for later passes and an instantiation for orders up to 8): random rigs of 33 / 64 / 100 / 130 bones (one, two, three
    passes), orders 4 / 6 / 8 with ragged and nil axes, translation on non-root bones, an animated or a partly static last pass,
    long parent chains — locomotion blends, the action layer, ground align and run lean on top. Palettes within 1e-5 of the
    oracle, CCD state bit-exact, over 40 steps of a mixed crowd."""
    gpu, cpu = engines
    rig = _SyntheticRig(bones, order, seed=100 + bones, animate_all=animate_all, translate_some=translate_some, chain=chain)
    n = 48
    for e in engines:
        e.set_option(sge.abi.OPT_STORE_POSE_DEBUG, 1)
        sge.crowd.upload_character_assets(e, rig, rings=2, segments=3)
        terrain = sge.crowd.upload_terrain(e, cells=(24, 16))
        st0 = sge.crowd.spawn_crowd(e, rig, n, terrain, seed=9, mode="ccd", mixed=True)
        a = st0["actions"].copy()
        a["flags"][::3] |= sge.abi.ACTION_ACTIVE                      # every third character starts its action layer
        a["weight"][::3] = 0.4
        e.upload(actions=a)
    for s in range(40):
        for e in engines:
            e.tick()
        if s in (0, 1, 7, 39):
            compare_states(sge, gpu, cpu, n)
    gp, gn, gt = gpu.skinned()
    cp, cn, ct = cpu.skinned()
    assert np.abs(gp - cp).max() <= REL * max(np.abs(cp).max(), 1.0)
