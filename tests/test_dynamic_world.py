"""CPU tests (oracle side) of the rest of the CollisionQuery surface: the dynamic triangle set, updateTransforms + BVH
refit, raycast, and PlatformCarry — SURVEY.md §8(a) rows C1 (updateTransforms), C3 (refit), C4/C5/C9 (combined
static+dynamic queries), C10 (raycast), C21 (PlatformCarry), C23 (refresh order)."""
import numpy as np
import pytest

import oracle_binding as ob
from scenes import PlatformScene, assert_struct_equal, box_mesh, spawn_on_platforms, translation_matrix


@pytest.fixture()
def cpu():
    e = ob.oracle_engine()
    yield e
    e.close()


def rot_y(deg):
    a = np.radians(deg)
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)  # [col][row]
    return m


def test_update_transforms_refits_like_the_reference(sge, cpu):
    pos, idx = sge.assets.make_synthetic_static_mesh(24, 16, 1.0)
    box = box_mesh(2, 1, 3)
    ents = [{"positions": pos, "indices": idx}, {"positions": box[0], "indices": box[1], "modelMatrix": translation_matrix((0, 6, 0)), "layer": 2}]
    cpu.rebuild_static(ents)
    before = cpu.collision_copy()
    m1 = rot_y(30.0)
    m1[3, :3] = (3.0, 4.0, -2.0)
    cpu.update_transforms(sge.abi.SET_STATIC, [1], m1.reshape(1, 16))
    after = cpu.collision_copy()
    # topology, order and the other entity are untouched
    for k in ("indices", "triOrder", "triLeaf"):
        assert np.array_equal(before[k], after[k])
    for f in ("left", "right", "start", "count", "parent"):
        assert np.array_equal(before["nodes"][f], after["nodes"][f])
    V0 = pos.shape[0]
    assert np.array_equal(before["positions"][:V0], after["positions"][:V0])
    expect = (np.c_[box[0], np.ones(8, np.float32)] @ m1)[:, :3]  # [col][row] storage: row-vector form
    assert np.abs(after["positions"][V0:] - expect).max() < 1e-5
    # every node bounds exactly the union of what is below it, triangle AABBs are those of the moved vertices
    tri = after["positions"][after["indices"].reshape(-1, 3)]
    assert np.array_equal(after["aabbs"][:, 0], tri.min(1)) and np.array_equal(after["aabbs"][:, 1], tri.max(1))
    nodes = after["nodes"]
    for i in range(len(nodes) - 1, -1, -1):
        nd = nodes[i]
        if nd["left"] < 0:
            t = after["triOrder"][nd["start"]:nd["start"] + nd["count"]]
            lo, hi = after["aabbs"][t, 0].min(0), after["aabbs"][t, 1].max(0)
        else:
            lo = np.minimum(nodes[nd["left"]]["boundsMin"], nodes[nd["right"]]["boundsMin"])
            hi = np.maximum(nodes[nd["left"]]["boundsMax"], nodes[nd["right"]]["boundsMax"])
        assert np.array_equal(nd["boundsMin"], lo) and np.array_equal(nd["boundsMax"], hi), i
    # queries against the refitted world == queries against a world built from scratch around the moved entity
    fresh = ob.oracle_engine()
    fresh.rebuild_static([ents[0], dict(ents[1], modelMatrix=m1.reshape(16))])
    assert fresh.collision_counts() == cpu.collision_counts()
    rng = np.random.default_rng(3)
    n = 600
    origin = np.c_[rng.uniform(-6, 10, n), rng.uniform(2, 12, n), rng.uniform(-8, 6, n)].astype(np.float32)
    delta = rng.normal(0, 4, (n, 3)).astype(np.float32)
    q = sge.make_queries(origin, delta, radius=0.6, half_height=0.5)
    a, b = cpu.capsule_cast(q), fresh.capsule_cast(q)
    assert np.array_equal(a["hit"], b["hit"]) and a["hit"].sum() > 50
    # (which of two triangles sharing the touched edge is reported depends on the visit order, which a rebuild changes)
    same = a["triangleIndex"] == b["triangleIndex"]
    assert np.array_equal(a["toi"], b["toi"]) and same.mean() > 0.8
    assert np.array_equal(a["position"][same], b["position"][same]) and np.array_equal(a["normal"][same], b["normal"][same])
    # skipped cases: unknown entity, entity without surviving triangles, empty list
    cpu.update_transforms(sge.abi.SET_STATIC, [7], np.eye(4, dtype=np.float32).reshape(1, 16))
    cpu.update_transforms(sge.abi.SET_STATIC, [], np.zeros((0, 16), np.float32))
    assert_struct_equal(cpu.collision_copy()["nodes"], after["nodes"], "nodes")
    fresh.close()


def test_dynamic_set_is_consulted_after_the_static_one(sge, cpu):
    gp, gi, gm = sge.assets.ground_plane()
    box = box_mesh(2, 0.5, 2)
    cpu.rebuild_static([{"positions": gp, "indices": gi, "modelMatrix": gm}])
    cpu.rebuild_dynamic([{"positions": box[0], "indices": box[1], "modelMatrix": translation_matrix((0, 1.0, 0)), "layer": 2}])
    assert cpu.collision_counts(sge.abi.SET_DYNAMIC)[1] == 12 and cpu.collision_counts()[1] == 2
    down = sge.make_queries(np.array([[0, 6, 0], [10, 6, 0]], np.float32), np.array([[0, -20, 0]] * 2, np.float32))
    h = cpu.capsule_cast(down)
    assert h["hit"].tolist() == [1, 1]
    assert h["triangleIndex"][0] >= 2 and h["triangleIndex"][1] < 2          # dynamic indices are offset by the static count
    assert h["toi"][0] == pytest.approx(6 - 1.5 - 2.5, abs=2e-3) and h["toi"][1] == pytest.approx(6 + 3 - 2.5, abs=2e-3)
    assert tuple(h["material"][0])[:2] == (np.float32(0.8), np.float32(0.6))
    # layer mask hides the platform
    down["mask"] = 1
    assert (cpu.capsule_cast(down)["triangleIndex"] < 2).all()
    # identical geometry in both sets: the static hit wins the tie (chooseNearest: a.toi <= b.toi ? a : b)
    cpu.rebuild_dynamic([{"positions": gp, "indices": gi, "modelMatrix": gm}])
    down["mask"] = 0xFFFFFFFF
    both = cpu.capsule_cast(down)
    assert (both["triangleIndex"] < 2).all()
    ov, cnt = cpu.capsule_overlap_all(sge.make_queries(np.array([[0, -3 + 2.2, 0]], np.float32)), 8)
    assert cnt[0] == 4 and sorted(ov[0]["triangleIndex"][:4].tolist()) == [0, 1, 2, 3]   # static pair first, then the dynamic pair
    assert ov[0]["triangleIndex"][:2].max() < 2 <= ov[0]["triangleIndex"][2:4].min()
    one, found = cpu.capsule_overlap(sge.make_queries(np.array([[0, -3 + 2.2, 0]], np.float32)))
    assert found[0] == 1 and one[0]["triangleIndex"] < 2                                  # a.depth >= b.depth ? a : b
    ov3, cnt3 = cpu.capsule_overlap_all(sge.make_queries(np.array([[0, -3 + 2.2, 0]], np.float32)), 3)
    assert cnt3[0] == 3 and (ov3[0]["triangleIndex"][:3] == ov[0]["triangleIndex"][:3]).all()
    # emptying the dynamic set
    cpu.rebuild_dynamic([])
    assert cpu.collision_counts(sge.abi.SET_DYNAMIC) == (0, 0, 0)
    assert cpu.capsule_overlap_all(sge.make_queries(np.array([[0, -3 + 2.2, 0]], np.float32)), 8)[1][0] == 2


def test_raycast_known_answers(sge, cpu):
    gp, gi, gm = sge.assets.ground_plane()
    box = box_mesh(1, 1, 1)
    cpu.rebuild_static([{"positions": gp, "indices": gi, "modelMatrix": gm, "material": (0.9, 0.8, 0)}])
    cpu.rebuild_dynamic([{"positions": box[0], "indices": box[1], "modelMatrix": translation_matrix((5, 0, 0)), "layer": 2}])
    o = np.array([[0, 7, 0], [0, 7, 0], [0, 7, 0], [0, -10, 0], [5, 7, 0], [5, 7, 0], [0, 0.5, 0], [0, 7, 0], [5.2, 0.1, 0.3]], np.float32)
    d = np.array([[0, -1, 0], [0, -2, 0], [0, 1, 0], [0, 1, 0], [0, -1, 0], [0, -1, 0], [1, 0, 0], [1, 0, 0], [0, 0, 1]], np.float32)
    r = cpu.raycast(o, d, 100.0)
    assert r["hit"].tolist() == [1, 1, 0, 1, 1, 1, 1, 0, 1]
    assert r["distance"][0] == pytest.approx(10.0) and r["distance"][1] == pytest.approx(5.0)   # direction is not normalised
    assert np.allclose(r["position"][0], (0, -3, 0)) and np.allclose(r["normal"][0], (0, 1, 0))
    assert np.allclose(r["normal"][3], (0, -1, 0)) and r["distance"][3] == pytest.approx(7.0)   # from below: normal faces the ray
    assert r["distance"][4] == pytest.approx(6.0) and r["triangleIndex"][4] >= 2               # the box top, dynamic index
    assert r["distance"][6] == pytest.approx(4.0) and np.allclose(r["normal"][6], (-1, 0, 0))
    assert r["distance"][8] == pytest.approx(0.7) and np.allclose(r["normal"][8], (0, 0, -1))   # from inside the box
    assert tuple(r["material"][0])[:2] == (np.float32(0.9), np.float32(0.8))
    # maxDistance is exclusive-ish (`t < closestT` with closestT = maxDistance); the mask filters per triangle layer
    assert cpu.raycast(o[:1], d[:1], 10.0)["hit"][0] == 0 and cpu.raycast(o[:1], d[:1], 10.001)["hit"][0] == 1
    masked = cpu.raycast(o[4:5], d[4:5], 100.0, mask=1)
    assert masked["hit"][0] == 1 and masked["distance"][0] == pytest.approx(10.0) and masked["triangleIndex"][0] < 2
    # brute force over all triangles agrees on random rays over a terrain
    pos, idx = sge.assets.make_synthetic_static_mesh(20, 14, 1.0)
    cpu.rebuild_dynamic([])
    cpu.rebuild_static([{"positions": pos, "indices": idx}])
    rng = np.random.default_rng(2)
    n = 300
    o = np.c_[rng.uniform(-9, 9, n), rng.uniform(3, 9, n), rng.uniform(-6, 6, n)].astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d[:, 1] = -np.abs(d[:, 1]) - 0.2
    r = cpu.raycast(o, d, 50.0)
    tri = pos[idx.reshape(-1, 3)].astype(np.float64)
    for k in range(0, n, 7):
        e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
        pv = np.cross(d[k].astype(np.float64), e2)
        det = (e1 * pv).sum(1)
        ok = np.abs(det) > 1e-9
        inv = np.where(ok, 1.0 / np.where(ok, det, 1), 0)
        tv = o[k] - tri[:, 0]
        u = (tv * pv).sum(1) * inv
        qv = np.cross(tv, e1)
        v = (d[k] * qv).sum(1) * inv
        t = (e2 * qv).sum(1) * inv
        good = ok & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t >= 0) & (t < 50)
        if good.any():
            assert r["hit"][k] == 1 and r["distance"][k] == pytest.approx(t[good].min(), rel=1e-4, abs=1e-4)
        else:
            assert r["hit"][k] == 0
    assert r["hit"].sum() > n // 2


def test_platform_carry_cases(sge, ybot, cpu):
    """PlatformCarry.computeDelta through one MOVE tick with gravity off and no collision geometry in reach."""
    cpu.rebuild_static([])
    P = sge.assets.default_controller_params(1)
    r, hh = float(P["radius"][0]), float(P["halfHeight"][0])
    top = 10.0
    platform = np.zeros(1, sge.abi.platform_dtype)
    platform["aabbMin"], platform["aabbMax"] = (-5, top - 1, -5), (5, top, 5)
    platform["delta"], platform["kinematic"], platform["hasAABB"] = (0.25, 0.0, -0.125), 1, 1
    stand = top + r + hh + 0.05

    def run(position, pf):
        spawn_on_platforms(sge, cpu, ybot, [position])
        cpu.upload_platforms(pf)
        cpu.tick(stages=sge.abi.STAGE_MOVE, gravity=(0, 0, 0))
        return cpu.download(what=("bodies",))["bodies"]["position"][0]

    assert np.allclose(run((0, stand, 0), platform) - (0, stand, 0), (0.25, 0, -0.125), atol=1e-6)       # riding: carried
    assert np.allclose(run((0, stand + 3, 0), platform), (0, stand + 3, 0))                                # too high above the top
    assert np.allclose(run((30, stand, 0), platform), (30, stand, 0))                                      # elsewhere
    side = run((-5 - r - 0.1, top - 0.5, 0), platform)                                                     # beside it, platform moving away
    assert np.allclose(side, (-5 - r - 0.1, top - 0.5, 0))
    toward = platform.copy()
    toward["delta"] = (-0.25, 0.3, 0.0)
    pushed = run((-5 - r - 0.1, top - 0.5, 0), toward)                                                     # moving into the capsule: pushed in XZ only
    assert np.allclose(pushed - (-5 - r - 0.1, top - 0.5, 0), (-0.25, 0, 0), atol=1e-6)
    off = platform.copy()
    off["kinematic"] = 0
    assert np.allclose(run((0, stand, 0), off), (0, stand, 0))                                             # not a kinematic body
    still = platform.copy()
    still["delta"] = (1e-5, 0, 0)
    assert np.allclose(run((0, stand, 0), still), (0, stand, 0))                                           # |delta|^2 < 1e-8
    two = np.concatenate([platform, platform])
    two["delta"][1] = (0.5, 0, 0)
    assert np.allclose(run((0, stand, 0), two) - (0, stand, 0), (0.5, 0, 0), atol=1e-6)                    # the larger carry wins
    assert np.allclose(run((0, stand, 0), None), (0, stand, 0))


def test_character_rides_a_moving_platform(sge, ybot, cpu):
    scene = PlatformScene(sge, cpu, starts=[(0, 2, 0)], velocities=[(3.0, 0, 1.5)])
    spawn_on_platforms(sge, cpu, ybot, [(0.5, 2 + 0.5 + 2.5 + 1.0, 0.25), (40, 4, 40)])
    for _ in range(150):
        scene.step()
    d = cpu.download(what=("bodies", "controllers"))
    flags = d["controllers"]["flags"]
    assert (flags & sge.abi.CTRL_GROUNDED_NEAR).all()
    rider, bystander = d["bodies"]["position"]
    travelled = 150 * np.array([3.0, 0, 1.5]) / 60.0
    moved = rider[[0, 2]] - (0.5, 0.25)
    assert (moved > 0.9 * travelled[[0, 2]]).all() and (moved <= travelled[[0, 2]] + 1e-6).all()  # carried once it landed
    assert rider[1] == pytest.approx(2 + 0.5 + 2.5 + 0.05, abs=0.05)                    # standing on its top (groundSnapSkin above)
    assert d["controllers"]["groundTriangleIndex"][0] >= 2                              # a dynamic-set triangle
    assert np.allclose(bystander[[0, 2]], (40, 40), atol=1e-3) and bystander[1] == pytest.approx(-3 + 2.5 + 0.05, abs=0.05)


def test_collision_query_service_decisions(sge, cpu):
    """CollisionQueryService (SceneServices.swift:33-207): rebuild vs transform update, from the entity snapshot."""
    A = sge.abi
    gp, gi, _ = sge.assets.ground_plane()
    box = box_mesh(2, 0.5, 2)
    ident = (0, 0, 0, 1)
    world = [
        {"id": 1, "translation": (0, -3, 0), "rotation": ident, "scale": (1, 1, 1), "positions": gp, "indices": gi, "bodyType": A.BODY_STATIC},
        {"id": 2, "translation": (5, 1, 0), "rotation": ident, "scale": (1, 1, 1), "positions": box[0], "indices": box[1]},            # no body: static set
        {"id": 3, "translation": (-5, 2, 0), "rotation": ident, "scale": (1, 1, 1), "positions": box[0], "indices": box[1],
         "bodyType": A.BODY_KINEMATIC, "platform": True, "position": (-5, 2, 0), "prevPosition": (-5, 2, 0), "layer": 2},
        {"id": 4, "translation": (0, 9, 0), "rotation": ident, "scale": (1, 1, 1), "positions": box[0], "indices": box[1], "collides": False},
    ]
    svc = sge.services.CollisionQueryService(cpu)
    svc.update(world)                                   # first call: no query yet -> rebuild
    assert svc.log == ["rebuild"] and cpu.collision_counts()[1] == 2 + 12 and cpu.collision_counts(A.SET_DYNAMIC)[1] == 12
    assert svc.slot == {1: (A.SET_STATIC, 0), 2: (A.SET_STATIC, 1), 3: (A.SET_DYNAMIC, 0)}
    svc.update(world)
    assert svc.log == []                                # nothing moved
    world[2]["translation"] = (-4.5, 2, 0)              # the kinematic platform moved
    world[2]["prevPosition"], world[2]["position"] = (-5, 2, 0), (-4.5, 2, 0)
    svc.update(world)
    assert svc.log == [("dynamic", [3])]
    probe = sge.make_queries(np.array([[-4.5 + 1.9, 8, 0], [-5 - 1.9, 8, 0]], np.float32), np.array([[0, -5.4, 0]] * 2, np.float32), radius=0.25, half_height=0.25)
    assert cpu.capsule_cast(probe)["hit"].tolist() == [1, 0]     # the refitted set is where the platform now is
    pf = svc.upload_platforms(world)
    assert pf.shape[0] == 1 and pf[0]["kinematic"] == 1 and np.allclose(pf[0]["delta"], (0.5, 0, 0))
    assert np.allclose(pf[0]["aabbMin"], (-6.5, 1.5, -2)) and np.allclose(pf[0]["aabbMax"], (-2.5, 2.5, 2))
    world[1]["rotation"] = tuple(sge.formats.quat_angle_axis(0.3, (0, 1, 0)))   # a static prop turned: static-set update
    world[2]["scale"] = (1, 2, 1)
    svc.update(world)
    assert sorted(svc.log) == [("dynamic", [3]), ("static", [2])]
    tiny = dict(world[1], translation=(5 + 5e-4, 1, 0))                           # |delta|^2 <= 1e-6: not a change
    world[1] = tiny
    svc.update(world)
    assert svc.log == []
    # structural changes -> rebuild
    for change in ("collides", "body", "mesh", "dirty", "added", "active", "marked"):
        if change == "collides":
            world[3]["collides"] = True
        elif change == "body":
            world[1]["bodyType"] = A.BODY_DYNAMIC
        elif change == "mesh":
            world[1]["indices"] = box[1][:-3]
        elif change == "dirty":
            world[0]["dirty"] = True
        elif change == "added":
            world.append(dict(world[1], id=9))
        elif change == "marked":
            svc.mark_dirty()
        svc.update(world, active_ids={1, 2, 3} if change == "active" else None)
        assert svc.log == ["rebuild"], change
        svc.update(world, active_ids={1, 2, 3} if change == "active" else None)
        assert svc.log == [], change
    assert world[0]["dirty"] is False
    assert cpu.collision_counts(A.SET_DYNAMIC)[1] == 12 + 11 + 11      # platform, the re-typed prop (one triangle dropped), its copy
