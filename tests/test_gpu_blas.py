"""GPU parity tests of the step after skinning: the per-frame refit of the skinned characters'
acceleration structures (RTAccelerationBuilder.swift:113-145) and what a ray reads at a hit (RayTracing.metalinc:242-296).

Bars: the refitted boxes are EXACT (min / max of float32 vertices) against a numpy scan of the index buffer; the closest
hit of a ray (hit / miss, primitive id) is EXACT against the oracle's scan over every triangle on the SAME skinned
vertices, distances / barycentrics / shading frame within 1e-6 (same IEEE arithmetic on both sides).
"""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from blas_ref import expected_bounds
from scenes import build_scene

pytestmark = pytest.mark.gpu


def _scene(sge, eng, n, real, **kw):
    if real:
        return build_scene(sge, eng, n, terrain_cells=(24, 16), real_mesh=True, **kw)
    return build_scene(sge, eng, n, terrain_cells=(24, 16), rings=9, segments=8, **kw)


def _topology(sge, eng):
    return sge.CharacterEngine.blas_topology(eng.mesh["positions"], eng.mesh["indices"])


@pytest.mark.parametrize("real", [False, True])
@pytest.mark.parametrize("layout", ["packed", "padded16"])
def test_refit_boxes_are_exact(sge, real, layout):
    gpu = sge.CharacterEngine(0)
    try:
        n = 5 if real else 9
        gpu.set_option(sge.abi.OPT_SKIN_LAYOUT, sge.abi.LAYOUT_PADDED16 if layout == "padded16" else sge.abi.LAYOUT_PACKED)
        _scene(sge, gpu, n, real, mixed=True)
        info = gpu.blas_build(gpu.mesh["indices"])
        topo = _topology(sge, gpu)
        assert (info.entryCount, info.wideCount, info.clusterCount) == (topo["info"].entryCount, topo["info"].wideCount, topo["info"].clusterCount)
        V = gpu.vertex_count
        gpu.set_option(sge.abi.OPT_FUSE_BLAS_REFIT, 0)  # this test is about blas_refit_kernel (the fused form: test_fused_skin_and_refit)
        for step in range(3):
            gpu.tick(stages=sge.abi.STAGE_ALL | sge.abi.STAGE_BLAS_REFIT)  # refit enqueued behind the skinning of the same step
            pos, _, _ = gpu.skinned()
            got = gpu.blas_bounds()
            for c in range(n):
                want = expected_bounds(topo, gpu.mesh["indices"], pos[c * V:(c + 1) * V])
                assert np.array_equal(got[c], want), (step, c, np.argwhere(got[c] != want)[:4])
        # a refit on its own, over a sub-range: only those characters' rows change
        before = gpu.blas_bounds()
        gpu.tick(stages=sge.abi.STAGE_ALL)
        gpu.blas_refit(first=1, count=2)
        pos, _, _ = gpu.skinned()
        got = gpu.blas_bounds()
        for c in range(n):
            if c in (1, 2):
                assert np.array_equal(got[c], expected_bounds(topo, gpu.mesh["indices"], pos[c * V:(c + 1) * V]))
            else:
                assert np.array_equal(got[c], before[c])
        # idempotent
        gpu.blas_refit()
        a = gpu.blas_bounds()
        gpu.blas_refit()
        assert np.array_equal(a, gpu.blas_bounds())
    finally:
        gpu.close()


@pytest.mark.parametrize("real", [False, True])
@pytest.mark.parametrize("layout", ["packed", "padded16"])
def test_fused_skin_and_refit(sge, real, layout):
    """SGE_OPT_FUSE_BLAS_REFIT: one kernel skins and reduces the boxes from the positions it has just computed. The boxes
    must be the exact min / max of the positions it stored; the streams must match the two-kernel path."""
    gpu = sge.CharacterEngine(0)
    try:
        n = 4 if real else 7
        gpu.set_option(sge.abi.OPT_SKIN_LAYOUT, sge.abi.LAYOUT_PADDED16 if layout == "padded16" else sge.abi.LAYOUT_PACKED)
        _scene(sge, gpu, n, real, mixed=True, seed=9)
        gpu.blas_build(gpu.mesh["indices"])
        topo = _topology(sge, gpu)
        V = gpu.vertex_count
        st = sge.abi.STAGE_ALL | sge.abi.STAGE_BLAS_REFIT
        gpu.set_option(sge.abi.OPT_FUSE_BLAS_REFIT, 0)  # the two-kernel path is the reference here
        for _ in range(3):
            gpu.tick(stages=st)
        ref = [a.copy() for a in gpu.skinned()]
        ref_boxes = gpu.blas_bounds()
        gpu.set_option(sge.abi.OPT_FUSE_BLAS_REFIT, 1)
        gpu.tick(dt=0.0, stages=sge.abi.STAGE_SKIN | sge.abi.STAGE_BLAS_REFIT)  # same palettes, fused kernel
        got = gpu.skinned()
        boxes = gpu.blas_bounds()
        for g, r in zip(got, ref):
            assert np.abs(g - r).max() <= 1e-6 * max(1.0, np.abs(r).max())
        for c in range(n):
            assert np.array_equal(boxes[c], expected_bounds(topo, gpu.mesh["indices"], got[0][c * V:(c + 1) * V])), c
        assert np.abs(boxes - ref_boxes).max() <= 1e-6 * np.abs(ref_boxes).max()
        # a sub-range through the fused path, then a few whole steps
        gpu.tick(stages=st, first=1, count=2)
        for _ in range(2):
            gpu.tick(stages=st)
        pos = gpu.skinned()[0]
        boxes = gpu.blas_bounds()
        for c in range(n):
            assert np.array_equal(boxes[c], expected_bounds(topo, gpu.mesh["indices"], pos[c * V:(c + 1) * V])), c
    finally:
        gpu.close()


def test_refit_over_caller_buffers(sge):
    """sge_blas_refit_buffers: the same kernel over device buffers the caller owns (here: the context's own streams,
    read through sge_crowd_buffers, written to a separately allocated box table)."""
    import torch
    gpu = sge.CharacterEngine(0)
    try:
        n = 4
        _scene(sge, gpu, n, False)
        info = gpu.blas_build(gpu.mesh["indices"])
        gpu.tick(stages=sge.abi.STAGE_ALL)
        gpu.synchronize()
        op = C.c_void_p()
        gpu._call("crowd_buffers", None, C.byref(op), None, None)
        out = torch.zeros((n - 1, info.entryCount + 1, 6), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        gpu._call("blas_refit_buffers", op, sge.abi.LAYOUT_PACKED, gpu.vertex_count, n - 1, C.c_void_p(out.data_ptr()))
        gpu.synchronize()
        gpu.blas_refit()
        assert np.array_equal(out.cpu().numpy(), gpu.blas_bounds()[1:])
    finally:
        gpu.close()


def _rotation(axis, angle):
    a = np.asarray(axis, np.float64); a /= np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return (np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K).astype(np.float32)


def _rays_at(rng, lo, hi, k):
    """k rays from a shell around the box [lo, hi] towards random points inside it."""
    centre, half = (lo + hi) / 2, (hi - lo) / 2
    d = rng.normal(size=(k, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    origins = centre + d * (np.linalg.norm(half) * rng.uniform(1.2, 3.0, (k, 1)))
    targets = centre + rng.uniform(-1, 1, (k, 3)) * half
    dirs = targets - origins
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    return origins.astype(np.float32), dirs.astype(np.float32)


@pytest.mark.parametrize("real,layout", [(False, "packed"), (True, "packed"), (False, "padded16")])
def test_closest_hit_matches_the_triangle_scan(sge, real, layout):
    gpu, cpu = sge.CharacterEngine(0), ob.oracle_engine()
    try:
        n = 4
        gpu.set_option(sge.abi.OPT_SKIN_LAYOUT, sge.abi.LAYOUT_PADDED16 if layout == "padded16" else sge.abi.LAYOUT_PACKED)
        for e in (gpu, cpu):
            _scene(sge, e, n, real, mixed=True, seed=5)
            e.blas_build(e.mesh["indices"])
        uvs = gpu.mesh["uvs"] if "uvs" in gpu.mesh else np.random.default_rng(2).uniform(0, 1, (gpu.vertex_count, 2)).astype(np.float32)
        for e in (gpu, cpu):
            e.blas_set_uvs(uvs)
        for _ in range(4):
            gpu.tick(stages=sge.abi.STAGE_ALL | sge.abi.STAGE_BLAS_REFIT)
        gp, gn, gt = gpu.skinned()
        ob.skinned_upload(cpu, gp, gn, gt)  # both sides look at the same skinned vertices
        # instance matrices (item.modelMatrix): identity, translation, rotation + translation, rotation + non-uniform scale
        rng = np.random.default_rng(11)
        mats = np.zeros((n, 4, 4), np.float32)
        mats[0] = np.eye(4)
        mats[1] = np.eye(4); mats[1][:3, 3] = (7, -2, 3)
        mats[2] = np.eye(4); mats[2][:3, :3] = _rotation((0.3, 0.8, -0.5), 1.1)
        mats[2][:3, 3] = (-4, 1, 9)
        mats[3] = np.eye(4); mats[3][:3, :3] = np.diag([1.5, 0.75, 2.0]) @ np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32)
        mats[3][:3, 3] = (1, 2, 3)
        cols = np.ascontiguousarray(mats.transpose(0, 2, 1)).reshape(n, 16)  # column-major
        for e in (gpu, cpu):
            e.blas_instances(cols)
        V = gpu.vertex_count
        O, D, I = [], [], []
        for c in range(n):
            world = gp[c * V:(c + 1) * V] @ mats[c][:3, :3].T + mats[c][:3, 3]
            o, d = _rays_at(rng, world.min(0), world.max(0), 600)
            O.append(o); D.append(d); I.append(np.full(len(o), c, np.int32))
            # exactly through vertices and edge midpoints: several triangles report (nearly) the same distance
            tri = gpu.mesh["indices"].reshape(-1, 3)[rng.integers(0, len(gpu.mesh["indices"]) // 3, 60)]
            tgt = np.concatenate([world[tri[:30, 0]], (world[tri[30:, 0]] + world[tri[30:, 1]]) / 2]).astype(np.float32)
            o2 = (world.mean(0) + (tgt - world.mean(0)) * 4).astype(np.float32)
            d2 = tgt - o2
            O.append(o2); D.append((d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)); I.append(np.full(len(o2), c, np.int32))
        O, D, I = np.concatenate(O), np.concatenate(D), np.concatenate(I)
        # axis-parallel rays (zero direction components) and rays that start inside the character
        O = np.concatenate([O, gp[:8] @ mats[0][:3, :3].T + (0, 5, 0), gp[100:108] * 0.5]).astype(np.float32)
        D = np.concatenate([D, np.tile(np.array([0, -1, 0], np.float32), (8, 1)), np.tile(np.array([1, 0, 0], np.float32), (8, 1))])
        I = np.concatenate([I, np.zeros(16, np.int32)])
        g = gpu.blas_intersect(O, D, I)
        c_ = cpu.blas_intersect(O, D, I)
        assert g["hit"].sum() > 0.2 * len(O), "the rays are aimed at the character"
        assert (g["hit"] == 0).any()
        assert np.array_equal(g["hit"], c_["hit"])
        assert np.array_equal(g["primitive"], c_["primitive"]), np.argwhere(g["primitive"] != c_["primitive"])[:5]
        assert np.abs(g["uv"][g["hit"] == 1]).max() > 0.1
        for f in ("distance", "bary", "geomNormal", "normal", "tangent", "bitangent", "uv"):
            assert np.abs(g[f] - c_[f]).max() <= 1e-6 * max(1.0, np.abs(c_[f]).max()), f
        # distance limits
        lim = np.where(g["hit"] == 1, g["distance"] * 0.5, 1.0).astype(np.float32)
        assert gpu.blas_intersect(O, D, I, max_distance=lim)["distance"].max() <= lim.max()
        g2, c2 = gpu.blas_intersect(O, D, I, min_distance=lim), cpu.blas_intersect(O, D, I, min_distance=lim)
        assert np.array_equal(g2["primitive"], c2["primitive"]) and np.array_equal(g2["hit"], c2["hit"])
        assert (g2["distance"][g2["hit"] == 1] >= lim[g2["hit"] == 1]).all()
        assert np.array_equal(g["instance"], np.where(g["hit"] == 1, I, -1))
        # out-of-range instance = miss
        assert gpu.blas_intersect(O[:2], D[:2], [n, n + 7])["hit"].tolist() == [0, 0]
        # instance < 0: the instance level — closest hit over ALL characters (the reference's TLAS), against the oracle's loop
        # over every character and every triangle. Characters 0 and 1 are then put at the same place: the smaller index wins.
        A = np.full(len(O), -1, np.int32)
        ga, ca = gpu.blas_intersect(O, D, A), cpu.blas_intersect(O, D, A)
        for f in ("hit", "primitive", "instance"):
            assert np.array_equal(ga[f], ca[f]), (f, np.argwhere(ga[f] != ca[f])[:5])
        assert np.abs(ga["distance"] - ca["distance"]).max() <= 1e-6 * max(1.0, np.abs(ca["distance"]).max())
        assert (ga["hit"] >= g["hit"]).all() and (ga["distance"][g["hit"] == 1] <= g["distance"][g["hit"] == 1]).all()
        assert len(np.unique(ga["instance"][ga["hit"] == 1])) == n, "rays aimed at every character find it"
        # the device-memory entry point gives the same records
        import torch
        r = np.zeros(len(O), sge.abi.blas_ray_dtype)
        r["origin"], r["direction"], r["minDistance"], r["maxDistance"], r["instance"] = O, D, 0.001, 1e6, A
        d_r = torch.from_numpy(r.view(np.uint8).copy()).to("cuda:0")
        d_h = torch.zeros(len(O) * sge.abi.blas_hit_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        gpu.blas_intersect_device(d_r.data_ptr(), len(O), d_h.data_ptr())
        gpu.synchronize()
        hd = d_h.cpu().numpy().view(sge.abi.blas_hit_dtype)
        for f in hd.dtype.names:
            assert np.array_equal(hd[f], ga[f]), f
        # a caller that promises "no ray asks for every instance" and breaks the promise gets misses for those rays, not a walk
        # over instance boxes this launch did not refresh; rays that name their character are unaffected
        mixed = r.copy()
        mixed["instance"][::2] = I[::2]
        d_r2 = torch.from_numpy(mixed.view(np.uint8).copy()).to("cuda:0")
        d_h.zero_()
        torch.cuda.synchronize()
        gpu.blas_intersect_device(d_r2.data_ptr(), len(O), d_h.data_ptr(), any_instance=False)
        gpu.synchronize()
        hm = d_h.cpu().numpy().view(sge.abi.blas_hit_dtype)
        assert (hm["hit"][1::2] == 0).all()
        for f in ("hit", "primitive", "instance"):
            assert np.array_equal(hm[f][::2], g[f][::2]), f
        cols[1] = cols[0]
        for e in (gpu, cpu):
            e.blas_instances(cols)
        sel = I <= 1
        gb, cb = gpu.blas_intersect(O[sel], D[sel], A[sel]), cpu.blas_intersect(O[sel], D[sel], A[sel])
        for f in ("hit", "primitive", "instance"):
            assert np.array_equal(gb[f], cb[f]), f
        assert gb["hit"].sum() > 100
    finally:
        gpu.close(); cpu.close()


def _custom_mesh_engine(sge, positions, indices, n):
    """An engine whose 'skinned mesh' is an arbitrary vertex / index set rigidly bound to the root bone; with the bind-pose
    palette (identity) the skinned positions are the source positions, per clone."""
    eng = sge.CharacterEngine(0)
    ybot = sge.assets.YBotAssets()
    eng.upload_skeleton(ybot)
    eng.upload_profiles(ybot.profiles)
    V = len(positions)
    nrm = np.tile(np.array([0, 1, 0], np.float32), (V, 1))
    mesh = {"positions": np.ascontiguousarray(positions, np.float32), "normals": nrm,
            "tangents": np.tile(np.array([1, 0, 0, 1], np.float32), (V, 1)), "boneIndices": np.zeros((V, 4), np.uint16),
            "boneWeights": np.tile(np.array([1, 0, 0, 0], np.float32), (V, 1)), "indices": np.ascontiguousarray(indices, np.uint32)}
    eng.upload_skinned_mesh(mesh)
    eng.resize(n)
    L = sge.assets.default_locomotion(n, ybot)
    L["flags"] = 0  # bind-pose branch: identity palette
    eng.upload(bodies=sge.assets.default_bodies(n, np.zeros((n, 3))), params=sge.assets.default_controller_params(n),
               controllers=sge.assets.default_controller_state(n), intents=sge.assets.default_intents(n), locomotion=L,
               actions=sge.assets.default_actions(n))
    return eng


@pytest.mark.parametrize("case", ["one-triangle", "unreferenced+degenerate", "soup", "soup-fused"])
def test_odd_meshes(sge, case):
    """Edge cases of the topology / schedule: a single triangle; vertices no triangle uses and zero-area triangles; an
    incoherent triangle soup whose vertex count is not a multiple of four and whose one tile has more rounds than the
    workgroup has wavefronts (the on-demand path of the refit kernels)."""
    rng = np.random.default_rng(3)
    if case == "one-triangle":
        pos, idx = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([0, 1, 2], np.uint32)
    elif case == "unreferenced+degenerate":
        pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [9, 9, 9], [2, 2, 2]], np.float32)
        idx = np.array([0, 1, 2, 0, 0, 1, 2, 2, 2, 4, 1, 0], np.uint32)  # vertex 3 unused; two degenerate triangles
    else:
        pos = rng.uniform(-2, 2, (1237, 3)).astype(np.float32)
        idx = rng.integers(0, 1237, 15000).astype(np.uint32)
    n = 3
    gpu = _custom_mesh_engine(sge, pos, idx, n)
    try:
        info = gpu.blas_build(idx)
        topo = sge.CharacterEngine.blas_topology(pos, idx)
        if case.startswith("soup"):
            assert info.incidenceCount / 16 / 64 > 8, "this case is meant to have more than eight rounds in its tile"
        gpu.set_option(sge.abi.OPT_FUSE_BLAS_REFIT, 1 if case == "soup-fused" else 0)
        gpu.tick(dt=0.0, stages=sge.abi.STAGE_POSE | sge.abi.STAGE_SKIN | sge.abi.STAGE_BLAS_REFIT)
        p = gpu.skinned()[0]
        V = len(pos)
        assert np.abs(p[:V] - pos).max() < 1e-5
        b = gpu.blas_bounds()
        for c in range(n):
            assert np.array_equal(b[c], expected_bounds(topo, idx, p[c * V:(c + 1) * V])), c
        used = np.unique(idx)
        assert np.array_equal(b[0, -1, :3], p[:V][used].min(0)) and np.array_equal(b[0, -1, 3:], p[:V][used].max(0))
        if case == "unreferenced+degenerate":
            assert b[0, -1, 3:].max() < 9, "the unreferenced vertex is in no box"
            h = gpu.blas_intersect([[0.25, 0.25, -5]], [[0, 0, 1]], [1])  # triangle 0 at z = 0, triangle 3 just behind it
            assert h["hit"][0] == 1 and h["primitive"][0] == 0 and abs(h["distance"][0] - 5) < 1e-6, "degenerate triangles are never hit"
    finally:
        gpu.close()


def test_state_errors(sge):
    gpu = sge.CharacterEngine(0)
    try:
        _scene(sge, gpu, 2, False)
        with pytest.raises(sge.SgeError, match="sge_blas_build"):
            gpu.blas_refit(0, 2)
        with pytest.raises(sge.SgeError, match="sge_blas_build"):
            gpu.tick(stages=sge.abi.STAGE_ALL | sge.abi.STAGE_BLAS_REFIT)
        with pytest.raises(sge.SgeError, match="out of range"):
            gpu.blas_build(np.array([0, 1, 10 ** 6], np.uint32))
        gpu.blas_build(gpu.mesh["indices"])
        gpu.tick(stages=sge.abi.STAGE_ALL | sge.abi.STAGE_BLAS_REFIT)
        with pytest.raises(sge.SgeError):
            gpu.blas_refit(1, 5)
        # a new mesh invalidates the structure built for the old one
        _scene(sge, gpu, 2, False)
        with pytest.raises(sge.SgeError, match="sge_blas_build"):
            gpu.blas_refit(0, 2)
        # resizing the crowd keeps it
        gpu.blas_build(gpu.mesh["indices"])
        gpu.resize(3)
        gpu.tick(dt=0.0, stages=sge.abi.STAGE_POSE | sge.abi.STAGE_SKIN | sge.abi.STAGE_BLAS_REFIT)
        assert gpu.blas_bounds().shape[0] == 3
    finally:
        gpu.close()


def test_full_size_refit_properties(sge):
    """BASELINE configs[2] size (10k characters): every character's root box is the min/max of its skinned vertices, and
    every entry's box lies inside its parent's (checked on a sample of characters; the whole table for containment)."""
    gpu = sge.CharacterEngine(0)
    try:
        n = 10000
        ybot = sge.assets.YBotAssets()
        sge.crowd.upload_character_assets(gpu, ybot)
        terrain = sge.crowd.upload_terrain(gpu)
        sge.crowd.spawn_crowd(gpu, ybot, n, terrain, seed=3, mode="ccd", mixed=True)
        info = gpu.blas_build(gpu.mesh["indices"])
        topo = _topology(sge, gpu)
        for fused in (0, 1):  # two launches (blas_refit_kernel), then the default: folded into the LBS kernel (skin_refit_kernel)
            gpu.set_option(sge.abi.OPT_FUSE_BLAS_REFIT, fused)
            for _ in range(3 if fused == 0 else 2):
                gpu.tick(stages=sge.abi.STAGE_ALL | sge.abi.STAGE_BLAS_REFIT)
            V = gpu.vertex_count
            b = gpu.blas_bounds()
            assert np.isfinite(b).all() and (b[:, :, :3] <= b[:, :, 3:]).all()
            used = np.unique(gpu.mesh["indices"])
            for c in (0, 1, 4999, 9999):
                p = gpu.skinned(first_vertex=c * V, vertex_count=V, normals=False, tangents=False)[0]
                assert np.array_equal(b[c, -1, :3], p[used].min(0)) and np.array_equal(b[c, -1, 3:], p[used].max(0))
                assert np.array_equal(b[c], expected_bounds(topo, gpu.mesh["indices"], p))
            # the fused kernel hands characters out through a ticket counter: a wider sample of root boxes against this step's positions
            for c in np.random.default_rng(17 + fused).integers(0, n, 48):
                p = gpu.skinned(first_vertex=int(c) * V, vertex_count=V, normals=False, tangents=False)[0]
                assert np.array_equal(b[c, -1, :3], p[used].min(0)) and np.array_equal(b[c, -1, 3:], p[used].max(0)), c
            link, first, parent = topo["entryLink"], topo["wideFirst"], topo["wideParentEntry"]
            for w in range(info.wideCount):
                dst = info.entryCount if parent[w] < 0 else parent[w]
                rows = slice(first[w], first[w + 1])
                assert np.array_equal(b[:, rows, :3].min(1), b[:, dst, :3]) and np.array_equal(b[:, rows, 3:].max(1), b[:, dst, 3:])
    finally:
        gpu.close()


def test_instance_level_of_31250_characters_matches_the_flat_scan(sge, monkeypatch):
    """The instance level of a ray that names no character (the reference rebuilds its TLAS over all items per frame,
    RTAccelerationBuilder.swift:168-185) at one GPU's share of configs[3]: 31,250 characters. The grid-ordered three-level form
    (characters sorted by XZ cell, 64 per group, 64 groups per super-group) against the flat scan over groups of 64 consecutive
    indices (SGE_BLAS_FLAT_INSTANCES, round 3's form): hit / instance / primitive / distance identical for 4,096 rays — rays into
    the crowd, grazing rays along rows, rays that miss everything — including a tie: two characters far apart in index stand in
    the same place in the same pose, the smaller index must win whichever the traversal meets first."""
    gpu = sge.CharacterEngine(0)
    try:
        n = 31250
        ybot, _, st = build_scene(sge, gpu, n, terrain_cells=None, mode="lbs", rings=3, segments=3, pose_debug=False)
        # characters 77 and 20,000 become the same character (same body, same clocks): identical skinned vertices
        twin = gpu.download(first=77, count=1)
        gpu.upload(first=20000, **{k: twin[k] for k in ("bodies", "locomotion", "actions", "controllers")})
        gpu.blas_build(gpu.mesh["indices"])
        lbs = sge.abi.STAGE_LOCOMOTION | sge.abi.STAGE_ACTION | sge.abi.STAGE_POSE | sge.abi.STAGE_WRITEBACK | sge.abi.STAGE_SKIN | sge.abi.STAGE_BLAS_REFIT
        for _ in range(3):
            gpu.tick(stages=lbs)
        V = gpu.vertex_count
        b = gpu.download(what=("bodies",))["bodies"]["position"].astype(np.float32)
        a77, a20k = gpu.skinned(77 * V, V)[0], gpu.skinned(20000 * V, V)[0]
        assert np.array_equal(a77, a20k)
        rng = np.random.default_rng(8)
        k = 4096
        P = gpu.skinned(normals=False, tangents=False)[0]                # every skinned vertex of the crowd
        target = P[rng.integers(0, len(P), k)] + rng.normal(0, 0.05, (k, 3)).astype(np.float32)
        lo, hi = P.min(0) - 3, P.max(0) + 3
        origin = np.stack([rng.uniform(lo[0] - 40, hi[0] + 40, k), rng.uniform(5, 60, k), rng.uniform(lo[2] - 40, hi[2] + 40, k)], 1).astype(np.float32)
        origin[:512, 1] = target[:512, 1]                            # grazing: level rays along the rows
        target[:64] = a77[rng.integers(0, V, 64)]                     # at the twins, from just above them (nobody else in the way)
        origin[:64] = target[:64] + rng.uniform(-0.4, 0.4, (64, 3)).astype(np.float32) + (0, 6, 0)
        origin[64:128] = origin[64:128] * (1, 0, 1) + (0, 500, 0)    # far above, straight down the middle of nothing
        target[64:128] = origin[64:128] - (0, 1, 0) + (1e4, 0, 0)
        d = target - origin
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        inst = np.full(k, -1, np.int32)
        tree = gpu.blas_intersect(origin, d, inst)
        monkeypatch.setenv("SGE_BLAS_FLAT_INSTANCES", "1")
        flat = gpu.blas_intersect(origin, d, inst)
        monkeypatch.delenv("SGE_BLAS_FLAT_INSTANCES")
        for f in ("hit", "instance", "primitive"):
            assert np.array_equal(tree[f], flat[f]), (f, np.argwhere(tree[f] != flat[f])[:5])
        assert np.array_equal(tree["distance"].view(np.uint32), flat["distance"].view(np.uint32))
        assert tree["hit"].mean() > 0.15 and (tree["hit"] == 0).sum() >= 32, tree["hit"].mean()
        assert len(np.unique(tree["instance"][tree["hit"] == 1])) > 300
        # the twins: a ray that names character 77 and one that names character 20,000 report the same hit bit for bit; the ray
        # that names nobody reports character 77 wherever that hit is the closest, never character 20,000
        n77 = gpu.blas_intersect(origin[:64], d[:64], np.full(64, 77, np.int32))
        n20k = gpu.blas_intersect(origin[:64], d[:64], np.full(64, 20000, np.int32))
        assert np.array_equal(n77["hit"], n20k["hit"]) and np.array_equal(n77["primitive"], n20k["primitive"])
        assert np.array_equal(n77["distance"].view(np.uint32), n20k["distance"].view(np.uint32)) and n77["hit"].sum() >= 16
        first = (n77["hit"] == 1) & (tree["distance"][:64] == n77["distance"])
        assert first.sum() >= 4 and (tree["instance"][:64][first] == 77).all(), "at equal distance the smaller character index wins"
        assert not (tree["instance"] == 20000).any()
        # against a ray that names its character: the all-instances answer is never farther
        named = gpu.blas_intersect(origin[:512], d[:512], np.where(tree["hit"][:512] == 1, tree["instance"][:512], 0).astype(np.int32))
        sel = tree["hit"][:512] == 1
        assert np.array_equal(named["primitive"][sel], tree["primitive"][:512][sel]) and np.array_equal(named["distance"][sel], tree["distance"][:512][sel])
    finally:
        gpu.close()
