"""CPU tests of the skinned-geometry acceleration structure's host side: the topology builder behind sge_blas_build
(exposed as the context-free helper sge_blas_topology) and the oracle's brute-force ray scan on known answers."""
import numpy as np
import pytest

import oracle_binding as ob
from blas_ref import check_topology, expected_bounds


def _grid(nx, nz):
    xs, zs = np.meshgrid(np.arange(nx + 1, dtype=np.float32), np.arange(nz + 1, dtype=np.float32), indexing="ij")
    pos = np.stack([xs, np.sin(xs * 0.7) * np.cos(zs * 0.3), zs], -1).reshape(-1, 3).astype(np.float32)
    idx = []
    for i in range(nx):
        for j in range(nz):
            a, b, c, d = i * (nz + 1) + j, (i + 1) * (nz + 1) + j, i * (nz + 1) + j + 1, (i + 1) * (nz + 1) + j + 1
            idx += [a, b, c, c, b, d]
    return pos, np.asarray(idx, np.uint32)


@pytest.mark.parametrize("shape", [(1, 1), (4, 8), (8, 4), (33, 17), (64, 64), (150, 160)])
def test_topology_invariants(sge, shape):
    pos, idx = _grid(*shape)
    topo = sge.CharacterEngine.blas_topology(pos, idx)
    check_topology(topo, len(pos), idx)
    info = topo["info"]
    T = len(idx) // 3
    assert info.clusterCount == -(-T // 64), "all clusters but the last are full"
    assert info.levels == (1 if info.clusterCount <= 64 else 2)
    b = expected_bounds(topo, idx, pos)
    assert np.array_equal(b[-1, :3], pos[np.unique(idx)].min(0)) and np.array_equal(b[-1, 3:], pos[np.unique(idx)].max(0))
    # deterministic
    again = sge.CharacterEngine.blas_topology(pos, idx)
    for k in ("entryLink", "wideFirst", "wideParentEntry", "slotTriangle", "vertexEntryStart", "vertexEntries"):
        assert np.array_equal(topo[k], again[k]), k


def test_topology_of_the_real_ybot(sge):
    z = np.load(sge.assets.GOLDEN_DIR + "/ybot_skinned.npz")
    pos, idx = z["positions"].reshape(-1, 3), z["indices"].astype(np.uint32)
    topo = sge.CharacterEngine.blas_topology(pos, idx)
    check_topology(topo, len(pos), idx)
    info = topo["info"]
    assert (info.triangleCount, info.clusterCount, info.wideCount, info.levels) == (55320, 865, 17, 2)
    # spatial quality: the clusters' boxes are small next to the character (sum of cluster box areas / root box area)
    b = expected_bounds(topo, idx, pos)
    ext = b[:, 3:] - b[:, :3]
    area = 2 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0])
    leaf = topo["entryLink"][:, 0] < 0
    assert area[:-1][leaf].mean() < 0.01 * area[-1]


def test_topology_rejects_bad_input(sge):
    pos, idx = _grid(2, 2)
    lib = sge.abi.load_library()
    info = sge.abi.BlasInfo()
    P = sge.abi.ptr
    bad = idx.copy(); bad[3] = 1000
    assert lib.sge_blas_topology(P(pos), len(pos), P(bad), len(bad), info, None, None, None, None, None, None) == sge.abi.SGE_ERR_INVALID
    assert b"out of range" in lib.sge_last_error()
    assert lib.sge_blas_topology(P(pos), len(pos), P(idx), 4, info, None, None, None, None, None, None) == sge.abi.SGE_ERR_INVALID
    assert lib.sge_blas_topology(P(pos), len(pos), P(idx), 0, info, None, None, None, None, None, None) == sge.abi.SGE_ERR_INVALID
    assert lib.sge_blas_topology(None, 0, P(idx), len(idx), info, None, None, None, None, None, None) == sge.abi.SGE_ERR_INVALID


def test_oracle_ray_scan_known_answers(sge):
    """One character whose 'skinned' streams are set by hand: a unit quad at y = 0 facing +y, tangent +x."""
    cpu = ob.oracle_engine()
    ybot = sge.assets.YBotAssets()
    cpu.upload_skeleton(ybot)
    cpu.upload_profiles(ybot.profiles)
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1], [1, 0, 1]], np.float32)
    nrm = np.tile(np.array([0, 1, 0], np.float32), (4, 1))
    mesh = {"positions": pos, "normals": nrm, "tangents": np.tile(np.array([1, 0, 0, 1], np.float32), (4, 1)),
            "boneIndices": np.zeros((4, 4), np.uint16), "boneWeights": np.tile(np.array([1, 0, 0, 0], np.float32), (4, 1))}
    cpu.upload_skinned_mesh(mesh)
    cpu.resize(2)
    idx = np.array([0, 2, 1, 1, 2, 3], np.uint32)  # counter-clockwise seen from +y
    cpu.blas_build(idx)
    tan = mesh["tangents"]
    ob.skinned_upload(cpu, np.tile(pos, (2, 1)), np.tile(nrm, (2, 1)), np.tile(tan, (2, 1)))
    m = np.eye(4, dtype=np.float32); m[:3, 3] = (10, 2, 0)  # column-major: translation in the last column
    cpu.blas_instances(np.stack([np.eye(4, dtype=np.float32).T.reshape(16), m.T.reshape(16)]))
    o = np.array([[0.25, 3, 0.25], [0.75, 3, 0.75], [10.25, 5, 0.25], [0.25, 3, 0.25], [5, 3, 5], [0.25, -3, 0.25], [0.25, 3, 0.25]], np.float32)
    d = np.array([[0, -1, 0]] * 5 + [[0, 1, 0]] + [[0, -1, 0]], np.float32)
    inst = np.array([0, 0, 1, 1, 0, 0, 0], np.int32)
    h = cpu.blas_intersect(o, d, inst, max_distance=[1e6] * 6 + [2.0])
    assert h["hit"].tolist() == [1, 1, 1, 0, 0, 1, 0]
    assert h["primitive"].tolist() == [0, 1, 0, -1, -1, 0, -1]
    assert np.allclose(h["distance"][[0, 1, 2, 5]], 3.0)
    # triangle 0 = (v0, v2, v1): bary = weights of its 2nd and 3rd vertex = (z, x) of the hit point
    assert np.allclose(h["bary"][0], (0.25, 0.25)) and np.allclose(h["bary"][2], (0.25, 0.25))
    assert np.allclose(h["geomNormal"][0], (0, 1, 0)) and np.allclose(h["geomNormal"][5], (0, -1, 0)), "flipped against the ray"
    assert np.allclose(h["normal"][0], (0, 1, 0)) and np.allclose(h["tangent"][0], (1, 0, 0))
    assert np.allclose(h["bitangent"][0], np.cross((0, 1, 0), (1, 0, 0)))
    assert h["instance"].tolist() == [0, 0, 1, -1, -1, 0, -1]
    assert not h["uv"].any(), "no uvs uploaded: (0, 0)"
    cpu.blas_set_uvs(pos[:, [0, 2]] * 10)   # uv = 10 * (x, z): interp_uv of the hit point
    hu = cpu.blas_intersect(o, d, inst, max_distance=[1e6] * 6 + [2.0])
    assert np.allclose(hu["uv"][0], (2.5, 2.5)) and np.allclose(hu["uv"][1], (7.5, 7.5)) and np.allclose(hu["uv"][2], (2.5, 2.5))
    # instance < 0: every character; ray 3 (above character 0, asked for character 1 before) now finds character 0
    a = cpu.blas_intersect(o, d, np.full(len(o), -1, np.int32), max_distance=[1e6] * 6 + [2.0])
    assert a["hit"].tolist() == [1, 1, 1, 1, 0, 1, 0] and a["instance"].tolist() == [0, 0, 1, 0, -1, 0, -1]
    cpu.close()
