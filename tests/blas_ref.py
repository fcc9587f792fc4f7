"""TEST INFRASTRUCTURE: what a refit must produce, computed without the product's kernels.

`expected_bounds` follows the definition of the structure (include/sge_amd.h): the box of a leaf entry is the min/max of
the vertices of its triangles, the box of an inner entry the union of its wide node's entries, the last row the union
of the root's entries.  Plain numpy over the index buffer: min/max are exact, so the comparison is value-exact.
"""
import numpy as np


def expected_bounds(topo, indices, positions):
    """positions [V][3] of ONE character -> float32 [entryCount + 1][6]."""
    link, first, parent = topo["entryLink"], topo["wideFirst"], topo["wideParentEntry"]
    tri = np.asarray(indices, np.int64).reshape(-1, 3)
    E = link.shape[0]
    out = np.zeros((E + 1, 6), np.float32)
    done = np.zeros(E + 1, bool)
    for e in range(E):
        if link[e, 0] < 0:
            s0 = ~int(link[e, 0])
            prims = topo["slotTriangle"][s0:s0 + int(link[e, 1])].astype(np.int64)
            v = positions[tri[prims].reshape(-1)]
            out[e, :3], out[e, 3:] = v.min(0), v.max(0)
            done[e] = True
    for w in range(len(parent) - 1, -1, -1):  # children are numbered after their parent
        rows = np.arange(first[w], first[w + 1])
        assert done[rows].all()
        dst = E if parent[w] < 0 else parent[w]
        out[dst, :3], out[dst, 3:] = out[rows, :3].min(0), out[rows, 3:].max(0)
        done[dst] = True
    assert done.all()
    return out


def check_topology(topo, vertex_count, indices, width=64, cluster=64):
    """Structural invariants of sge_blas_topology's result."""
    info, link, first, parent = topo["info"], topo["entryLink"], topo["wideFirst"], topo["wideParentEntry"]
    T = len(indices) // 3
    assert info.triangleCount == T and info.entryCount == link.shape[0] and info.wideCount == len(parent)
    assert sorted(topo["slotTriangle"].tolist()) == list(range(T)), "every triangle sits at exactly one slot"
    leaf = link[:, 0] < 0
    assert info.clusterCount == int(leaf.sum())
    counts = link[leaf, 1]
    assert counts.min() >= 1 and counts.max() <= cluster and counts.sum() == T
    starts = np.sort(~link[leaf, 0])
    assert starts[0] == 0 and np.array_equal(np.sort(~link[leaf, 0] + counts), np.append(starts[1:], T)), "clusters tile the slots"
    per_node = np.diff(first)
    assert first[0] == 0 and first[-1] == link.shape[0] and per_node.min() >= 1 and per_node.max() <= width
    # inner entries point at later wide nodes, each wide node (but the root) has exactly one parent entry
    inner = np.flatnonzero(~leaf)
    assert np.array_equal(np.sort(link[inner, 0]), np.arange(1, info.wideCount))
    assert parent[0] == -1
    for e in inner:
        w = link[e, 0]
        assert parent[w] == e and first[w] > e, "children come after their parent"
    # vertex -> clusters CSR == the incidence implied by the index buffer
    tri = np.asarray(indices, np.int64).reshape(-1, 3)
    start, ents = topo["vertexEntryStart"], topo["vertexEntries"]
    assert start[0] == 0 and start[-1] == len(ents) == info.incidenceCount and len(start) == vertex_count + 1
    pairs = set()
    for e in np.flatnonzero(leaf):
        s0 = ~int(link[e, 0])
        for v in np.unique(tri[topo["slotTriangle"][s0:s0 + int(link[e, 1])].astype(np.int64)]):
            pairs.add((int(v), int(e)))
    got = {(v, int(e)) for v in range(vertex_count) for e in ents[start[v]:start[v + 1]]}
    assert got == pairs
    return True
