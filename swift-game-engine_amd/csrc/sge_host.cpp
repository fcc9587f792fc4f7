// Host-side (CPU) parts of libsge_amd.so that run once at load / rebuild time:
//   sge_skeleton_build          SkeletonLoader.buildSkeleton  (Game/SkeletonLoader.swift:28-87)
//                               + Skeleton.init invBindModel  (Game/Skeleton.swift:153-156)
//   sge_mesh_tangents_compute   MeshTangents.compute          (Game/MeshTangents.swift:10-83)
//   HostCollision::rebuild      TriangleMeshSet.rebuild       (Game/CollisionQuery.swift:331-417)
//                               + BVH.build                   (Game/CollisionQuery.swift:577-706)
// The BVH must reproduce the reference's triOrder exactly: first-hit selection
// and capsuleOverlapAll's "first maxHits" depend on the order in which its
// right-child-first traversal reaches triangles; that order is flattened here
// into a per-triangle visit rank the kernels use as a tie-breaker.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include "sge_internal.hpp"

namespace sge {

struct M4 { float m[16]; }; // column-major

static M4 m4FromAff(const Aff& a) {
    M4 r;
    const F3 c[4] = {a.c0, a.c1, a.c2, a.c3};
    for (int j = 0; j < 4; ++j) {
        r.m[j * 4 + 0] = c[j].x; r.m[j * 4 + 1] = c[j].y; r.m[j * 4 + 2] = c[j].z; r.m[j * 4 + 3] = j == 3 ? 1.0f : 0.0f;
    }
    return r;
}

// simd_inverse(float4x4) restated as adjugate / determinant.
static M4 m4Inverse(const M4& in) {
    const float* m = in.m;
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    float id = 1.0f / det;
    M4 r;
    for (int i = 0; i < 16; ++i) r.m[i] = inv[i] * id;
    return r;
}

// ---- collision world ----------------------------------------------------- //

static inline float axisOf(const float* v, int axis) { return v[axis]; }

struct Builder {
    HostCollision& hc;
    const float* aabb(int t) const { return &hc.aabbs[(size_t)t * 6]; }
    void centroid(int t, float c[3]) const { // (min + max) * 0.5
        const float* b = aabb(t);
        for (int k = 0; k < 3; ++k) c[k] = (b[k] + b[3 + k]) * 0.5f;
    }
    void rangeBounds(int start, int count, float mn[3], float mx[3]) const {
        const float* f = aabb(hc.triOrder[start]);
        for (int k = 0; k < 3; ++k) { mn[k] = f[k]; mx[k] = f[3 + k]; }
        for (int i = 1; i < count; ++i) {
            const float* b = aabb(hc.triOrder[start + i]);
            for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], b[k]); mx[k] = fmaxf(mx[k], b[3 + k]); }
        }
    }
    int build(int start, int count, int parent, int depth) {
        int nodeIndex = (int)hc.nodes.size();
        HostBVHNode n;
        rangeBounds(start, count, n.mn, n.mx);
        n.left = n.right = -1; n.start = start; n.count = count; n.parent = parent;
        hc.nodes.push_back(n);
        hc.maxDepth = std::max(hc.maxDepth, depth);
        if (count <= 4) { // StaticTriMesh.leafTriangleLimit
            for (int i = 0; i < count; ++i) hc.triLeaf[hc.triOrder[start + i]] = nodeIndex;
            return nodeIndex;
        }
        float cmn[3], cmx[3], c[3];
        centroid(hc.triOrder[start], c);
        for (int k = 0; k < 3; ++k) cmn[k] = cmx[k] = c[k];
        for (int i = 1; i < count; ++i) {
            centroid(hc.triOrder[start + i], c);
            for (int k = 0; k < 3; ++k) { cmn[k] = fminf(cmn[k], c[k]); cmx[k] = fmaxf(cmx[k], c[k]); }
        }
        float ex = cmx[0] - cmn[0], ey = cmx[1] - cmn[1], ez = cmx[2] - cmn[2];
        int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
        float pivot = (cmn[axis] + cmx[axis]) * 0.5f;
        int i = start, j = start + count - 1;
        while (i <= j) {
            centroid(hc.triOrder[i], c);
            if (c[axis] < pivot) {
                ++i;
            } else {
                std::swap(hc.triOrder[i], hc.triOrder[j]);
                --j;
            }
        }
        int end = start + count;
        if (i == start || i == end) {
            // Swift's sort is stable (merge sort since Swift 5).
            std::stable_sort(hc.triOrder.begin() + start, hc.triOrder.begin() + end, [&](int a, int b) {
                float ca[3], cb[3];
                centroid(a, ca); centroid(b, cb);
                return ca[axis] < cb[axis];
            });
            i = start + count / 2;
        }
        int mid = i;
        int left = build(start, mid - start, nodeIndex, depth + 1);
        int right = build(mid, end - mid, nodeIndex, depth + 1);
        HostBVHNode& me = hc.nodes[nodeIndex];
        me.left = left; me.right = right; me.start = 0; me.count = 0;
        for (int k = 0; k < 3; ++k) {
            me.mn[k] = fminf(hc.nodes[left].mn[k], hc.nodes[right].mn[k]);
            me.mx[k] = fmaxf(hc.nodes[left].mx[k], hc.nodes[right].mx[k]);
        }
        return nodeIndex;
    }
};

void HostCollision::rebuild(const sge_static_mesh_entity* ents, int count) {
    positions.clear(); localPositions.clear(); indices.clear(); aabbs.clear(); materials.clear(); layers.clear();
    nodes.clear(); triOrder.clear(); triLeaf.clear(); rank.clear(); wide.clear(); wideBinary.clear();
    slices.assign(count > 0 ? count : 0, HostMeshSlice{});
    root = -1; maxDepth = 0;
    const float areaEps = 1e-10f;
    for (int e = 0; e < count; ++e) {
        const sge_static_mesh_entity& m = ents[e];
        const float* M = m.modelMatrix;
        uint32_t baseVertex = (uint32_t)(positions.size() / 3);
        const int indexStart = (int)indices.size(), triStart = (int)layers.size();
        for (int v = 0; v < m.vertexCount; ++v) {
            float x = m.positions[v * 3], y = m.positions[v * 3 + 1], z = m.positions[v * 3 + 2];
            for (int r = 0; r < 3; ++r) // simd_mul(modelMatrix, (p,1)).xyz
                positions.push_back(((M[r] * x + M[4 + r] * y) + M[8 + r] * z) + M[12 + r] * 1.0f);
            localPositions.push_back(x); localPositions.push_back(y); localPositions.push_back(z);
        }
        int triCount = m.indexCount / 3;
        bool perTri = m.triangleMaterials && m.triangleMaterialCount == triCount;
        int t = 0, triLocal = 0;
        while (t + 2 < m.indexCount) {
            uint32_t i0 = baseVertex + m.indices[t], i1 = baseVertex + m.indices[t + 1], i2 = baseVertex + m.indices[t + 2];
            F3 p0{positions[i0 * 3], positions[i0 * 3 + 1], positions[i0 * 3 + 2]};
            F3 p1{positions[i1 * 3], positions[i1 * 3 + 1], positions[i1 * 3 + 2]};
            F3 p2{positions[i2 * 3], positions[i2 * 3 + 1], positions[i2 * 3 + 2]};
            if (lengthSq(cross(p1 - p0, p2 - p0)) <= areaEps) { t += 3; ++triLocal; continue; }
            indices.push_back(i0); indices.push_back(i1); indices.push_back(i2);
            F3 mn = vmin(p0, vmin(p1, p2)), mx = vmax(p0, vmax(p1, p2));
            const float bb[6] = {mn.x, mn.y, mn.z, mx.x, mx.y, mx.z};
            aabbs.insert(aabbs.end(), bb, bb + 6);
            materials.push_back(perTri ? m.triangleMaterials[triLocal] : m.material);
            layers.push_back(m.collisionLayer);
            t += 3; ++triLocal;
        }
        const int indexEnd = (int)indices.size(), triEnd = (int)layers.size();
        if (indexEnd > indexStart && triEnd > triStart) // a slice exists only for entities that kept a triangle (:404-410)
            slices[e] = HostMeshSlice{(int)baseVertex, (int)(positions.size() / 3), indexStart, indexEnd, triStart, triEnd, true};
    }
    int T = (int)layers.size();
    triOrder.resize(T);
    for (int i = 0; i < T; ++i) triOrder[i] = i;
    triLeaf.assign(T, -1);
    rank.assign(T, -1);
    if (T == 0) return;
    nodes.reserve((size_t)T);
    Builder b{*this};
    root = b.build(0, T, -1, 0);
    // visit rank: CollisionQuery's DFS pushes left then right and pops the last (CollisionQuery.swift:1106-1108)
    std::vector<int> stack{root};
    int counter = 0;
    while (!stack.empty()) {
        int ni = stack.back(); stack.pop_back();
        const HostBVHNode& n = nodes[ni];
        if (n.left < 0) {
            for (int i = n.start; i < n.start + n.count; ++i) rank[triOrder[i]] = counter++;
        } else {
            stack.push_back(n.left);
            stack.push_back(n.right);
        }
    }
    buildWide();
}

// Treelet cut of the binary BVH (see DevCollision): starting from a binary node, repeatedly open the frontier
// entry with the most triangles until kWideWidth entries exist or every entry holds <= kWideWidth triangles.
// Subtree triangle ranges are contiguous in triOrder because BVH.build partitions in place
// (left = [start, mid), right = [mid, end), CollisionQuery.swift:655-663).
// TriangleMeshSet.updateTransforms (CollisionQuery.swift:419-462)
int HostCollision::updateTransforms(const int32_t* entities, const float* modelMatrices, int n) {
    if (n <= 0 || layers.empty()) return 0;
    std::vector<int> updated;
    for (int k = 0; k < n; ++k) {
        const int e = entities[k];
        if (e < 0 || e >= (int)slices.size() || !slices[e].valid) continue;
        const HostMeshSlice& sl = slices[e];
        const float* M = modelMatrices + (size_t)k * 16;
        for (int i = sl.vertexBegin; i < sl.vertexEnd; ++i) {
            const float x = localPositions[(size_t)i * 3], y = localPositions[(size_t)i * 3 + 1], z = localPositions[(size_t)i * 3 + 2];
            for (int r = 0; r < 3; ++r) positions[(size_t)i * 3 + r] = ((M[r] * x + M[4 + r] * y) + M[8 + r] * z) + M[12 + r] * 1.0f;
        }
        int tri = sl.triBegin;
        for (int i = sl.indexBegin; i + 2 < sl.indexEnd; i += 3, ++tri) {
            const float *p0 = &positions[(size_t)indices[i] * 3], *p1 = &positions[(size_t)indices[i + 1] * 3], *p2 = &positions[(size_t)indices[i + 2] * 3];
            float* bb = &aabbs[(size_t)tri * 6];
            for (int a = 0; a < 3; ++a) {
                bb[a] = fminf(p0[a], fminf(p1[a], p2[a]));
                bb[3 + a] = fmaxf(p0[a], fmaxf(p1[a], p2[a]));
            }
            updated.push_back(tri);
        }
    }
    if (!updated.empty()) { refit(updated); reboundWide(); }
    return (int)updated.size();
}

// BVH.refit (CollisionQuery.swift:528-575): dirty leaves from their triangles, then every ancestor from its children,
// deepest first.
void HostCollision::refit(const std::vector<int>& updatedTriangles) {
    if (nodes.empty()) return;
    std::vector<char> leafDirty(nodes.size(), 0), parentDirty(nodes.size(), 0);
    std::vector<int> leaves, parents;
    for (int tri : updatedTriangles) {
        int leaf = triLeaf[tri];
        if (leaf >= 0 && !leafDirty[leaf]) { leafDirty[leaf] = 1; leaves.push_back(leaf); }
    }
    Builder b{*this};
    for (int leaf : leaves) b.rangeBounds(nodes[leaf].start, nodes[leaf].count, nodes[leaf].mn, nodes[leaf].mx);
    for (int leaf : leaves)
        for (int p = nodes[leaf].parent; p >= 0; p = nodes[p].parent)
            if (!parentDirty[p]) { parentDirty[p] = 1; parents.push_back(p); }
    std::vector<int> depth(parents.size()), order(parents.size());
    for (size_t i = 0; i < parents.size(); ++i) {
        int d = 0;
        for (int nidx = parents[i]; nidx >= 0; nidx = nodes[nidx].parent) ++d;
        depth[i] = d; order[i] = (int)i;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return depth[a] > depth[c]; });
    for (int i : order) {
        HostBVHNode& p = nodes[parents[i]];
        const HostBVHNode &l = nodes[p.left], &r = nodes[p.right];
        for (int k = 0; k < 3; ++k) { p.mn[k] = fminf(l.mn[k], r.mn[k]); p.mx[k] = fmaxf(l.mx[k], r.mx[k]); }
    }
}

void HostCollision::reboundWide() {
    for (size_t i = 0; i < wide.size(); ++i) {
        int b = wideBinary[i];
        if (b < 0) continue;
        const HostBVHNode& n = nodes[b];
        wide[i].mnx = n.mn[0]; wide[i].mny = n.mn[1]; wide[i].mnz = n.mn[2];
        wide[i].mxx = n.mx[0]; wide[i].mxy = n.mx[1]; wide[i].mxz = n.mx[2];
    }
}

void HostCollision::buildWide() {
    wide.clear(); wideBinary.clear();
    wideLevels = 0;
    if (root < 0) return;
    const int N = (int)nodes.size();
    std::vector<int> cnt(N), lo(N);
    for (int i = N - 1; i >= 0; --i) { // children follow their parent (pre-order), so a reverse sweep is post-order
        const HostBVHNode& n = nodes[i];
        if (n.left < 0) { cnt[i] = n.count; lo[i] = n.start; }
        else { cnt[i] = cnt[n.left] + cnt[n.right]; lo[i] = std::min(lo[n.left], lo[n.right]); }
    }
    struct Job { int binary, wideIndex, level; };
    std::vector<Job> jobs{{root, 0, 1}};
    wideLevels = 1;
    wide.resize(kWideWidth);
    wideBinary.assign(kWideWidth, -1);
    for (size_t q = 0; q < jobs.size(); ++q) {
        std::vector<int> frontier{jobs[q].binary};
        while ((int)frontier.size() < kWideWidth) {
            int best = -1;
            for (int k = 0; k < (int)frontier.size(); ++k) {
                int b = frontier[k];
                if (nodes[b].left >= 0 && cnt[b] > kWideWidth && (best < 0 || cnt[b] > cnt[frontier[best]])) best = k;
            }
            if (best < 0) break;
            int b = frontier[best];
            frontier[best] = nodes[b].left;
            frontier.push_back(nodes[b].right);
        }
        for (int k = 0; k < kWideWidth; ++k) {
            DevNode d{kFloatMax, kFloatMax, kFloatMax, -kFloatMax, -kFloatMax, -kFloatMax, 0, 0};
            if (k < (int)frontier.size()) {
                int b = frontier[k];
                const HostBVHNode& n = nodes[b];
                d = DevNode{n.mn[0], n.mn[1], n.mn[2], n.mx[0], n.mx[1], n.mx[2], 0, 0};
                if (cnt[b] <= kWideWidth) { d.a = ~lo[b]; d.b = cnt[b]; }
                else {
                    int w = (int)(wide.size() / kWideWidth);
                    wide.resize(wide.size() + kWideWidth);
                    wideBinary.resize(wide.size(), -1);
                    jobs.push_back({b, w, jobs[q].level + 1});
                    wideLevels = std::max(wideLevels, jobs[q].level + 1);
                    d.a = w; d.b = 0;
                }
            }
            wide[(size_t)jobs[q].wideIndex * kWideWidth + k] = d;
            wideBinary[(size_t)jobs[q].wideIndex * kWideWidth + k] = k < (int)frontier.size() ? frontier[k] : -1;
        }
    }
}

// ---------------------------------------------------------------------------
// HostBlas::build — topology of the skinned-geometry acceleration structure (RTAccelerationBuilder.swift:75-112 builds
// one per skinned item; Metal's result is opaque, so the layout is this library's own): triangles are split at the
// centroid median of the widest centroid axis, left halves rounded up to whole clusters of 64, until a range holds
// <= 64 triangles (a cluster); the binary tree is then cut into 64-entry wide nodes, top down, every wide node expanding
// its largest entries until each holds at most 64^(levels below - 1) clusters. Deterministic: ties on the centroid go
// to the smaller triangle index, triangles inside a cluster are ordered by index.
// ---------------------------------------------------------------------------
bool HostBlas::build(const float* pos, int V, const uint32_t* idx, int indexCount, std::string& err) {
    *this = HostBlas{};
    if (!pos || !idx || V <= 0 || indexCount < 3 || indexCount % 3 != 0) { err = "blas: needs positions and a non-empty triangle list"; return false; }
    const int T = indexCount / 3;
    for (int i = 0; i < indexCount; ++i)
        if (idx[i] >= (uint32_t)V) { err = "blas: vertex index out of range"; return false; }
    std::vector<float> cen((size_t)T * 3);
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k)
            cen[(size_t)t * 3 + k] = ((pos[(size_t)idx[t * 3] * 3 + k] + pos[(size_t)idx[t * 3 + 1] * 3 + k]) + pos[(size_t)idx[t * 3 + 2] * 3 + k]) * (1.0f / 3.0f);
    std::vector<int> order(T);
    for (int t = 0; t < T; ++t) order[t] = t;
    struct BNode { int begin, end, left, right; };
    std::vector<BNode> nodes;
    nodes.reserve((size_t)T / 16 + 8);
    // explicit stack: (node index) to split
    nodes.push_back(BNode{0, T, -1, -1});
    std::vector<int> todo{0};
    while (!todo.empty()) {
        const int ni = todo.back();
        todo.pop_back();
        const int begin = nodes[ni].begin, end = nodes[ni].end, n = end - begin;
        if (n <= SGE_BLAS_CLUSTER) { std::sort(order.begin() + begin, order.begin() + end); continue; }
        float mn[3] = {kFloatMax, kFloatMax, kFloatMax}, mx[3] = {-kFloatMax, -kFloatMax, -kFloatMax};
        for (int s = begin; s < end; ++s)
            for (int k = 0; k < 3; ++k) {
                const float c = cen[(size_t)order[s] * 3 + k];
                if (c < mn[k]) mn[k] = c;
                if (c > mx[k]) mx[k] = c;
            }
        int axis = 0;
        if (mx[1] - mn[1] > mx[axis] - mn[axis]) axis = 1;
        if (mx[2] - mn[2] > mx[axis] - mn[axis]) axis = 2;
        const int clusters = (n + SGE_BLAS_CLUSTER - 1) / SGE_BLAS_CLUSTER;
        const int leftCount = (clusters + 1) / 2 * SGE_BLAS_CLUSTER;
        std::nth_element(order.begin() + begin, order.begin() + begin + leftCount, order.begin() + end, [&](int a, int b) {
            const float ca = cen[(size_t)a * 3 + axis], cb = cen[(size_t)b * 3 + axis];
            return ca < cb || (ca == cb && a < b);
        });
        const int l = (int)nodes.size();
        nodes.push_back(BNode{begin, begin + leftCount, -1, -1});
        nodes.push_back(BNode{begin + leftCount, end, -1, -1});
        nodes[ni].left = l;
        nodes[ni].right = l + 1;
        todo.push_back(l);
        todo.push_back(l + 1);
    }
    auto clustersOf = [&](int ni) { return (nodes[ni].end - nodes[ni].begin + SGE_BLAS_CLUSTER - 1) / SGE_BLAS_CLUSTER; };

    // cut into wide nodes, breadth first (children are numbered after all nodes of their parent's level)
    struct Pending { int binary, parentEntry; };
    std::vector<Pending> level{Pending{0, -1}}, next;
    wideFirst.push_back(0);
    wideLevelStart.push_back(0);
    int wideTotal = 1; // wide nodes numbered so far (including the ones waiting in `level` / `next`)
    while (!level.empty()) {
        next.clear();
        for (const Pending& pn : level) {
            long long target = 1;
            while (target * SGE_BLAS_WIDTH < clustersOf(pn.binary)) target *= SGE_BLAS_WIDTH;
            std::vector<int> ents{pn.binary};
            while (true) {
                int pick = -1;
                for (int k = 0; k < (int)ents.size(); ++k)
                    if (nodes[ents[k]].left >= 0 && clustersOf(ents[k]) > target && (pick < 0 || clustersOf(ents[k]) > clustersOf(ents[pick]))) pick = k;
                if (pick < 0) break;
                const int b = ents[pick];
                ents[pick] = nodes[b].left;
                ents.insert(ents.begin() + pick + 1, nodes[b].right);
            }
            if ((int)ents.size() > SGE_BLAS_WIDTH) { err = "blas: internal error, wide node overflow"; return false; }
            wideParentEntry.push_back(pn.parentEntry);
            for (int b : ents) {
                const int e = entryCount();
                if (nodes[b].left < 0) {
                    entryLink.push_back(~nodes[b].begin);
                    entryLink.push_back(nodes[b].end - nodes[b].begin);
                    ++clusterCount;
                } else {
                    entryLink.push_back(wideTotal++);
                    entryLink.push_back(0);
                    next.push_back(Pending{b, e});
                }
            }
            wideFirst.push_back(entryCount());
        }
        wideLevelStart.push_back((int)wideParentEntry.size());
        ++levels;
        level.swap(next);
    }
    if (entryCount() >= (1 << 24)) { err = "blas: more than 2^24 entries"; return false; }
    if (blasRefitLdsBytes(entryCount(), V / 512 + 2) + blasTopoBytes(wideCount(), levels) > kBlasMaxLdsBytes) { err = "blas: mesh too large (more than ~250k triangles per character)"; return false; }
    tileCap = kBlasTileVerts;
    for (int cap : kBlasTileSteps)
        if (cap >= 2048 && 3 * (blasRefitLdsBytes(entryCount(), V / cap + 2, cap) + blasTopoBytes(wideCount(), levels) + 256) <= (size_t)160 * 1024) { tileCap = cap; break; }
    if (const char* e = getenv("SGE_BLAS_TILE_CAP")) { const int v = atoi(e); if (v == 2048 || v == 3072 || v == 4096) tileCap = v; } // experiments
    tileCount = (V + tileCap - 1) / tileCap;
    tileVerts = ((V + tileCount - 1) / tileCount + 63) / 64 * 64;
    tileCount = (V + tileVerts - 1) / tileVerts;

    triCount = T;
    vertexCount = V;
    slotTriangle.resize(T);
    slotIndices.resize((size_t)T * 3);
    for (int s = 0; s < T; ++s) {
        slotTriangle[s] = (uint32_t)order[s];
        for (int k = 0; k < 3; ++k) slotIndices[(size_t)s * 3 + k] = idx[(size_t)order[s] * 3 + k];
    }
    // vertex -> clusters (sorted, unique)
    std::vector<std::pair<int, int>> inc; // (vertex, entry)
    inc.reserve((size_t)T * 3);
    for (int e = 0; e < entryCount(); ++e) {
        if (entryLink[e * 2] >= 0) continue;
        const int first = ~entryLink[e * 2], cnt = entryLink[e * 2 + 1];
        for (int s = first; s < first + cnt; ++s)
            for (int k = 0; k < 3; ++k) inc.emplace_back((int)slotIndices[(size_t)s * 3 + k], e);
    }
    std::sort(inc.begin(), inc.end());
    inc.erase(std::unique(inc.begin(), inc.end()), inc.end());
    vertexEntryStart.assign((size_t)V + 1, 0);
    vertexEntries.resize(inc.size());
    for (size_t i = 0; i < inc.size(); ++i) { vertexEntryStart[(size_t)inc[i].first + 1]++; vertexEntries[i] = inc[i].second; }
    for (int v = 0; v < V; ++v) vertexEntryStart[(size_t)v + 1] += vertexEntryStart[v];

    // refit schedule (see HostBlas): per tile, the (cluster, vertices of the tile in it) pairs cut into chunks of <= 16,
    // longest first, 64 chunks per round
    std::vector<std::vector<uint16_t>> bucket(entryCount());
    std::vector<int> touched;
    struct Chunk { int cluster; std::vector<uint16_t> ids; };
    chunkCount = 0;
    tileRoundStart.assign(1, 0);
    for (int tile = 0; tile < tileCount; ++tile) {
        const int base = tile * tileVerts, end = std::min(V, base + tileVerts);
        touched.clear();
        for (int v = base; v < end; ++v)
            for (int j = vertexEntryStart[v]; j < vertexEntryStart[(size_t)v + 1]; ++j) {
                const int e = vertexEntries[j];
                if (bucket[e].empty()) touched.push_back(e);
                bucket[e].push_back((uint16_t)(v - base));
            }
        std::sort(touched.begin(), touched.end());
        std::vector<Chunk> chunks;
        for (int e : touched) {
            const std::vector<uint16_t>& ids = bucket[e];
            for (size_t k = 0; k < ids.size(); k += 16)
                chunks.push_back(Chunk{e, std::vector<uint16_t>(ids.begin() + k, ids.begin() + std::min(ids.size(), k + 16))});
            bucket[e].clear();
        }
        std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk& a, const Chunk& b) { return a.ids.size() > b.ids.size(); });
        chunkCount += (int)chunks.size();
        for (size_t r = 0; r < chunks.size(); r += 64) {
            const int len = (int)chunks[r].ids.size();
            roundLen.push_back(len);
            const size_t idBase = roundIds.size();
            roundIds.resize(idBase + 8 * 64, 0u);
            // The order of a chunk's vertices is free (min / max), so it is chosen against LDS bank conflicts: at step i the
            // 32 lanes of a half-wave read X[id] (bank = id mod 32, same for Y and Z) together; every lane takes, among the
            // vertices it has not read yet (all of them again once it ran out: padding), the one on the least loaded bank.
            std::vector<uint16_t> order[64];
            for (int lane = 0; lane < 64; ++lane) roundCluster.push_back(chunks[r + lane < chunks.size() ? r + lane : r].cluster | (len << 24));
            for (int half = 0; half < 2; ++half) {
                std::vector<uint16_t> left[32];
                for (int l = 0; l < 32; ++l) left[l] = chunks[r + half * 32 + l < chunks.size() ? r + half * 32 + l : r].ids;
                for (int i = 0; i < len; ++i) {
                    // lanes that still have unread vertices are matched to distinct banks as far as possible (augmenting
                    // paths over lane -> candidate banks); the rest, and the padding lanes, take their least loaded bank
                    int owner[32], pick[32];
                    for (int b = 0; b < 32; ++b) owner[b] = -1;
                    for (int l = 0; l < 32; ++l) pick[l] = -1;
                    bool seen[32];
                    std::function<bool(int)> augment = [&](int l) -> bool {
                        for (size_t k = 0; k < left[l].size(); ++k) {
                            const int b = left[l][k] & 31;
                            if (seen[b]) continue;
                            seen[b] = true;
                            if (owner[b] < 0 || augment(owner[b])) { owner[b] = l; pick[l] = (int)k; return true; }
                        }
                        return false;
                    };
                    for (int l = 0; l < 32; ++l)
                        if (!left[l].empty()) { for (bool& f : seen) f = false; augment(l); }
                    // an augmenting path re-picks by bank: resolve every matched lane's vertex from the bank it owns
                    for (int b = 0; b < 32; ++b)
                        if (owner[b] >= 0) {
                            const int l = owner[b];
                            for (size_t k = 0; k < left[l].size(); ++k)
                                if ((left[l][k] & 31) == b) { pick[l] = (int)k; break; }
                        }
                    int load[32] = {0};
                    bool matched[32];
                    for (int l = 0; l < 32; ++l) matched[l] = false;
                    for (int b = 0; b < 32; ++b) if (owner[b] >= 0) { matched[owner[b]] = true; load[b] = 1; }
                    for (int l = 0; l < 32; ++l) {
                        const Chunk& ch = chunks[r + half * 32 + l < chunks.size() ? r + half * 32 + l : r];
                        std::vector<uint16_t>& pool = left[l];
                        if (matched[l]) {
                            order[half * 32 + l].push_back(pool[pick[l]]);
                            pool.erase(pool.begin() + pick[l]);
                            continue;
                        }
                        const bool pad = pool.empty();
                        const std::vector<uint16_t>& from = pad ? ch.ids : pool;
                        size_t best = 0;
                        for (size_t k = 1; k < from.size(); ++k)
                            if (load[from[k] & 31] < load[from[best] & 31]) best = k;
                        const uint16_t id = from[best];
                        load[id & 31]++;
                        order[half * 32 + l].push_back(id);
                        if (!pad) pool.erase(pool.begin() + (long)best);
                    }
                }
            }
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 16; ++i) {
                    const uint32_t off = 4u * order[lane][i < len ? i : 0];
                    roundIds[idBase + (size_t)lane * 8 + (i >> 1)] |= off << (16 * (i & 1));
                }
        }
        tileRoundStart.push_back((int)roundLen.size());
    }
    return true;
}

} // namespace sge

using namespace sge;

extern "C" {

int sge_skeleton_build(int32_t boneCount, const int32_t* parent, const float* rawTranslations,
                       const float* preRotationDegrees, const float rootFixDegrees[3], float unitScale, int zeroRoot,
                       float* restTranslation, float* bindLocal, float* invBindModel, float* rootRotationFix) {
    if (boneCount <= 0 || boneCount > SGE_MAX_BONES || !parent || !rawTranslations || !preRotationDegrees ||
        !rootFixDegrees || !restTranslation || !bindLocal || !invBindModel || !rootRotationFix) {
        set_error("sge_skeleton_build: bad argument");
        return SGE_ERR_INVALID;
    }
    Aff rootFix = rotationXYZDegrees(F3{rootFixDegrees[0], rootFixDegrees[1], rootFixDegrees[2]});
    std::vector<Aff> local(boneCount), model(boneCount);
    for (int i = 0; i < boneCount; ++i) {
        if (parent[i] >= i) { set_error("sge_skeleton_build: parent index must precede child"); return SGE_ERR_INVALID; }
        F3 raw = (zeroRoot && i == 0) ? F3{0, 0, 0} : F3{rawTranslations[i * 3], rawTranslations[i * 3 + 1], rawTranslations[i * 3 + 2]};
        F3 t = raw * unitScale;
        restTranslation[i * 3] = t.x; restTranslation[i * 3 + 1] = t.y; restTranslation[i * 3 + 2] = t.z;
        Aff rot = affMul(rotationXYZDegrees(F3{preRotationDegrees[i * 3], preRotationDegrees[i * 3 + 1], preRotationDegrees[i * 3 + 2]}),
                         rotationXYZDegrees(F3{0, 0, 0}));
        if (i == 0) rot = affMul(rootFix, rot);
        Aff tr = affIdentity();
        tr.c3 = t;
        local[i] = affMul(tr, rot); // simd_mul(trans, rot)
        model[i] = parent[i] < 0 ? local[i] : affMul(model[parent[i]], local[i]);
    }
    for (int i = 0; i < boneCount; ++i) {
        M4 l = m4FromAff(local[i]);
        M4 inv = m4Inverse(m4FromAff(model[i]));
        std::memcpy(bindLocal + i * 16, l.m, 64);
        std::memcpy(invBindModel + i * 16, inv.m, 64);
    }
    M4 rf = m4FromAff(rootFix);
    std::memcpy(rootRotationFix, rf.m, 64);
    return SGE_OK;
}

int sge_mesh_tangents_compute(int32_t vCount, const float* positions, const float* normals, const float* uvs,
                              const uint16_t* indices16, const uint32_t* indices32, int32_t indexCount, float* tangents) {
    if (vCount < 0 || !positions || !normals || !uvs || !tangents || (indices16 && indices32)) {
        set_error("sge_mesh_tangents_compute: bad argument");
        return SGE_ERR_INVALID;
    }
    std::vector<F3> tan1(vCount, F3{0, 0, 0}), tan2(vCount, F3{0, 0, 0});
    auto P = [&](int i) { return F3{positions[i * 3], positions[i * 3 + 1], positions[i * 3 + 2]}; };
    for (int idx = 0; idx + 2 < indexCount; idx += 3) {
        int i0, i1, i2;
        if (indices16) { i0 = indices16[idx]; i1 = indices16[idx + 1]; i2 = indices16[idx + 2]; }
        else if (indices32) { i0 = (int)indices32[idx]; i1 = (int)indices32[idx + 1]; i2 = (int)indices32[idx + 2]; }
        else break;
        F3 dp1 = P(i1) - P(i0), dp2 = P(i2) - P(i0);
        float d1x = uvs[i1 * 2] - uvs[i0 * 2], d1y = uvs[i1 * 2 + 1] - uvs[i0 * 2 + 1];
        float d2x = uvs[i2 * 2] - uvs[i0 * 2], d2y = uvs[i2 * 2 + 1] - uvs[i0 * 2 + 1];
        float denom = d1x * d2y - d1y * d2x;
        if (fabsf(denom) < 1e-6f) continue;
        float r = 1.0f / denom;
        F3 t = (dp1 * d2y - dp2 * d1y) * r;
        F3 b = (dp2 * d1x - dp1 * d2x) * r;
        tan1[i0] = tan1[i0] + t; tan1[i1] = tan1[i1] + t; tan1[i2] = tan1[i2] + t;
        tan2[i0] = tan2[i0] + b; tan2[i1] = tan2[i1] + b; tan2[i2] = tan2[i2] + b;
    }
    for (int i = 0; i < vCount; ++i) {
        F3 n = normalize(F3{normals[i * 3], normals[i * 3 + 1], normals[i * 3 + 2]});
        F3 t = tan1[i];
        float* o = tangents + i * 4;
        if (lengthSq(t) < 1e-8f) { o[0] = 1; o[1] = 0; o[2] = 0; o[3] = 1; continue; }
        t = normalize(t - n * dot(n, t));
        float w = dot(cross(n, t), tan2[i]) < 0.0f ? -1.0f : 1.0f;
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = w;
    }
    return SGE_OK;
}

int sge_blas_topology(const float* positions, int32_t vertex_count, const uint32_t* indices, int32_t index_count,
                      sge_blas_info* info, int32_t* entry_link, int32_t* wide_first, int32_t* wide_parent_entry,
                      uint32_t* slot_triangle, int32_t* vertex_entry_start, int32_t* vertex_entries) {
    HostBlas hb;
    std::string err;
    if (!hb.build(positions, vertex_count, indices, index_count, err)) { set_error(err); return SGE_ERR_INVALID; }
    if (info) *info = sge_blas_info{hb.triCount, hb.clusterCount, hb.entryCount(), hb.wideCount(), hb.levels, (int32_t)hb.vertexEntries.size()};
    if (entry_link) std::memcpy(entry_link, hb.entryLink.data(), hb.entryLink.size() * 4);
    if (wide_first) std::memcpy(wide_first, hb.wideFirst.data(), hb.wideFirst.size() * 4);
    if (wide_parent_entry) std::memcpy(wide_parent_entry, hb.wideParentEntry.data(), hb.wideParentEntry.size() * 4);
    if (slot_triangle) std::memcpy(slot_triangle, hb.slotTriangle.data(), hb.slotTriangle.size() * 4);
    if (vertex_entry_start) std::memcpy(vertex_entry_start, hb.vertexEntryStart.data(), hb.vertexEntryStart.size() * 4);
    if (vertex_entries) std::memcpy(vertex_entries, hb.vertexEntries.data(), hb.vertexEntries.size() * 4);
    return SGE_OK;
}

} // extern "C"
