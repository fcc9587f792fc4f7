// ORACLE — TEST INFRASTRUCTURE ONLY (see sge_oracle_math.h).
//
// CPU float32 restatement of Game/CollisionQuery.swift, static and dynamic triangle sets:
//   TriangleMeshSet.rebuild / updateTransforms  :331-417, :419-462
//   BVH.build + helpers / BVH.refit             :577-706, :528-575
//   raycast / raycastBVH / rayTriangle / rayAABB :768-785, :916-978, :1575-1631
//   capsuleCastCombined / capsuleCastBVH / chooseNearest   :980-1117, :909-914
//   capsuleOverlap / capsuleOverlapBVH          :830-850, :1119-1199
//   capsuleOverlapAll / capsuleOverlapBVHAll    :852-882, :1201-1283
//   sweepCapsuleTriangle / refineTOI            :1285-1394
//   segmentTriangleDistance and helpers         :1396-1573
// Parity unpinned: the reference holds no test or golden vector for any of it
// (GameTests/GameTests.swift:12-16 is empty); tests pin this file with analytic
// known answers and brute-force cross-checks instead.
#include <algorithm>
#include <limits>
#include "sge_oracle.h"

namespace sgeo {

static const int kLeafTriangleLimit = 4; // CollisionQuery.swift:473

static thread_local QueryStats g_tls_stats;
QueryStats& thread_stats() { return g_tls_stats; }
QueryStats take_thread_stats() { QueryStats s = g_tls_stats; g_tls_stats = QueryStats(); return s; }

static inline V3 centroid(const AABB& b) { return (b.min + b.max) * 0.5f; }               // :700
static inline AABB merge(const AABB& a, const AABB& b) { return AABB{vmin(a.min, b.min), vmax(a.max, b.max)}; } // :704
static inline float axisOf(V3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

struct BVHBuilder {
    BVH& bvh;
    const std::vector<AABB>& tri;
    AABB boundsForRange(int start, int count) const { // :672
        AABB first = tri[bvh.triOrder[start]];
        V3 bmin = first.min, bmax = first.max;
        for (int i = 1; i < count; ++i) {
            const AABB& b = tri[bvh.triOrder[start + i]];
            bmin = vmin(bmin, b.min);
            bmax = vmax(bmax, b.max);
        }
        return AABB{bmin, bmax};
    }
    AABB centroidBoundsForRange(int start, int count) const { // :686
        V3 first = centroid(tri[bvh.triOrder[start]]);
        V3 bmin = first, bmax = first;
        for (int i = 1; i < count; ++i) {
            V3 c = centroid(tri[bvh.triOrder[start + i]]);
            bmin = vmin(bmin, c);
            bmax = vmax(bmax, c);
        }
        return AABB{bmin, bmax};
    }
    int build(int start, int count, int parent) { // :577-670
        int nodeIndex = (int)bvh.nodes.size();
        AABB bounds = boundsForRange(start, count);
        bvh.nodes.push_back(BVHNode{bounds, -1, -1, start, count, parent});
        if (count <= kLeafTriangleLimit) {
            for (int i = 0; i < count; ++i) bvh.triLeaf[bvh.triOrder[start + i]] = nodeIndex;
            return nodeIndex;
        }
        AABB cb = centroidBoundsForRange(start, count);
        V3 extent = cb.max - cb.min;
        int axis;
        if (extent.x >= extent.y && extent.x >= extent.z) axis = 0;
        else if (extent.y >= extent.z) axis = 1;
        else axis = 2;
        float pivot = (axisOf(cb.min, axis) + axisOf(cb.max, axis)) * 0.5f;
        int i = start, j = start + count - 1;
        while (i <= j) {
            int t = bvh.triOrder[i];
            float value = axisOf(centroid(tri[t]), axis);
            if (value < pivot) {
                i += 1;
            } else {
                std::swap(bvh.triOrder[i], bvh.triOrder[j]);
                j -= 1;
            }
        }
        int end = start + count;
        if (i == start || i == end) {
            // Swift's sort is a stable merge sort since 5.0.
            std::stable_sort(bvh.triOrder.begin() + start, bvh.triOrder.begin() + end, [&](int a, int b) {
                return axisOf(centroid(tri[a]), axis) < axisOf(centroid(tri[b]), axis);
            });
            i = start + count / 2;
        }
        int mid = i;
        int left = build(start, mid - start, nodeIndex);
        int right = build(mid, start + count - mid, nodeIndex);
        bvh.nodes[nodeIndex].left = left;
        bvh.nodes[nodeIndex].right = right;
        bvh.nodes[nodeIndex].start = 0;
        bvh.nodes[nodeIndex].count = 0;
        bvh.nodes[nodeIndex].bounds = merge(bvh.nodes[left].bounds, bvh.nodes[right].bounds);
        return nodeIndex;
    }
};

void BVH::build(const std::vector<AABB>& triangleAABBs) { // :502-513
    nodes.clear();
    int n = (int)triangleAABBs.size();
    triOrder.resize(n);
    for (int i = 0; i < n; ++i) triOrder[i] = i;
    triLeaf.assign(n, -1);
    root = -1;
    if (n > 0) {
        BVHBuilder b{*this, triangleAABBs};
        root = b.build(0, n, -1);
    }
}

void TriangleMeshSet::rebuild(const sge_static_mesh_entity* ents, int count) { // :331-417
    positions.clear(); indices.clear(); triangleAABBs.clear(); triangleMaterials.clear(); triangleLayers.clear();
    slices.assign(count > 0 ? count : 0, MeshSlice{});
    localPositions.clear();
    const float areaEps = 1e-10f;
    for (int e = 0; e < count; ++e) {
        const sge_static_mesh_entity& m = ents[e];
        M4 model;
        for (int k = 0; k < 16; ++k) (&model.c[0].x)[k] = m.modelMatrix[k];
        uint32_t baseVertex = (uint32_t)positions.size();
        const int indexStart = (int)indices.size(), triStart = (int)triangleAABBs.size();
        for (int v = 0; v < m.vertexCount; ++v) {
            V4 p = V4{m.positions[v * 3], m.positions[v * 3 + 1], m.positions[v * 3 + 2], 1};
            V4 wp = mul(model, p);
            positions.push_back(V3{wp.x, wp.y, wp.z});
            localPositions.push_back(V3{p.x, p.y, p.z});
        }
        int triCount = m.indexCount / 3;
        bool perTri = m.triangleMaterials && m.triangleMaterialCount == triCount;
        int t = 0, triLocal = 0;
        while (t + 2 < m.indexCount) {
            uint32_t i0 = baseVertex + m.indices[t], i1 = baseVertex + m.indices[t + 1], i2 = baseVertex + m.indices[t + 2];
            V3 p0 = positions[i0], p1 = positions[i1], p2 = positions[i2];
            V3 e1 = p1 - p0, e2 = p2 - p0;
            if (length_squared(cross(e1, e2)) <= areaEps) { t += 3; triLocal += 1; continue; }
            indices.push_back(i0); indices.push_back(i1); indices.push_back(i2);
            triangleAABBs.push_back(AABB{vmin(p0, vmin(p1, p2)), vmax(p0, vmax(p1, p2))});
            triangleMaterials.push_back(perTri ? m.triangleMaterials[triLocal] : m.material);
            triangleLayers.push_back(m.collisionLayer);
            t += 3; triLocal += 1;
        }
        const int indexEnd = (int)indices.size(), triEnd = (int)triangleAABBs.size();
        if (indexEnd > indexStart && triEnd > triStart) // :404-410
            slices[e] = MeshSlice{(int)baseVertex, (int)positions.size(), indexStart, indexEnd, triStart, triEnd, true};
    }
    hasBVH = !triangleAABBs.empty();
    if (hasBVH) bvh.build(triangleAABBs);
    else { bvh.nodes.clear(); bvh.triOrder.clear(); bvh.triLeaf.clear(); bvh.root = -1; }
}

// :419-462
std::vector<int> TriangleMeshSet::updateTransforms(const int32_t* entities, const float* modelMatrices, int n) {
    std::vector<int> updatedTriangles;
    if (n <= 0 || triangleAABBs.empty()) return updatedTriangles;
    for (int k = 0; k < n; ++k) {
        const int e = entities[k];
        if (e < 0 || e >= (int)slices.size() || !slices[e].valid) continue; // `guard let slice = slices[e]`
        const MeshSlice& slice = slices[e];
        M4 model;
        for (int j = 0; j < 16; ++j) (&model.c[0].x)[j] = modelMatrices[(size_t)k * 16 + j];
        for (int i = slice.vertexBegin; i < slice.vertexEnd; ++i) {
            V3 pl = localPositions[i];
            V4 wp = mul(model, V4{pl.x, pl.y, pl.z, 1});
            positions[i] = V3{wp.x, wp.y, wp.z};
        }
        int triIndex = slice.triBegin;
        int i = slice.indexBegin;
        while (i + 2 < slice.indexEnd) {
            V3 p0 = positions[indices[i]], p1 = positions[indices[i + 1]], p2 = positions[indices[i + 2]];
            triangleAABBs[triIndex] = AABB{vmin(p0, vmin(p1, p2)), vmax(p0, vmax(p1, p2))};
            updatedTriangles.push_back(triIndex);
            i += 3;
            triIndex += 1;
        }
    }
    if (!updatedTriangles.empty() && hasBVH) bvh.refit(updatedTriangles, triangleAABBs);
    return updatedTriangles;
}

// :528-575 — leaves holding an updated triangle are re-bounded from their triangles, then every ancestor,
// deepest first, from its two children
void BVH::refit(const std::vector<int>& updatedTriangles, const std::vector<AABB>& triangleAABBs) {
    if (nodes.empty()) return;
    std::vector<char> leafDirty(nodes.size(), 0), parentDirty(nodes.size(), 0);
    std::vector<int> updatedLeaves, dirtyParents;
    for (int tri : updatedTriangles) {
        int leaf = triLeaf[tri];
        if (leaf >= 0 && !leafDirty[leaf]) { leafDirty[leaf] = 1; updatedLeaves.push_back(leaf); }
    }
    for (int leaf : updatedLeaves) {
        BVHNode& node = nodes[leaf];
        AABB b = triangleAABBs[triOrder[node.start]]; // boundsForRange :672-683
        for (int i = 1; i < node.count; ++i) b = merge(b, triangleAABBs[triOrder[node.start + i]]);
        node.bounds = b;
    }
    for (int leaf : updatedLeaves) {
        int parent = nodes[leaf].parent;
        while (parent >= 0) {
            if (!parentDirty[parent]) { parentDirty[parent] = 1; dirtyParents.push_back(parent); }
            parent = nodes[parent].parent;
        }
    }
    if (dirtyParents.empty()) return;
    std::vector<int> depths(dirtyParents.size());
    for (size_t i = 0; i < dirtyParents.size(); ++i) {
        int depth = 0, node = dirtyParents[i];
        while (node >= 0) { depth += 1; node = nodes[node].parent; }
        depths[i] = depth;
    }
    std::vector<int> order(dirtyParents.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return depths[a] > depths[b]; });
    for (int i : order) {
        BVHNode& p = nodes[dirtyParents[i]];
        p.bounds = merge(nodes[p.left].bounds, nodes[p.right].bounds);
    }
}

// ---- primitive distance queries ----

// :1440-1462
static bool segmentTriangleIntersect(V3 a, V3 b, V3 v0, V3 v1, V3 v2, V3& out) {
    V3 dir = b - a;
    const float eps = 1e-6f;
    V3 e1 = v1 - v0, e2 = v2 - v0;
    V3 pvec = cross(dir, e2);
    float det = dot(e1, pvec);
    if (fabsf(det) < eps) return false;
    float invDet = 1.0f / det;
    V3 tvec = a - v0;
    float u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    V3 qvec = cross(tvec, e1);
    float v = dot(dir, qvec) * invDet;
    if (v < 0 || (u + v) > 1) return false;
    float t = dot(e2, qvec) * invDet;
    if (t < 0 || t > 1) return false;
    out = a + dir * t;
    return true;
}

// :1464-1517
static float closestPointOnTriangle(V3 p, V3 a, V3 b, V3 c, V3& point) {
    V3 ab = b - a, ac = c - a, ap = p - a;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) { point = a; return length_squared(p - a); }
    V3 bp = p - b;
    float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) { point = b; return length_squared(p - b); }
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) {
        float v = d1 / (d1 - d3);
        point = a + ab * v;
        return length_squared(p - point);
    }
    V3 cp = p - c;
    float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) { point = c; return length_squared(p - c); }
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) {
        float w = d2 / (d2 - d6);
        point = a + ac * w;
        return length_squared(p - point);
    }
    float va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
        float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        point = b + (c - b) * w;
        return length_squared(p - point);
    }
    float denom = 1.0f / (va + vb + vc);
    float v = vb * denom, w = vc * denom;
    point = (a + ab * v) + ac * w;
    return length_squared(p - point);
}

// :1519-1569
static float segmentSegmentDistanceSq(V3 p1, V3 q1, V3 p2, V3 q2, V3& c1o, V3& c2o) {
    V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    float s = 0, t = 0;
    const float eps = 1e-6f;
    if (a <= eps && e <= eps) { c1o = p1; c2o = p2; return length_squared(p1 - p2); }
    if (a <= eps) {
        t = clampf(f / e, 0, 1);
        V3 c2 = p2 + d2 * t;
        c1o = p1; c2o = c2;
        return length_squared(p1 - c2);
    }
    float c = dot(d1, r);
    if (e <= eps) {
        s = clampf(-c / a, 0, 1);
        V3 c1 = p1 + d1 * s;
        c1o = c1; c2o = p2;
        return length_squared(c1 - p2);
    }
    float b = dot(d1, d2);
    float denom = a * e - b * b;
    if (denom != 0) s = clampf((b * f - c * e) / denom, 0, 1);
    else s = 0;
    float tNom = b * s + f;
    if (tNom < 0) { t = 0; s = clampf(-c / a, 0, 1); }
    else if (tNom > e) { t = 1; s = clampf((b - c) / a, 0, 1); }
    else t = tNom / e;
    V3 c1 = p1 + d1 * s, c2 = p2 + d2 * t;
    c1o = c1; c2o = c2;
    return length_squared(c1 - c2);
}

// :1396-1438
static float segmentTriangleDistance(V3 center, float halfHeight, V3 v0, V3 v1, V3 v2, V3& segPoint, V3& triPoint) {
    V3 up = V3{0, 1, 0};
    V3 a = center + up * halfHeight;
    V3 b = center - up * halfHeight;
    V3 hit;
    if (segmentTriangleIntersect(a, b, v0, v1, v2, hit)) { segPoint = hit; triPoint = hit; return 0; }
    float bestDistSq = 3.40282347e+38f;
    V3 bestSeg = a, bestTri = v0;
    V3 p0, p1;
    float d0 = closestPointOnTriangle(a, v0, v1, v2, p0);
    if (d0 < bestDistSq) { bestDistSq = d0; bestSeg = a; bestTri = p0; }
    float dd1 = closestPointOnTriangle(b, v0, v1, v2, p1);
    if (dd1 < bestDistSq) { bestDistSq = dd1; bestSeg = b; bestTri = p1; }
    V3 e0s[3] = {v0, v1, v2}, e1s[3] = {v1, v2, v0};
    for (int k = 0; k < 3; ++k) {
        V3 s, t;
        float d = segmentSegmentDistanceSq(a, b, e0s[k], e1s[k], s, t);
        if (d < bestDistSq) { bestDistSq = d; bestSeg = s; bestTri = t; }
    }
    segPoint = bestSeg; triPoint = bestTri;
    return sqrtf(fmax_s(bestDistSq, 0.0f));
}

// :1361-1394
static float refineTOI(V3 from, V3 dir, float radius, float halfHeight, V3 v0, V3 v1, V3 v2,
                       float t0, float t1, float maxDistance) {
    float clampT0 = fmax_s(0.0f, fmin_s(t0, maxDistance));
    float clampT1 = fmax_s(0.0f, fmin_s(t1, maxDistance));
    float lo = fmin_s(clampT0, clampT1);
    float hi = fmax_s(clampT0, clampT1);
    if (hi - lo < 1e-5f) return hi;
    for (int it = 0; it < 10; ++it) {
        float mid = 0.5f * (lo + hi);
        V3 center = from + dir * mid;
        V3 s, t;
        float dist = segmentTriangleDistance(center, halfHeight, v0, v1, v2, s, t);
        if (dist <= radius) hi = mid; else lo = mid;
    }
    return hi;
}

// :1285-1359
static bool sweepCapsuleTriangle(V3 from, V3 dir, float maxDistance, float radius, float halfHeight,
                                 V3 v0, V3 v1, V3 v2, int triangleIndex, int& iterations, CapsuleCastHit& out) {
    float minAdvance = fmax_s(radius * 0.02f, 1e-4f);
    int maxIter = std::min(256, (int)ceilf(maxDistance / minAdvance) + 1);
    const float contactEps = 1e-5f;
    V3 triNormal = normalize(cross(v1 - v0, v2 - v0));
    float t = 0, lastSafeT = 0;
    for (int it = 0; it < maxIter; ++it) {
        iterations += 1;
        if (t > maxDistance) return false;
        V3 center = from + dir * t;
        V3 s0, t0;
        float dist = segmentTriangleDistance(center, halfHeight, v0, v1, v2, s0, t0);
        if (dist <= radius + contactEps) {
            float tHit = refineTOI(from, dir, radius, halfHeight, v0, v1, v2, lastSafeT, t, maxDistance);
            V3 hitCenter = from + dir * tHit;
            V3 hitSeg, hitTri;
            float hitDist = segmentTriangleDistance(hitCenter, halfHeight, v0, v1, v2, hitSeg, hitTri);
            V3 n;
            if (hitDist < 1e-6f) n = dot(triNormal, dir) > 0 ? -triNormal : triNormal;
            else n = normalize(hitSeg - hitTri);
            V3 triN = triNormal;
            if (dot(triN, n) < 0) triN = -triN;
            out = CapsuleCastHit{tHit, hitTri, n, triN, triangleIndex, sge_surface_material{0.8f, 0.6f, 0}};
            return true;
        }
        lastSafeT = t;
        float advance = fmax_s(dist - radius, minAdvance);
        if (advance <= 0) t += minAdvance; else t += advance;
    }
    return false;
}

static inline bool aabbDisjoint(const AABB& b, V3 minP, V3 maxP) {
    return b.max.x < minP.x || b.min.x > maxP.x || b.max.y < minP.y || b.min.y > maxP.y ||
           b.max.z < minP.z || b.min.z > maxP.z;
}

// :1011-1117
static bool capsuleCastBVH(const TriangleMeshSet& set, int triangleIndexOffset, QueryStats& stats, V3 from, V3 delta, float radius,
                           float halfHeight, bool blockingOnly, bool hasMinNormalY, float minNormalY,
                           uint32_t mask, CapsuleCastHit& bestHit) {
    if (!set.hasBVH || set.bvh.root < 0) return false;
    const BVH& bvh = set.bvh;
    float len = length(delta);
    if (len < 1e-6f) return false;
    V3 dir = delta / len;
    V3 up = V3{0, 1, 0};
    V3 a0 = from + up * halfHeight, b0 = from - up * halfHeight;
    V3 a1 = a0 + delta, b1 = b0 + delta;
    V3 minP = vmin(vmin(a0, b0), vmin(a1, b1));
    V3 maxP = vmax(vmax(a0, b0), vmax(a1, b1));
    V3 ext = V3{radius, radius, radius};
    minP -= ext; maxP += ext;
    bool have = false;
    float bestT = len;
    std::vector<int> stack;
    stack.push_back(bvh.root);
    while (!stack.empty()) {
        int nodeIndex = stack.back(); stack.pop_back();
        const BVHNode& node = bvh.nodes[nodeIndex];
        if (aabbDisjoint(node.bounds, minP, maxP)) continue;
        if (node.left < 0) {
            for (int i = node.start; i < node.start + node.count; ++i) {
                int triIndex = bvh.triOrder[i];
                if ((set.triangleLayers[triIndex] & mask) == 0) continue;
                if (aabbDisjoint(set.triangleAABBs[triIndex], minP, maxP)) continue;
                stats.candidates += 1;
                int base = triIndex * 3;
                if (base + 2 >= (int)set.indices.size()) continue;
                V3 v0 = set.positions[set.indices[base]], v1 = set.positions[set.indices[base + 1]], v2 = set.positions[set.indices[base + 2]];
                stats.sweeps += 1;
                int iterCount = 0;
                CapsuleCastHit hit;
                bool got = sweepCapsuleTriangle(from, dir, len, radius, halfHeight, v0, v1, v2, triIndex, iterCount, hit);
                stats.iterations += iterCount; // the reference skips this add on filtered hits (stats only)
                if (got && hit.toi < bestT) {
                    hit.material = set.triangleMaterials[triIndex];
                    hit.triangleIndex = triIndex + triangleIndexOffset;
                    if (blockingOnly) {
                        if (dot(delta, hit.normal) >= 0) continue;
                        if (dot(delta, hit.triangleNormal) >= 0) continue;
                    }
                    if (hasMinNormalY && hit.triangleNormal.y < minNormalY) continue;
                    bestT = hit.toi;
                    bestHit = hit;
                    have = true;
                }
            }
        } else {
            stack.push_back(node.left);
            stack.push_back(node.right);
        }
    }
    return have;
}

// :980-1009
bool CollisionQuery::capsuleCastCombined(V3 from, V3 delta, float radius, float halfHeight, bool blockingOnly,
                                         bool hasMinNormalY, float minNormalY, uint32_t mask,
                                         CapsuleCastHit& out) const {
    float len = length(delta);
    if (len < 1e-6f) return false;
    g_tls_stats.queries += 1;
    CapsuleCastHit a, b;
    bool haveA = capsuleCastBVH(staticSet, 0, g_tls_stats, from, delta, radius, halfHeight, blockingOnly, hasMinNormalY, minNormalY, mask, a);
    bool haveB = capsuleCastBVH(dynamicSet, (int)staticSet.triangleAABBs.size(), g_tls_stats, from, delta, radius, halfHeight,
                                blockingOnly, hasMinNormalY, minNormalY, mask, b);
    if (haveA && haveB) { out = a.toi <= b.toi ? a : b; return true; } // chooseNearest :909-914
    if (haveA) { out = a; return true; }
    if (haveB) { out = b; return true; }
    return false;
}

// :1201-1283
static int capsuleOverlapBVHAll(const TriangleMeshSet& set, int triangleIndexOffset, V3 from, float radius, float halfHeight,
                                int maxHits, uint32_t mask, CapsuleOverlapHit* hits) {
    if (!set.hasBVH || set.bvh.root < 0) return 0;
    const BVH& bvh = set.bvh;
    V3 up = V3{0, 1, 0};
    V3 a0 = from + up * halfHeight, b0 = from - up * halfHeight;
    V3 minP = vmin(a0, b0), maxP = vmax(a0, b0);
    V3 ext = V3{radius, radius, radius};
    minP -= ext; maxP += ext;
    int n = 0;
    std::vector<int> stack;
    stack.push_back(bvh.root);
    while (!stack.empty()) {
        int nodeIndex = stack.back(); stack.pop_back();
        const BVHNode& node = bvh.nodes[nodeIndex];
        if (aabbDisjoint(node.bounds, minP, maxP)) continue;
        if (node.left < 0) {
            for (int i = node.start; i < node.start + node.count; ++i) {
                int triIndex = bvh.triOrder[i];
                if ((set.triangleLayers[triIndex] & mask) == 0) continue;
                if (aabbDisjoint(set.triangleAABBs[triIndex], minP, maxP)) continue;
                int base = triIndex * 3;
                if (base + 2 >= (int)set.indices.size()) continue;
                V3 v0 = set.positions[set.indices[base]], v1 = set.positions[set.indices[base + 1]], v2 = set.positions[set.indices[base + 2]];
                V3 segPoint, triPoint;
                float dist = segmentTriangleDistance(from, halfHeight, v0, v1, v2, segPoint, triPoint);
                if (dist >= radius) continue;
                float depth = radius - dist;
                V3 triNormal = normalize(cross(v1 - v0, v2 - v0));
                V3 nn = dist < 1e-6f ? triNormal : normalize(segPoint - triPoint);
                V3 triN = triNormal;
                if (dot(triN, nn) < 0) triN = -triN;
                hits[n++] = CapsuleOverlapHit{depth, triPoint, nn, triN, triIndex + triangleIndexOffset, set.triangleMaterials[triIndex]};
                if (n >= maxHits) return n;
            }
        } else {
            stack.push_back(node.left);
            stack.push_back(node.right);
        }
    }
    return n;
}

// :852-882 — static hits first, the dynamic set fills what is left of maxHits (so the over-budget sort never triggers)
int CollisionQuery::capsuleOverlapAll(V3 from, float radius, float halfHeight, int maxHits, uint32_t mask,
                                      CapsuleOverlapHit* hits) const {
    if (staticSet.hasBVH || dynamicSet.hasBVH) g_tls_stats.queries += 1;
    int n = capsuleOverlapBVHAll(staticSet, 0, from, radius, halfHeight, maxHits, mask, hits);
    int remaining = maxHits - n;
    if (remaining > 0)
        n += capsuleOverlapBVHAll(dynamicSet, (int)staticSet.triangleAABBs.size(), from, radius, halfHeight, remaining, mask, hits + n);
    return n;
}

// :1119-1199: the deepest overlap of one set; `depth <= bestDepth` keeps the first visited on ties
static bool capsuleOverlapBVH(const TriangleMeshSet& set, int triangleIndexOffset, V3 from, float radius, float halfHeight,
                              uint32_t mask, CapsuleOverlapHit& best) {
    if (!set.hasBVH || set.bvh.root < 0) return false;
    const BVH& bvh = set.bvh;
    V3 up = V3{0, 1, 0};
    V3 a0 = from + up * halfHeight, b0 = from - up * halfHeight;
    V3 minP = vmin(a0, b0), maxP = vmax(a0, b0);
    V3 ext = V3{radius, radius, radius};
    minP -= ext; maxP += ext;
    bool have = false;
    float bestDepth = 0;
    std::vector<int> stack;
    stack.push_back(bvh.root);
    while (!stack.empty()) {
        int nodeIndex = stack.back(); stack.pop_back();
        const BVHNode& node = bvh.nodes[nodeIndex];
        if (aabbDisjoint(node.bounds, minP, maxP)) continue;
        if (node.left < 0) {
            for (int i = node.start; i < node.start + node.count; ++i) {
                int triIndex = bvh.triOrder[i];
                if ((set.triangleLayers[triIndex] & mask) == 0) continue;
                if (aabbDisjoint(set.triangleAABBs[triIndex], minP, maxP)) continue;
                int base = triIndex * 3;
                if (base + 2 >= (int)set.indices.size()) continue;
                V3 v0 = set.positions[set.indices[base]], v1 = set.positions[set.indices[base + 1]], v2 = set.positions[set.indices[base + 2]];
                V3 segPoint, triPoint;
                float dist = segmentTriangleDistance(from, halfHeight, v0, v1, v2, segPoint, triPoint);
                if (dist >= radius) continue;
                float depth = radius - dist;
                if (depth <= bestDepth) continue;
                V3 triNormal = normalize(cross(v1 - v0, v2 - v0));
                V3 nn = dist < 1e-6f ? triNormal : normalize(segPoint - triPoint);
                V3 triN = triNormal;
                if (dot(triN, nn) < 0) triN = -triN;
                bestDepth = depth;
                best = CapsuleOverlapHit{depth, triPoint, nn, triN, triIndex + triangleIndexOffset, set.triangleMaterials[triIndex]};
                have = true;
            }
        } else {
            stack.push_back(node.left);
            stack.push_back(node.right);
        }
    }
    return have;
}

// :830-850
bool CollisionQuery::capsuleOverlap(V3 from, float radius, float halfHeight, uint32_t mask, CapsuleOverlapHit& best) const {
    CapsuleOverlapHit a, b;
    bool haveA = capsuleOverlapBVH(staticSet, 0, from, radius, halfHeight, mask, a);
    bool haveB = capsuleOverlapBVH(dynamicSet, (int)staticSet.triangleAABBs.size(), from, radius, halfHeight, mask, b);
    if (haveA && haveB) { best = a.depth >= b.depth ? a : b; return true; }
    if (haveA) { best = a; return true; }
    if (haveB) { best = b; return true; }
    return false;
}

// ---- raycast (:768-785, 916-978, 1575-1631) ----
static bool rayTriangle(V3 origin, V3 direction, V3 v0, V3 v1, V3 v2, float eps, float& tOut) { // :1575-1601
    V3 e1 = v1 - v0, e2 = v2 - v0;
    V3 pvec = cross(direction, e2);
    float det = dot(e1, pvec);
    if (fabsf(det) < eps) return false;
    float invDet = 1.0f / det;
    V3 tvec = origin - v0;
    float u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    V3 qvec = cross(tvec, e1);
    float v = dot(direction, qvec) * invDet;
    if (v < 0 || (u + v) > 1) return false;
    float t = dot(e2, qvec) * invDet;
    if (!(t >= 0)) return false;
    tOut = t;
    return true;
}
static bool rayAABB(V3 origin, V3 direction, const AABB& bounds, float& tminOut) { // :1603-1630
    const float big = std::numeric_limits<float>::max();
    float invX = direction.x != 0 ? 1.0f / direction.x : big;
    float invY = direction.y != 0 ? 1.0f / direction.y : big;
    float invZ = direction.z != 0 ? 1.0f / direction.z : big;
    float tmin = (bounds.min.x - origin.x) * invX, tmax = (bounds.max.x - origin.x) * invX;
    if (tmin > tmax) std::swap(tmin, tmax);
    float tymin = (bounds.min.y - origin.y) * invY, tymax = (bounds.max.y - origin.y) * invY;
    if (tymin > tymax) std::swap(tymin, tymax);
    if (tmin > tymax || tymin > tmax) return false;
    tmin = fmax_s(tmin, tymin);
    tmax = fmin_s(tmax, tymax);
    float tzmin = (bounds.min.z - origin.z) * invZ, tzmax = (bounds.max.z - origin.z) * invZ;
    if (tzmin > tzmax) std::swap(tzmin, tzmax);
    if (tmin > tzmax || tzmin > tmax) return false;
    tmin = fmax_s(tmin, tzmin);
    tminOut = tmin;
    return true;
}
static bool raycastBVH(const TriangleMeshSet& set, int triangleIndexOffset, V3 origin, V3 direction, float maxDistance,
                       uint32_t mask, RaycastHit& hit) { // :916-978
    if (!set.hasBVH || set.bvh.root < 0) return false;
    const BVH& bvh = set.bvh;
    const float eps = 1e-6f;
    float closestT = maxDistance;
    bool have = false;
    std::vector<int> stack{bvh.root};
    while (!stack.empty()) {
        int nodeIndex = stack.back(); stack.pop_back();
        const BVHNode& node = bvh.nodes[nodeIndex];
        float rangeMin;
        if (!rayAABB(origin, direction, node.bounds, rangeMin)) continue;
        if (rangeMin > closestT) continue;
        if (node.left < 0) {
            for (int i = node.start; i < node.start + node.count; ++i) {
                int triIndex = bvh.triOrder[i];
                if ((set.triangleLayers[triIndex] & mask) == 0) continue;
                int base = triIndex * 3;
                if (base + 2 >= (int)set.indices.size()) continue;
                V3 v0 = set.positions[set.indices[base]], v1 = set.positions[set.indices[base + 1]], v2 = set.positions[set.indices[base + 2]];
                float t;
                if (rayTriangle(origin, direction, v0, v1, v2, eps, t) && t < closestT) {
                    V3 n = normalize(cross(v1 - v0, v2 - v0));
                    V3 normal = dot(n, direction) > 0 ? -n : n;
                    closestT = t;
                    hit = RaycastHit{t, origin + direction * t, normal, triIndex + triangleIndexOffset, set.triangleMaterials[triIndex]};
                    have = true;
                }
            }
        } else {
            stack.push_back(node.left);
            stack.push_back(node.right);
        }
    }
    return have;
}
bool CollisionQuery::raycast(V3 origin, V3 direction, float maxDistance, uint32_t mask, RaycastHit& out) const {
    RaycastHit a, b;
    bool haveA = raycastBVH(staticSet, 0, origin, direction, maxDistance, mask, a);
    bool haveB = raycastBVH(dynamicSet, (int)staticSet.triangleAABBs.size(), origin, direction, maxDistance, mask, b);
    if (haveA && haveB) { out = a.distance <= b.distance ? a : b; return true; }
    if (haveA) { out = a; return true; }
    if (haveB) { out = b; return true; }
    return false;
}

} // namespace sgeo
