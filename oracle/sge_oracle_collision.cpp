// ORACLE — TEST INFRASTRUCTURE ONLY (see sge_oracle_math.h).
//
// CPU float32 restatement of Game/CollisionQuery.swift (static set only; the
// dynamic set / refit path, CollisionQuery.swift:419-462,528-575, is row f4 of
// SURVEY.md §8 and not part of the benchmark configs):
//   TriangleMeshSet.rebuild            :331-417
//   BVH.build + helpers                :577-706
//   capsuleCastCombined / capsuleCastBVH   :980-1117
//   capsuleOverlapAll / capsuleOverlapBVHAll :852-882, :1201-1283
//   sweepCapsuleTriangle / refineTOI   :1285-1394
//   segmentTriangleDistance and helpers :1396-1573
// Parity unpinned: the reference holds no test or golden vector for any of it
// (GameTests/GameTests.swift:12-16 is empty); tests pin this file with analytic
// known answers and brute-force cross-checks instead.
#include <algorithm>
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
    const float areaEps = 1e-10f;
    for (int e = 0; e < count; ++e) {
        const sge_static_mesh_entity& m = ents[e];
        M4 model;
        for (int k = 0; k < 16; ++k) (&model.c[0].x)[k] = m.modelMatrix[k];
        uint32_t baseVertex = (uint32_t)positions.size();
        for (int v = 0; v < m.vertexCount; ++v) {
            V4 p = V4{m.positions[v * 3], m.positions[v * 3 + 1], m.positions[v * 3 + 2], 1};
            V4 wp = mul(model, p);
            positions.push_back(V3{wp.x, wp.y, wp.z});
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
    }
    hasBVH = !triangleAABBs.empty();
    if (hasBVH) bvh.build(triangleAABBs);
    else { bvh.nodes.clear(); bvh.triOrder.clear(); bvh.triLeaf.clear(); bvh.root = -1; }
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

// :1011-1117 (static set; triangleIndexOffset 0)
static bool capsuleCastBVH(const TriangleMeshSet& set, QueryStats& stats, V3 from, V3 delta, float radius,
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
                    hit.triangleIndex = triIndex;
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
    return capsuleCastBVH(staticSet, g_tls_stats, from, delta, radius, halfHeight, blockingOnly, hasMinNormalY, minNormalY, mask, out);
}

// :852-882 + :1201-1283 (static set only, so the over-budget sort never triggers)
int CollisionQuery::capsuleOverlapAll(V3 from, float radius, float halfHeight, int maxHits, uint32_t mask,
                                      CapsuleOverlapHit* hits) const {
    const TriangleMeshSet& set = staticSet;
    if (!set.hasBVH || set.bvh.root < 0) return 0;
    g_tls_stats.queries += 1;
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
                hits[n++] = CapsuleOverlapHit{depth, triPoint, nn, triN, triIndex, set.triangleMaterials[triIndex]};
                if (n >= maxHits) return n;
            }
        } else {
            stack.push_back(node.left);
            stack.push_back(node.right);
        }
    }
    return n;
}

// :830-850 + :1119-1199 (static set): the deepest overlap; `depth <= bestDepth` keeps the first visited on ties
bool CollisionQuery::capsuleOverlap(V3 from, float radius, float halfHeight, uint32_t mask, CapsuleOverlapHit& best) const {
    const TriangleMeshSet& set = staticSet;
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
                best = CapsuleOverlapHit{depth, triPoint, nn, triN, triIndex, set.triangleMaterials[triIndex]};
                have = true;
            }
        } else {
            stack.push_back(node.left);
            stack.push_back(node.right);
        }
    }
    return have;
}

} // namespace sgeo
