// ORACLE — TEST INFRASTRUCTURE ONLY (see sge_oracle.h).
//
// The step after skinning, restated without any acceleration structure:
//   RTAccelerationBuilder.swift:75-145   the reference hands the skinned vertex buffer + the item's index slice to
//                                        Metal (build once, refit per frame); the structure itself is opaque, so the
//                                        oracle of "what a ray sees" is a scan over every triangle of the index buffer.
//   RayTracing.metalinc:242-296          what the raytraceKernel reads at a hit: the three skinned vertices through the
//                                        index buffer, the geometric normal of the world-space triangle flipped against
//                                        the ray, and the barycentric blend of the skinned normals / tangents (nW, tW, bW).
// Ray-triangle test: the engine's own rayTriangle (CollisionQuery.swift:1575-1601), which also yields the barycentrics;
// Metal's intersector is not specified to the bit -> parity unpinned (the product is checked against this scan).
#include <cstring>
#include "sge_oracle.h"

namespace sgeo {

static bool rayTriangleUV(V3 origin, V3 direction, V3 v0, V3 v1, V3 v2, float eps, float& tOut, float& uOut, float& vOut) {
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
    tOut = t + 0.0f;
    uOut = u;
    vOut = v;
    return true;
}

static void intersectInstance(const World& w, const sge_blas_ray& R, int instance, sge_blas_hit& H);

// instance < 0: every character in ascending order, a later one winning only with a strictly smaller distance
void blas_intersect(const World& w, const sge_blas_ray& R, sge_blas_hit& H) {
    std::memset(&H, 0, sizeof(H));
    H.primitive = -1;
    H.instance = -1;
    const int N = (int)w.bodies.size();
    if (R.instance >= N || w.blasIndices.empty()) return;
    if (R.instance >= 0) { intersectInstance(w, R, R.instance, H); return; }
    for (int i = 0; i < N; ++i) {
        sge_blas_hit h;
        intersectInstance(w, R, i, h);
        if (h.hit && (!H.hit || h.distance < H.distance)) H = h;
    }
}

static void intersectInstance(const World& w, const sge_blas_ray& R0, int instance, sge_blas_hit& H) {
    std::memset(&H, 0, sizeof(H));
    H.primitive = -1;
    H.instance = -1;
    sge_blas_ray R = R0;
    R.instance = instance;
    const int V = w.mesh.vertexCount;
    M4 M = m4_identity();
    if ((size_t)(R.instance + 1) * 16 <= w.blasInstances.size()) std::memcpy(&M, &w.blasInstances[(size_t)R.instance * 16], 64);
    const V3 a{M.c[0].x, M.c[0].y, M.c[0].z}, b{M.c[1].x, M.c[1].y, M.c[1].z}, c{M.c[2].x, M.c[2].y, M.c[2].z}, tr{M.c[3].x, M.c[3].y, M.c[3].z};
    // object-space ray: inverse of the 3x3 part by cofactors
    const V3 bc = cross(b, c), ca = cross(c, a), ab = cross(a, b);
    const float r = 1.0f / dot(a, bc);
    const V3 r0 = bc * r, r1 = ca * r, r2 = ab * r;
    const V3 wo{R.origin[0], R.origin[1], R.origin[2]}, wd{R.direction[0], R.direction[1], R.direction[2]};
    const V3 rel = wo - tr;
    const V3 o{dot(r0, rel), dot(r1, rel), dot(r2, rel)}, d{dot(r0, wd), dot(r1, wd), dot(r2, wd)};
    const float tMin = fmax_s(R.minDistance, 0.0f);
    const float* P = w.outPositions.data() + (size_t)R.instance * V * 3;
    const float* NB = w.outNormals.data() + (size_t)R.instance * V * 3;
    const float* TB = w.outTangents.data() + (size_t)R.instance * V * 4;
    auto ld = [](const float* base, uint32_t i) { return V3{base[(size_t)i * 3], base[(size_t)i * 3 + 1], base[(size_t)i * 3 + 2]}; };
    const int T = (int)w.blasIndices.size() / 3;
    bool found = false;
    float bestT = 0, bestU = 0, bestV = 0;
    int bestPrim = -1;
    for (int p = 0; p < T; ++p) { // ascending primitive id: ties on distance keep the smaller id
        const uint32_t* ix = &w.blasIndices[(size_t)p * 3];
        float t, u, v;
        if (!rayTriangleUV(o, d, ld(P, ix[0]), ld(P, ix[1]), ld(P, ix[2]), 1e-6f, t, u, v)) continue;
        if (!(t >= tMin && t <= R.maxDistance)) continue;
        if (!found || t < bestT) { found = true; bestT = t; bestU = u; bestV = v; bestPrim = p; }
    }
    if (!found) return;
    const uint32_t* ix = &w.blasIndices[(size_t)bestPrim * 3];
    auto xf = [&](V3 p) { return ((a * p.x + b * p.y) + c * p.z) + tr; };
    auto rot = [&](V3 p) { return (a * p.x + b * p.y) + c * p.z; };
    const V3 w0 = xf(ld(P, ix[0])), w1 = xf(ld(P, ix[1])), w2 = xf(ld(P, ix[2]));
    V3 Ng = normalize(cross(w1 - w0, w2 - w0)); // :262-264
    if (dot(Ng, wd) > 0.0f) Ng = -Ng;
    const float bx = bestU, by = bestV, bw = 1.0f - bx - by; // :285
    const V3 nObj = normalize((ld(NB, ix[0]) * bw + ld(NB, ix[1]) * bx) + ld(NB, ix[2]) * by); // :292
    float t4[4];
    for (int k = 0; k < 4; ++k) t4[k] = (TB[(size_t)ix[0] * 4 + k] * bw + TB[(size_t)ix[1] * 4 + k] * bx) + TB[(size_t)ix[2] * 4 + k] * by;
    const float r4 = 1.0f / sqrtf(((t4[0] * t4[0] + t4[1] * t4[1]) + t4[2] * t4[2]) + t4[3] * t4[3]); // :293 normalize(float4)
    const V3 tObj = normalize(V3{t4[0] * r4, t4[1] * r4, t4[2] * r4}); // :294
    const float tw = t4[3] * r4;
    const V3 nW = normalize(rot(nObj)), tW = normalize(rot(tObj)); // :298-299
    const V3 bW = normalize(cross(nW, tW) * tw);                   // :300
    H.hit = 1; H.primitive = bestPrim; H.instance = R.instance; H.distance = bestT; H.bary[0] = bx; H.bary[1] = by;
    H.geomNormal[0] = Ng.x; H.geomNormal[1] = Ng.y; H.geomNormal[2] = Ng.z;
    H.normal[0] = nW.x; H.normal[1] = nW.y; H.normal[2] = nW.z;
    H.tangent[0] = tW.x; H.tangent[1] = tW.y; H.tangent[2] = tW.z;
    H.bitangent[0] = bW.x; H.bitangent[1] = bW.y; H.bitangent[2] = bW.z;
    if (w.blasUVs.size() == (size_t)V * 2) { // interp_uv, RayTracing.metalinc:106-119
        const float* uv = w.blasUVs.data();
        for (int k = 0; k < 2; ++k) H.uv[k] = (uv[(size_t)ix[0] * 2 + k] * bw + uv[(size_t)ix[1] * 2 + k] * bx) + uv[(size_t)ix[2] * 2 + k] * by;
    }
}

} // namespace sgeo
