// Primitive distance queries of the capsule sweep (CollisionQuery.swift:1396-1573), host + device.
//
// Same arithmetic as the reference on every input — each lane computes exactly the values its branch of the
// reference would — but written without data-dependent branches around the divisions: in a wavefront the lanes sit
// in different Voronoi regions, so branchy code executes every region's IEEE division sequence (4 per
// closestPointOnTriangle, up to 5 per segmentSegmentDistanceSq, ~21 per capsule-triangle distance); here the
// numerator and denominator are selected first and ONE division serves all regions (9 per distance).
// `*Branchy` are the reference-shaped forms, kept for the equivalence fuzz test (tools/fuzz_prims.cpp).
#pragma once
#include "sge_math.hpp"

namespace sge {

SGE_HD F3 sel(bool c, F3 a, F3 b) { return F3{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }

SGE_HD bool segmentTriangleIntersect(F3 a, F3 b, F3 v0, F3 v1, F3 v2, F3& out) { // :1440
    F3 dir = b - a;
    const float eps = 1e-6f;
    F3 e1 = v1 - v0, e2 = v2 - v0;
    F3 pvec = cross(dir, e2);
    float det = dot(e1, pvec);
    if (fabsf(det) < eps) return false;
    float invDet = 1.0f / det;
    F3 tvec = a - v0;
    float u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    F3 qvec = cross(tvec, e1);
    float v = dot(dir, qvec) * invDet;
    if (v < 0 || (u + v) > 1) return false;
    float t = dot(e2, qvec) * invDet;
    if (t < 0 || t > 1) return false;
    out = a + dir * t;
    return true;
}

// :1464-1517. Regions in the reference's test order: vertex A, vertex B, edge AB, vertex C, edge AC, edge BC, face.
SGE_HD float closestPointOnTriangle(F3 p, F3 a, F3 b, F3 c, F3& point) {
    const F3 ab = b - a, ac = c - a, ap = p - a, bp = p - b, cp = p - c;
    const float d1 = dot(ab, ap), d2 = dot(ac, ap);
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    const float d5 = dot(ab, cp), d6 = dot(ac, cp);
    const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    const bool rA = d1 <= 0 && d2 <= 0;
    const bool rB = !rA && (d3 >= 0 && d4 <= d3);
    const bool rAB = !rA && !rB && (vc <= 0 && d1 >= 0 && d3 <= 0);
    const bool rC = !rA && !rB && !rAB && (d6 >= 0 && d5 <= d6);
    const bool rAC = !rA && !rB && !rAB && !rC && (vb <= 0 && d2 >= 0 && d6 <= 0);
    const float d43 = d4 - d3, d56 = d5 - d6;
    const bool rBC = !rA && !rB && !rAB && !rC && !rAC && (va <= 0 && d43 >= 0 && d56 >= 0);
    const bool vertex = rA || rB || rC;
    // one division: d1/(d1-d3) | d2/(d2-d6) | (d4-d3)/((d4-d3)+(d5-d6)) | 1/(va+vb+vc)
    const float num = vertex ? 0.0f : (rAB ? d1 : (rAC ? d2 : (rBC ? d43 : 1.0f)));
    const float den = vertex ? 1.0f : (rAB ? d1 - d3 : (rAC ? d2 - d6 : (rBC ? d43 + d56 : (va + vb) + vc)));
    const float q = num / den;
    const bool face = !vertex && !rAB && !rAC && !rBC;
    // edge: base + dir * q; face: (a + ab * (vb*q)) + ac * (vc*q)
    const F3 base = rBC ? b : a;
    const F3 dir = rAC ? ac : (rBC ? c - b : ab);
    const float s1 = face ? vb * q : q;
    const F3 onEdge = base + dir * s1;
    const F3 onFace = onEdge + ac * (vc * q);
    const F3 vert = rA ? a : (rB ? b : c);
    point = sel(vertex, vert, sel(face, onFace, onEdge));
    return lengthSq(p - point);
}

// :1519-1569
SGE_HD float segmentSegmentDistanceSq(F3 p1, F3 q1, F3 p2, F3 q2, F3& c1o, F3& c2o) {
    const F3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    const float eps = 1e-6f;
    const bool aDeg = a <= eps, eDeg = e <= eps;
    const bool onlyA = aDeg && !eDeg, onlyE = !aDeg && eDeg, general = !aDeg && !eDeg;
    const float c = dot(d1, r), b = dot(d1, d2);
    const float denom = a * e - b * b;
    const bool haveDenom = general && denom != 0;
    // first division: f/e (first segment degenerate) | -c/a (second degenerate) | (b*f - c*e)/denom
    const float n1 = onlyA ? f : (onlyE ? -c : (haveDenom ? b * f - c * e : 0.0f));
    const float m1 = onlyA ? e : (onlyE ? a : (haveDenom ? denom : 1.0f));
    const float q1v = sclamp(n1 / m1, 0, 1);
    const float s0 = haveDenom ? q1v : 0.0f; // general case only
    const float tNom = b * s0 + f;
    const bool below = general && tNom < 0, above = general && !below && tNom > e, inside = general && !below && !above;
    // second division (general case): -c/a | (b-c)/a | tNom/e
    const float n2 = below ? -c : (above ? b - c : (inside ? tNom : 0.0f));
    const float m2 = (below || above) ? a : (inside ? e : 1.0f);
    const float q2v = n2 / m2;
    const float s = onlyE ? q1v : (general ? (inside ? s0 : sclamp(q2v, 0, 1)) : 0.0f);
    const float t = onlyA ? q1v : (general ? (below ? 0.0f : (above ? 1.0f : q2v)) : 0.0f);
    const F3 c1 = sel(aDeg, p1, p1 + d1 * s);
    const F3 c2 = sel(eDeg, p2, p2 + d2 * t);
    c1o = c1; c2o = c2;
    return lengthSq(c1 - c2);
}

SGE_HD float segmentTriangleDistance(F3 center, float halfHeight, F3 v0, F3 v1, F3 v2, F3& segPoint, F3& triPoint) { // :1396
    F3 up{0, 1, 0};
    F3 a = center + up * halfHeight;
    F3 b = center - up * halfHeight;
    F3 hit;
    if (segmentTriangleIntersect(a, b, v0, v1, v2, hit)) { segPoint = hit; triPoint = hit; return 0; }
    float bestDistSq = kFloatMax;
    F3 bestSeg = a, bestTri = v0;
    F3 p0, p1;
    float d0 = closestPointOnTriangle(a, v0, v1, v2, p0);
    if (d0 < bestDistSq) { bestDistSq = d0; bestSeg = a; bestTri = p0; }
    float dd1 = closestPointOnTriangle(b, v0, v1, v2, p1);
    if (dd1 < bestDistSq) { bestDistSq = dd1; bestSeg = b; bestTri = p1; }
    F3 s, t;
    float d = segmentSegmentDistanceSq(a, b, v0, v1, s, t);
    if (d < bestDistSq) { bestDistSq = d; bestSeg = s; bestTri = t; }
    d = segmentSegmentDistanceSq(a, b, v1, v2, s, t);
    if (d < bestDistSq) { bestDistSq = d; bestSeg = s; bestTri = t; }
    d = segmentSegmentDistanceSq(a, b, v2, v0, s, t);
    if (d < bestDistSq) { bestDistSq = d; bestSeg = s; bestTri = t; }
    segPoint = bestSeg; triPoint = bestTri;
    return sqrtf(smax(bestDistSq, 0.0f));
}

// ---- reference-shaped forms (equivalence test only) -----------------------------------------------------------------
SGE_HD float closestPointOnTriangleBranchy(F3 p, F3 a, F3 b, F3 c, F3& point) {
    F3 ab = b - a, ac = c - a, ap = p - a;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) { point = a; return lengthSq(p - a); }
    F3 bp = p - b;
    float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) { point = b; return lengthSq(p - b); }
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) {
        float v = d1 / (d1 - d3);
        point = a + ab * v;
        return lengthSq(p - point);
    }
    F3 cp = p - c;
    float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) { point = c; return lengthSq(p - c); }
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) {
        float w = d2 / (d2 - d6);
        point = a + ac * w;
        return lengthSq(p - point);
    }
    float va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
        float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        point = b + (c - b) * w;
        return lengthSq(p - point);
    }
    float denom = 1.0f / (va + vb + vc);
    float v = vb * denom, w = vc * denom;
    point = (a + ab * v) + ac * w;
    return lengthSq(p - point);
}

SGE_HD float segmentSegmentDistanceSqBranchy(F3 p1, F3 q1, F3 p2, F3 q2, F3& c1o, F3& c2o) {
    F3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    float s = 0, t = 0;
    const float eps = 1e-6f;
    if (a <= eps && e <= eps) { c1o = p1; c2o = p2; return lengthSq(p1 - p2); }
    if (a <= eps) {
        t = sclamp(f / e, 0, 1);
        F3 c2 = p2 + d2 * t;
        c1o = p1; c2o = c2;
        return lengthSq(p1 - c2);
    }
    float c = dot(d1, r);
    if (e <= eps) {
        s = sclamp(-c / a, 0, 1);
        F3 c1 = p1 + d1 * s;
        c1o = c1; c2o = p2;
        return lengthSq(c1 - p2);
    }
    float b = dot(d1, d2);
    float denom = a * e - b * b;
    if (denom != 0) s = sclamp((b * f - c * e) / denom, 0, 1);
    else s = 0;
    float tNom = b * s + f;
    if (tNom < 0) { t = 0; s = sclamp(-c / a, 0, 1); }
    else if (tNom > e) { t = 1; s = sclamp((b - c) / a, 0, 1); }
    else t = tNom / e;
    F3 c1 = p1 + d1 * s, c2 = p2 + d2 * t;
    c1o = c1; c2o = c2;
    return lengthSq(c1 - c2);
}

} // namespace sge
