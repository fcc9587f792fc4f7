// Float32 vector / affine / quaternion helpers shared by the host code and the
// gfx950 kernels of libsge_amd.so.
//
// The arithmetic mirrors Apple simd as the reference uses it (column-major
// matrices, Hamilton quaternions stored (ix,iy,iz,r), simd_reduce_add order for
// dot products).  Operation order is explicit; the CCD and pose translation
// units are compiled with -ffp-contract=off so what is written is what runs.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define SGE_HD __host__ __device__ __forceinline__

namespace sge {

// Swift's Float.pi is rounded toward zero (0x40490FDA), one ulp below (float)M_PI.
constexpr float kSwiftPi = 0x1.921fb4p+1f;
constexpr float kFloatMax = 3.40282347e+38f; // Float.greatestFiniteMagnitude

struct F3 { float x, y, z; };
struct F4 { float x, y, z, w; };
struct D3 { double x, y, z; };
struct Quat { float x, y, z, w; };
// Affine 3x4 (rows of a simd float4x4 whose last row is 0,0,0,1), stored by COLUMN.
struct Aff { F3 c0, c1, c2, c3; };

// Swift's generic max(x,y) = y >= x ? y : x and min(x,y) = y < x ? y : x
SGE_HD float smax(float x, float y) { return y >= x ? y : x; }
SGE_HD float smin(float x, float y) { return y < x ? y : x; }
SGE_HD float sclamp(float v, float lo, float hi) { return smin(smax(v, lo), hi); }

SGE_HD F3 f3(float x, float y, float z) { return F3{x, y, z}; }
SGE_HD F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
SGE_HD F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
SGE_HD F3 operator-(F3 a) { return F3{-a.x, -a.y, -a.z}; }
SGE_HD F3 operator*(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
SGE_HD F3 operator*(float s, F3 a) { return F3{a.x * s, a.y * s, a.z * s}; }
SGE_HD F3 operator/(F3 a, float s) { return F3{a.x / s, a.y / s, a.z / s}; }
SGE_HD float dot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
SGE_HD float lengthSq(F3 a) { return dot(a, a); }
SGE_HD float length(F3 a) { return sqrtf(dot(a, a)); }
SGE_HD F3 cross(F3 a, F3 b) { return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
SGE_HD F3 normalize(F3 a) { float r = 1.0f / sqrtf(dot(a, a)); return a * r; }
SGE_HD F3 vmin(F3 a, F3 b) { return F3{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
SGE_HD F3 vmax(F3 a, F3 b) { return F3{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }

SGE_HD D3 toD(F3 v) { return D3{(double)v.x, (double)v.y, (double)v.z}; }
SGE_HD F3 toF(D3 v) { return F3{(float)v.x, (float)v.y, (float)v.z}; }
SGE_HD D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
SGE_HD D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
SGE_HD D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
SGE_HD D3 operator/(D3 a, double s) { return D3{a.x / s, a.y / s, a.z / s}; }
SGE_HD double dot(D3 a, D3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
SGE_HD double length(D3 a) { return sqrt(dot(a, a)); }

// simd_reduce_add(float4) = (x0 + x2) + (x1 + x3)
SGE_HD float dot4(F4 a, F4 b) { return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w); }

// ---- 3x4 affine algebra, bit-compatible with full 4x4 simd_mul on (0,0,0,1)-row matrices ----
// rotation part * vector, with the "+ c3*0" term a 4x4 multiply would add
SGE_HD F3 rotMulDir(const Aff& m, F3 v) { return ((m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z) + m.c3 * 0.0f; }
SGE_HD F3 affMulPoint(const Aff& m, F3 v) { return ((m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z) + m.c3; }
// a*b where both have last row (0,0,0,1)
SGE_HD Aff affMul(const Aff& a, const Aff& b) {
    Aff r;
    r.c0 = ((a.c0 * b.c0.x + a.c1 * b.c0.y) + a.c2 * b.c0.z) + a.c3 * 0.0f;
    r.c1 = ((a.c0 * b.c1.x + a.c1 * b.c1.y) + a.c2 * b.c1.z) + a.c3 * 0.0f;
    r.c2 = ((a.c0 * b.c2.x + a.c1 * b.c2.y) + a.c2 * b.c2.z) + a.c3 * 0.0f;
    r.c3 = ((a.c0 * b.c3.x + a.c1 * b.c3.y) + a.c2 * b.c3.z) + a.c3 * 1.0f;
    return r;
}
SGE_HD Aff affIdentity() { return Aff{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}}; }

// Math.swift:11-24 matrix4x4_rotation about a unit coordinate axis given as a vector
SGE_HD Aff axisRotation(float radians, F3 axis) {
    F3 u = normalize(axis);
    float ct = cosf(radians), st = sinf(radians);
    float ci = 1 - ct;
    float x = u.x, y = u.y, z = u.z;
    Aff m;
    m.c0 = F3{ct + x * x * ci, y * x * ci + z * st, z * x * ci - y * st};
    m.c1 = F3{x * y * ci - z * st, ct + y * y * ci, z * y * ci + x * st};
    m.c2 = F3{x * z * ci + y * st, y * z * ci - x * st, ct + z * z * ci};
    m.c3 = F3{0, 0, 0};
    return m;
}
SGE_HD float radiansFromDegrees(float deg) { return (deg / 180.0f) * kSwiftPi; } // Math.swift:48-50
// Skeleton.rotationXYZDegrees (Skeleton.swift:212-217) = Rz * (Ry * Rx)
SGE_HD Aff rotationXYZDegrees(F3 deg) {
    Aff rx = axisRotation(radiansFromDegrees(deg.x), F3{1, 0, 0});
    Aff ry = axisRotation(radiansFromDegrees(deg.y), F3{0, 1, 0});
    Aff rz = axisRotation(radiansFromDegrees(deg.z), F3{0, 0, 1});
    return affMul(rz, affMul(ry, rx));
}

// ---- quaternions: the published <simd/quaternion.h> algorithms ----
SGE_HD F3 qImag(Quat q) { return F3{q.x, q.y, q.z}; }
SGE_HD Quat quatAngleAxis(float angle, F3 axis) {
    float h = angle / 2;
    float s = sinf(h), c = cosf(h);
    return Quat{s * axis.x, s * axis.y, s * axis.z, c};
}
SGE_HD Quat quatFromRotation(const Aff& m) {
    float m00 = m.c0.x, m01 = m.c0.y, m02 = m.c0.z;
    float m10 = m.c1.x, m11 = m.c1.y, m12 = m.c1.z;
    float m20 = m.c2.x, m21 = m.c2.y, m22 = m.c2.z;
    float trace = m00 + m11 + m22;
    if (trace >= 0.0f) {
        float r = 2 * sqrtf(1 + trace);
        float rinv = 1.0f / r;
        return Quat{rinv * (m12 - m21), rinv * (m20 - m02), rinv * (m01 - m10), r / 4};
    } else if (m00 >= m11 && m00 >= m22) {
        float r = 2 * sqrtf(1 - m11 - m22 + m00);
        float rinv = 1.0f / r;
        return Quat{r / 4, rinv * (m01 + m10), rinv * (m02 + m20), rinv * (m12 - m21)};
    } else if (m11 >= m22) {
        float r = 2 * sqrtf(1 - m00 - m22 + m11);
        float rinv = 1.0f / r;
        return Quat{rinv * (m01 + m10), r / 4, rinv * (m12 + m21), rinv * (m20 - m02)};
    } else {
        float r = 2 * sqrtf(1 - m00 - m11 + m22);
        float rinv = 1.0f / r;
        return Quat{rinv * (m02 + m20), rinv * (m12 + m21), r / 4, rinv * (m01 - m10)};
    }
}
// rotation part of matrix_float4x4(simd_quatf); c3 = 0
SGE_HD Aff rotationFromQuat(Quat v) {
    Aff r;
    r.c0 = F3{1 - 2 * (v.y * v.y + v.z * v.z), 2 * (v.x * v.y + v.z * v.w), 2 * (v.x * v.z - v.y * v.w)};
    r.c1 = F3{2 * (v.x * v.y - v.z * v.w), 1 - 2 * (v.z * v.z + v.x * v.x), 2 * (v.y * v.z + v.x * v.w)};
    r.c2 = F3{2 * (v.z * v.x + v.y * v.w), 2 * (v.y * v.z - v.x * v.w), 1 - 2 * (v.y * v.y + v.x * v.x)};
    r.c3 = F3{0, 0, 0};
    return r;
}
SGE_HD float quatLengthSq(Quat q) { return dot4(F4{q.x, q.y, q.z, q.w}, F4{q.x, q.y, q.z, q.w}); }
SGE_HD Quat quatInverse(Quat q) {
    float r = 1.0f / quatLengthSq(q);
    return Quat{-q.x * r, -q.y * r, -q.z * r, q.w * r};
}
SGE_HD Quat quatMul(Quat p, Quat q) {
    float ax = q.w * p.x + q.z * p.y, ay = -q.z * p.x + q.w * p.y, az = q.y * p.x + -q.x * p.y, aw = -q.x * p.x + -q.y * p.y;
    float bx = -q.y * p.z + q.x * p.w, by = q.x * p.z + q.y * p.w, bz = q.w * p.z + q.z * p.w, bw = -q.z * p.z + q.w * p.w;
    return Quat{ax + bx, ay + by, az + bz, aw + bw};
}
SGE_HD Quat quatNormalize(Quat q) {
    float r = 1.0f / sqrtf(quatLengthSq(q));
    return Quat{q.x * r, q.y * r, q.z * r, q.w * r};
}
SGE_HD F3 quatAct(Quat q, F3 v) {
    F3 t = 2.0f * cross(qImag(q), v);
    return (v + t * q.w) + cross(qImag(q), t);
}
SGE_HD float simdSinc(float x) { return x == 0 ? 1.0f : sinf(x) / x; }
SGE_HD Quat slerpInternal(Quat q0, Quat q1, float t) {
    float s = 1 - t;
    F4 d = F4{q0.x - q1.x, q0.y - q1.y, q0.z - q1.z, q0.w - q1.w};
    F4 u = F4{q0.x + q1.x, q0.y + q1.y, q0.z + q1.z, q0.w + q1.w};
    float a = 2 * atan2f(sqrtf(dot4(d, d)), sqrtf(dot4(u, u)));
    float r = 1.0f / simdSinc(a);
    float k0 = simdSinc(s * a) * r * s;
    float k1 = simdSinc(t * a) * r * t;
    Quat q = Quat{k0 * q0.x + k1 * q1.x, k0 * q0.y + k1 * q1.y, k0 * q0.z + k1 * q1.z, k0 * q0.w + k1 * q1.w};
    return quatNormalize(q);
}
SGE_HD Quat quatSlerp(Quat q0, Quat q1, float t) {
    float d = dot4(F4{q0.x, q0.y, q0.z, q0.w}, F4{q1.x, q1.y, q1.z, q1.w});
    if (d >= 0) return slerpInternal(q0, q1, t);
    return slerpInternal(q0, Quat{-q1.x, -q1.y, -q1.z, -q1.w}, t);
}

} // namespace sge
