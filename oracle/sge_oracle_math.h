// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU float32 restatement of the arithmetic the reference's hot path leans on.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load anything built from oracle/. The product (swift-game-engine_amd/) never
// includes, links or calls this code.
//
// Parity status: the reference computes with Apple `simd` (macOS SDK, not
// vendored, no lockfile; call sites listed in SURVEY.md §8c). That library is
// absent here, so the functions below restate the published SDK header
// algorithms with a FIXED operation order and NO fused multiply-add
// (build with -ffp-contract=off). Last-bit agreement with a Mac is therefore
// "parity unpinned"; the pose chain is additionally pinned in float64 against
// the reference's own Tools/FitMotion/fit_motion.py (tests/golden/), and the
// quaternion path (matrix -> quaternion, slerp, yaw-stable root, run lean, action
// layer) against those vectors blended with scipy's slerp (tests/test_oracle_golden.py).
//
// Conventions: matrices are column-major 4x4 (columns[c][r]) exactly like
// simd::float4x4; quaternions are stored (ix, iy, iz, r) like simd_quatf.
#pragma once
#include <cmath>
#include <cstdint>

namespace sgeo {

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct D3 { double x, y, z; };
struct M4 { V4 c[4]; };          // c[col]
struct Q4 { float x, y, z, w; }; // (imag, real)

// Swift's generic max/min: max(x,y) = y >= x ? y : x ; min(x,y) = y < x ? y : x
static inline float fmax_s(float x, float y) { return y >= x ? y : x; }
static inline float fmin_s(float x, float y) { return y < x ? y : x; }
static inline double dmax_s(double x, double y) { return y >= x ? y : x; }
static inline double dmin_s(double x, double y) { return y < x ? y : x; }
static inline float clampf(float v, float lo, float hi) { return fmin_s(fmax_s(v, lo), hi); } // CollisionQuery.swift:1571

static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(float s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
static inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
static inline V3& operator-=(V3& a, V3 b) { a = a - b; return a; }
// simd_reduce_add(float3) = (x + y) + z
static inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float length_squared(V3 a) { return dot(a, a); }
static inline float length(V3 a) { return sqrtf(length_squared(a)); }
static inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// simd_normalize(x) = x * rsqrt(length_squared(x)); restated with an exact 1/sqrt.
static inline V3 normalize(V3 a) { float r = 1.0f / sqrtf(length_squared(a)); return a * r; }
static inline V3 vmin(V3 a, V3 b) { return V3{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
static inline V3 vmax(V3 a, V3 b) { return V3{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }

static inline D3 d3(V3 v) { return D3{(double)v.x, (double)v.y, (double)v.z}; }     // Systems.swift:428
static inline V3 f3(D3 v) { return V3{(float)v.x, (float)v.y, (float)v.z}; }        // Systems.swift:432
static inline D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
static inline D3 operator/(D3 a, double s) { return D3{a.x / s, a.y / s, a.z / s}; }
static inline D3& operator+=(D3& a, D3 b) { a = a + b; return a; }
static inline D3& operator-=(D3& a, D3 b) { a = a - b; return a; }
static inline double dot(D3 a, D3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline double length(D3 a) { return sqrt(dot(a, a)); }

static inline V4 v4(float x, float y, float z, float w) { return V4{x, y, z, w}; }
static inline V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
// simd_reduce_add(float4) = (x0 + x2) + (x1 + x3)
static inline float dot4(V4 a, V4 b) { return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w); }

static inline M4 m4_identity() {
    return M4{{V4{1, 0, 0, 0}, V4{0, 1, 0, 0}, V4{0, 0, 1, 0}, V4{0, 0, 0, 1}}};
}
// simd_mul(float4x4, float4): ((c0*x + c1*y) + c2*z) + c3*w, no fma.
static inline V4 mul(const M4& m, V4 v) {
    return ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3] * v.w;
}
static inline M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int j = 0; j < 4; ++j) r.c[j] = mul(a, b.c[j]);
    return r;
}

// Math.swift:11-24 matrix4x4_rotation(radians:axis:)
static inline M4 matrix4x4_rotation(float radians, V3 axis) {
    V3 u = normalize(axis);
    float ct = cosf(radians), st = sinf(radians);
    float ci = 1 - ct;
    float x = u.x, y = u.y, z = u.z;
    M4 m;
    m.c[0] = V4{ct + x * x * ci, y * x * ci + z * st, z * x * ci - y * st, 0};
    m.c[1] = V4{x * y * ci - z * st, ct + y * y * ci, z * y * ci + x * st, 0};
    m.c[2] = V4{x * z * ci + y * st, y * z * ci - x * st, ct + z * z * ci, 0};
    m.c[3] = V4{0, 0, 0, 1};
    return m;
}
// Math.swift:26-33
static inline M4 matrix4x4_translation(float tx, float ty, float tz) {
    M4 m = m4_identity();
    m.c[3] = V4{tx, ty, tz, 1};
    return m;
}
// Math.swift:48-50
// Swift's Float.pi is rounded TOWARD ZERO: 0x1.921fb4p+1 (one ulp below (float)M_PI).
static const float SWIFT_FLOAT_PI = 0x1.921fb4p+1f;
static inline float radians_from_degrees(float deg) { return (deg / 180.0f) * SWIFT_FLOAT_PI; }
// Skeleton.swift:212-217 rotationXYZDegrees: Rz * (Ry * Rx)
static inline M4 rotationXYZDegrees(V3 deg) {
    M4 rx = matrix4x4_rotation(radians_from_degrees(deg.x), V3{1, 0, 0});
    M4 ry = matrix4x4_rotation(radians_from_degrees(deg.y), V3{0, 1, 0});
    M4 rz = matrix4x4_rotation(radians_from_degrees(deg.z), V3{0, 0, 1});
    return mul(rz, mul(ry, rx));
}

// simd_inverse(float4x4): adjugate / determinant (SDK algorithm not published;
// used at load time only, Skeleton.swift:155-156).
static inline M4 inverse(const M4& mm) {
    const float* m = &mm.c[0].x;
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
    float* o = &r.c[0].x;
    for (int i = 0; i < 16; ++i) o[i] = inv[i] * id;
    return r;
}

// ---- quaternions (Apple <simd/quaternion.h> algorithms restated) ----
static inline Q4 q4(float x, float y, float z, float w) { return Q4{x, y, z, w}; }
static inline V3 q_imag(Q4 q) { return V3{q.x, q.y, q.z}; }
// simd_quatf(angle:axis:) = (sin(angle/2) * axis, cos(angle/2))
static inline Q4 quat_angle_axis(float angle, V3 axis) {
    float h = angle / 2;
    float s = sinf(h), c = cosf(h);
    return Q4{s * axis.x, s * axis.y, s * axis.z, c};
}
// simd_quaternion(matrix) — trace / largest-diagonal branches
static inline Q4 quat_from_matrix(const M4& m) {
    const V4* mat = m.c;
    float m00 = mat[0].x, m01 = mat[0].y, m02 = mat[0].z;
    float m10 = mat[1].x, m11 = mat[1].y, m12 = mat[1].z;
    float m20 = mat[2].x, m21 = mat[2].y, m22 = mat[2].z;
    float trace = m00 + m11 + m22;
    if (trace >= 0.0f) {
        float r = 2 * sqrtf(1 + trace);
        float rinv = 1.0f / r;
        return Q4{rinv * (m12 - m21), rinv * (m20 - m02), rinv * (m01 - m10), r / 4};
    } else if (m00 >= m11 && m00 >= m22) {
        float r = 2 * sqrtf(1 - m11 - m22 + m00);
        float rinv = 1.0f / r;
        return Q4{r / 4, rinv * (m01 + m10), rinv * (m02 + m20), rinv * (m12 - m21)};
    } else if (m11 >= m22) {
        float r = 2 * sqrtf(1 - m00 - m22 + m11);
        float rinv = 1.0f / r;
        return Q4{rinv * (m01 + m10), r / 4, rinv * (m12 + m21), rinv * (m20 - m02)};
    } else {
        float r = 2 * sqrtf(1 - m00 - m11 + m22);
        float rinv = 1.0f / r;
        return Q4{rinv * (m02 + m20), rinv * (m12 + m21), r / 4, rinv * (m01 - m10)};
    }
}
// matrix_float4x4(simd_quatf)
static inline M4 matrix_from_quat(Q4 v) {
    M4 r;
    r.c[0] = V4{1 - 2 * (v.y * v.y + v.z * v.z), 2 * (v.x * v.y + v.z * v.w), 2 * (v.x * v.z - v.y * v.w), 0};
    r.c[1] = V4{2 * (v.x * v.y - v.z * v.w), 1 - 2 * (v.z * v.z + v.x * v.x), 2 * (v.y * v.z + v.x * v.w), 0};
    r.c[2] = V4{2 * (v.z * v.x + v.y * v.w), 2 * (v.y * v.z - v.x * v.w), 1 - 2 * (v.y * v.y + v.x * v.x), 0};
    r.c[3] = V4{0, 0, 0, 1};
    return r;
}
static inline float q_length_squared(Q4 q) { return dot4(V4{q.x, q.y, q.z, q.w}, V4{q.x, q.y, q.z, q.w}); }
// simd_inverse(q) = conjugate(q) * recip(length_squared(q))
static inline Q4 q_inverse(Q4 q) {
    float r = 1.0f / q_length_squared(q);
    return Q4{-q.x * r, -q.y * r, -q.z * r, q.w * r};
}
// simd_mul(p, q): Hamilton product in the SDK's shuffle order
static inline Q4 q_mul(Q4 p, Q4 q) {
    V4 a = V4{q.w, -q.z, q.y, -q.x} * p.x + V4{q.z, q.w, -q.x, -q.y} * p.y;
    V4 b = V4{-q.y, q.x, q.w, -q.z} * p.z + V4{q.x, q.y, q.z, q.w} * p.w;
    V4 r = a + b;
    return Q4{r.x, r.y, r.z, r.w};
}
static inline Q4 q_normalize(Q4 q) {
    float r = 1.0f / sqrtf(q_length_squared(q));
    return Q4{q.x * r, q.y * r, q.z * r, q.w * r};
}
// simd_act(q, v): t = 2*cross(imag,v); v + real*t + cross(imag,t)
static inline V3 q_act(Q4 q, V3 v) {
    V3 t = 2.0f * cross(q_imag(q), v);
    return (v + t * q.w) + cross(q_imag(q), t);
}
static inline float simd_sinc(float x) { return x == 0 ? 1.0f : sinf(x) / x; }
static inline Q4 slerp_internal(Q4 q0, Q4 q1, float t) {
    float s = 1 - t;
    V4 d = V4{q0.x - q1.x, q0.y - q1.y, q0.z - q1.z, q0.w - q1.w};
    V4 u = V4{q0.x + q1.x, q0.y + q1.y, q0.z + q1.z, q0.w + q1.w};
    float a = 2 * atan2f(sqrtf(dot4(d, d)), sqrtf(dot4(u, u)));
    float r = 1.0f / simd_sinc(a);
    float k0 = simd_sinc(s * a) * r * s;
    float k1 = simd_sinc(t * a) * r * t;
    Q4 q = Q4{k0 * q0.x + k1 * q1.x, k0 * q0.y + k1 * q1.y, k0 * q0.z + k1 * q1.z, k0 * q0.w + k1 * q1.w};
    return q_normalize(q);
}
// simd_slerp: shortest arc
static inline Q4 q_slerp(Q4 q0, Q4 q1, float t) {
    float d = dot4(V4{q0.x, q0.y, q0.z, q0.w}, V4{q1.x, q1.y, q1.z, q1.w});
    if (d >= 0) return slerp_internal(q0, q1, t);
    return slerp_internal(q0, Q4{-q1.x, -q1.y, -q1.z, -q1.w}, t);
}

} // namespace sgeo
