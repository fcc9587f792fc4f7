// ORACLE — TEST INFRASTRUCTURE ONLY (see sge_oracle_math.h).
//
// CPU restatement of the character move-and-slide step, Game/Systems.swift:
//   PhysicsIntentSystem (controller branch)  :205-250, approachVecD :419-426
//   GravitySystem                            :596-620
//   DepenetrationResolver                    :734-808
//   GroundProbe / GroundSnap / SlopeFriction :826-1021
//   VelocityGate                             :1037-1051
//   AgentSweepSolver + capsuleCapsuleSweep   :1053-1091, :1417-1590
//   DefaultContactCachePolicy / ContactManifoldCache :1102-1205
//   SlideResolver.resolveHit / HitSelector   :1207-1399
//   KinematicMoveStopSystem.fixedUpdate      :1823-1902 (+ helpers :1592-1821)
//   PhysicsWritebackSystem (rotation part)   :2249-2267
//   PlatformCarry.computeDelta               :644-732 (over the step's sge_platform_state list; meshWorldAABB :627-642
//                                            is taken once per platform by the caller, sgeo_mesh_world_aabb)
// Entity iteration order in the reference is Swift Dictionary order
// (World.swift:99-117); here it is ascending character index.
// Parity unpinned (no reference tests); see DESIGN.md.
#include <algorithm>
#include <map>
#include "sge_oracle.h"

namespace sgeo {

static inline V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
static inline void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline D3 ldd3(const double* p) { return D3{p[0], p[1], p[2]}; }
static inline void std3(double* p, D3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// Systems.swift:419-426
static D3 approachVecD(D3 current, D3 target, double maxDelta) {
    D3 delta = target - current;
    double len = length(delta);
    if (len <= maxDelta || len < 0.00001) return target;
    return current + delta / len * maxDelta;
}

// Systems.swift:205-250, entities that own a CharacterControllerComponent
void intent_fixed_update(World& w, int first, int count, float dt) {
    for (int e = first; e < first + count; ++e) {
        sge_body_state& body = w.bodies[e];
        const sge_move_intent& in = w.intents[e];
        if (!(in.flags & SGE_INTENT_PRESENT)) continue;
        if (body.bodyType != SGE_BODY_DYNAMIC && body.bodyType != SGE_BODY_KINEMATIC) continue;
        if (in.flags & SGE_INTENT_DODGE_ACTIVE) {
            body.linearVelocity[0] = (double)in.desiredVelocity[0];
            body.linearVelocity[2] = (double)in.desiredVelocity[2];
        } else {
            D3 target = D3{(double)in.desiredVelocity[0], 0, (double)in.desiredVelocity[2]};
            D3 current = D3{body.linearVelocity[0], 0, body.linearVelocity[2]};
            float accel = length(target) >= length(current) ? in.maxAcceleration : in.maxDeceleration;
            D3 next = approachVecD(current, target, (double)accel * (double)dt);
            body.linearVelocity[0] = next.x;
            body.linearVelocity[2] = next.z;
        }
        if (in.flags & SGE_INTENT_HAS_FACING_YAW) {
            Q4 q = quat_angle_axis(in.desiredFacingYaw, V3{0, 1, 0});
            body.rotation[0] = q.x; body.rotation[1] = q.y; body.rotation[2] = q.z; body.rotation[3] = q.w;
        }
    }
}

// Systems.swift:603-619
void gravity_fixed_update(World& w, int first, int count, float dt, V3 gravity) {
    for (int e = first; e < first + count; ++e) {
        sge_body_state& body = w.bodies[e];
        if (body.bodyType != SGE_BODY_DYNAMIC) continue;
        const sge_controller_state& C = w.controllers[e];
        if ((C.flags & SGE_CTRL_GROUNDED) && (C.flags & SGE_CTRL_GROUNDED_NEAR)) continue;
        D3 v = ldd3(body.linearVelocity);
        v += d3(gravity) * (double)dt;
        std3(body.linearVelocity, v);
    }
}

// Systems.swift:2259-2265 (translation is implied by body.position)
void writeback_fixed_update(World& w, int first, int count) {
    for (int e = first; e < first + count; ++e) {
        sge_body_state& body = w.bodies[e];
        for (int k = 0; k < 4; ++k) body.transformRotation[k] = body.rotation[k];
    }
}

// ---- contact cache: DefaultContactCachePolicy + ContactManifoldCache ----
static void manifoldReset(sge_controller_state& c) { c.manifoldCount = 0; c.manifoldFrames = 0; } // :1163
static void cacheDecay(sge_controller_state& c) { // :1105-1116
    if (c.sideContactFrames > 0) c.sideContactFrames -= 1;
    if (c.manifoldFrames > 0) {
        c.manifoldFrames -= 1;
        if (c.manifoldFrames == 0) {
            manifoldReset(c);
            st3(c.sideContactNormal, V3{0, 0, 0});
        }
    }
}
static bool cachedNormal(const sge_controller_state& c, int triangleIndex, V3& out) { // :1169-1175
    for (int i = 0; i < c.manifoldCount; ++i)
        if (c.manifoldTriangles[i] == triangleIndex) { out = ld3(c.manifoldNormals[i]); return true; }
    return false;
}
static void manifoldUpdate(sge_controller_state& c, int triangleIndex, V3 normal) { // :1177-1204
    V3 n = normal;
    if (length_squared(n) < 1e-8f) return;
    c.manifoldFrames = 8;
    for (int i = 0; i < c.manifoldCount; ++i) {
        if (c.manifoldTriangles[i] == triangleIndex) {
            V3 cached = ld3(c.manifoldNormals[i]);
            if (dot(cached, n) < 0) n = -n;
            const float blend = 0.25f;
            V3 combined = normalize(cached * (1 - blend) + n * blend);
            st3(c.manifoldNormals[i], combined);
            st3(c.sideContactNormal, combined);
            return;
        }
    }
    if (c.manifoldCount >= SGE_MANIFOLD_MAX) c.manifoldCount -= 1; // removeLast
    for (int i = c.manifoldCount; i > 0; --i) {                    // insert at 0
        c.manifoldTriangles[i] = c.manifoldTriangles[i - 1];
        st3(c.manifoldNormals[i], ld3(c.manifoldNormals[i - 1]));
    }
    c.manifoldTriangles[0] = triangleIndex;
    st3(c.manifoldNormals[0], normalize(n));
    c.manifoldCount += 1;
    st3(c.sideContactNormal, ld3(c.manifoldNormals[0]));
}
static void cacheRecord(sge_controller_state& c, int triangleIndex, V3 normal, bool isSideContact) { // :1122-1133
    manifoldUpdate(c, triangleIndex, normal);
    if (isSideContact) {
        st3(c.sideContactNormal, normalize(normal));
        c.sideContactFrames = 3;
    }
}

// ---- DepenetrationResolver.resolve :734-808 ----
static bool depenetrationResolve(V3& position, D3& velocity, const sge_controller_params& P, sge_controller_state& C,
                                 const CollisionQuery& query, V3& outNormal, bool sideContactCacheOnly) {
    const float radius = P.radius, halfHeight = P.halfHeight, skinWidth = P.skinWidth;
    float slop = fmax_s(skinWidth * 0.5f, 0.001f);
    bool didResolve = false;
    V3 normalSum = V3{0, 0, 0};
    float normalWeight = 0;
    for (int it = 0; it < 4; ++it) {
        CapsuleOverlapHit hits[SGE_MAX_OVERLAP_HITS];
        int n = query.capsuleOverlapAll(position, radius, halfHeight, 8, P.collisionMask, hits);
        if (n == 0) break;
        std::stable_sort(hits, hits + n, [](const CapsuleOverlapHit& a, const CapsuleOverlapHit& b) { return a.depth > b.depth; });
        const CapsuleOverlapHit& deepest = hits[0];
        bool sideContact = deepest.normal.y < P.minGroundDot;
        int useCount = sideContact ? 1 : std::min(2, n);
        float maxDepth = deepest.depth;
        V3 frameNormal = V3{0, 0, 0};
        for (int k = 0; k < useCount; ++k) {
            const CapsuleOverlapHit& hit = hits[k];
            maxDepth = fmax_s(maxDepth, hit.depth);
            V3 nn = hit.normal;
            V3 cached;
            if (cachedNormal(C, hit.triangleIndex, cached)) nn = cached;
            frameNormal += nn * hit.depth;
            const bool isSide = hit.normal.y < P.minGroundDot;
            // contactCachePolicy.record: DefaultContactCachePolicy :1122-1133, or SideContactOnlyCachePolicy :1146-1156
            // (`guard isSideContact else { return }`, then the same update + side-contact memory)
            if (isSide || !sideContactCacheOnly) cacheRecord(C, hit.triangleIndex, nn, isSide);
        }
        float frameNormalLen = length(frameNormal);
        V3 depenNormal = frameNormalLen > 1e-6f ? frameNormal / frameNormalLen : frameNormal;
        float push = sideContact ? fmax_s(maxDepth, 0.0f) : fmax_s(maxDepth + slop, 0.0f);
        if (sideContact) push = fmin_s(push, skinWidth);
        if (push <= 1e-6f) break;
        position += depenNormal * push;
        D3 depenNormalD = d3(depenNormal);
        double vInto = dot(velocity, depenNormalD);
        if (vInto < 0) velocity -= depenNormalD * vInto;
        didResolve = true;
        normalSum += depenNormal * maxDepth;
        normalWeight += maxDepth;
    }
    if (!didResolve) return false;
    if (normalWeight > 1e-6f) outNormal = normalize(normalSum / normalWeight);
    else outNormal = normalize(normalSum);
    return true;
}

// ---- capsule-capsule sweep :1417-1590 ----
struct Interval { float s, e; bool ok; };
static Interval clampInterval(float start, float end) { // :1417
    float s = fmax_s(start, 0.0f), e = fmin_s(end, 1.0f);
    if (e < s) return Interval{0, 0, false};
    return Interval{s, e, true};
}
static Interval intervalGreaterEqual(float y0, float vy, float threshold) { // :1426
    const float eps = 1e-6f;
    if (fabsf(vy) < eps) return y0 >= threshold ? Interval{0, 1, true} : Interval{0, 0, false};
    float t = (threshold - y0) / vy;
    if (vy > 0) return clampInterval(t, 1);
    return clampInterval(0, t);
}
static Interval intervalLessEqual(float y0, float vy, float threshold) { // :1438
    const float eps = 1e-6f;
    if (fabsf(vy) < eps) return y0 <= threshold ? Interval{0, 1, true} : Interval{0, 0, false};
    float t = (threshold - y0) / vy;
    if (vy > 0) return clampInterval(0, t);
    return clampInterval(t, 1);
}
static bool earliestRoot(float A, float B, float C, float tMin, float tMax, float& out) { // :1450
    const float eps = 1e-6f;
    if (fabsf(A) < eps) {
        if (fabsf(B) < eps) { if (C <= 0) { out = tMin; return true; } return false; }
        float t = -C / B;
        if (t >= tMin && t <= tMax) { out = t; return true; }
        return false;
    }
    float disc = B * B - 4 * A * C;
    if (disc < 0) return false;
    float sqrtD = sqrtf(disc);
    float inv2A = 1 / (2 * A);
    float t0 = (-B - sqrtD) * inv2A;
    float t1 = (-B + sqrtD) * inv2A;
    float enter = fmin_s(t0, t1), exit = fmax_s(t0, t1);
    float s = fmax_s(enter, tMin), e = fmin_s(exit, tMax);
    if (e >= s) { out = s; return true; }
    return false;
}
static float capsuleCapsuleSeparationY(float yRel, float halfHeightSum) { // :1474
    if (yRel > halfHeightSum) return yRel - halfHeightSum;
    if (yRel < -halfHeightSum) return yRel + halfHeightSum;
    return 0;
}
static V3 capsuleCapsuleHitNormal(V3 rel, float halfHeightSum) { // :1484
    float sepY = capsuleCapsuleSeparationY(rel.y, halfHeightSum);
    V3 sep = V3{rel.x, sepY, rel.z};
    float lenSq = length_squared(sep);
    if (lenSq > 1e-8f) return sep / sqrtf(lenSq);
    V3 lateral = V3{rel.x, 0, rel.z};
    float lateralLenSq = length_squared(lateral);
    if (lateralLenSq > 1e-8f) return lateral / sqrtf(lateralLenSq);
    return V3{1, 0, 0};
}
static bool capsuleCapsuleOverlap(V3 rel, float radiusSum, float halfHeightSum) { // :1499
    float sepY = capsuleCapsuleSeparationY(rel.y, halfHeightSum);
    float distSq = rel.x * rel.x + rel.z * rel.z + sepY * sepY;
    return distSq <= radiusSum * radiusSum;
}
struct CapsuleCapsuleHit { float toi; V3 normal; int other; };
static bool capsuleCapsuleSweep(V3 from, V3 delta, float radius, float halfHeight, int other, V3 otherPos,
                                V3 otherDelta, float otherRadius, float otherHalfHeight, CapsuleCapsuleHit& out) { // :1505
    V3 relStart = from - otherPos;
    V3 relDelta = delta - otherDelta;
    float rSum = radius + otherRadius, hSum = halfHeight + otherHalfHeight;
    float relLen = length(relDelta), moveLen = length(delta);
    if (relLen < 1e-6f) {
        if (capsuleCapsuleOverlap(relStart, rSum, hSum)) {
            out = CapsuleCapsuleHit{0, capsuleCapsuleHitNormal(relStart, hSum), other};
            return true;
        }
        return false;
    }
    float y0 = relStart.y, vy = relDelta.y, vx = relDelta.x, vz = relDelta.z, r0x = relStart.x, r0z = relStart.z;
    bool have = false;
    float bestT = 0, t;
    Interval upper = intervalGreaterEqual(y0, vy, hSum);
    if (upper.ok) {
        float A = vx * vx + vz * vz + vy * vy;
        float B = 2 * (r0x * vx + r0z * vz + (y0 - hSum) * vy);
        float C = r0x * r0x + r0z * r0z + (y0 - hSum) * (y0 - hSum) - rSum * rSum;
        if (earliestRoot(A, B, C, upper.s, upper.e, t)) { bestT = t; have = true; }
    }
    Interval lower = intervalLessEqual(y0, vy, -hSum);
    if (lower.ok) {
        float A = vx * vx + vz * vz + vy * vy;
        float B = 2 * (r0x * vx + r0z * vz + (y0 + hSum) * vy);
        float C = r0x * r0x + r0z * r0z + (y0 + hSum) * (y0 + hSum) - rSum * rSum;
        if (earliestRoot(A, B, C, lower.s, lower.e, t)) { if (!have || t < bestT) { bestT = t; have = true; } }
    }
    const float eps = 1e-6f;
    if (fabsf(vy) < eps) {
        if (fabsf(y0) <= hSum) {
            float A = vx * vx + vz * vz;
            float B = 2 * (r0x * vx + r0z * vz);
            float C = r0x * r0x + r0z * r0z - rSum * rSum;
            if (earliestRoot(A, B, C, 0, 1, t)) { if (!have || t < bestT) { bestT = t; have = true; } }
        }
    } else {
        float t1 = (hSum - y0) / vy, t2 = (-hSum - y0) / vy;
        Interval ov = clampInterval(fmin_s(t1, t2), fmax_s(t1, t2));
        if (ov.ok) {
            float A = vx * vx + vz * vz;
            float B = 2 * (r0x * vx + r0z * vz);
            float C = r0x * r0x + r0z * r0z - rSum * rSum;
            if (earliestRoot(A, B, C, ov.s, ov.e, t)) { if (!have || t < bestT) { bestT = t; have = true; } }
        }
    }
    if (!have) return false;
    V3 relAtHit = relStart + relDelta * bestT;
    out = CapsuleCapsuleHit{bestT * moveLen, capsuleCapsuleHitNormal(relAtHit, hSum), other};
    return true;
}

// AgentSweepSolver.bestHit :1053-1091 (all-pairs over the start-of-step snapshot)
static bool agentBestHit(V3 position, V3 remaining, float remainingLen, float baseMoveLen, float dt, int selfEntity,
                         bool selfSolid, float selfRadius, float halfHeight,
                         const std::vector<AgentSweepState>& agents, CapsuleCapsuleHit& best) {
    if (!selfSolid) return false;
    bool have = false;
    float timeScale = baseMoveLen > 1e-6f ? fmin_s(remainingLen / baseMoveLen, 1.0f) : 1.0f;
    float segmentDt = dt * timeScale;
    for (const AgentSweepState& other : agents) {
        if (other.entity == selfEntity) continue;
        V3 otherDelta = other.velocity * segmentDt;
        CapsuleCapsuleHit hit;
        if (capsuleCapsuleSweep(position, remaining, selfRadius, halfHeight, other.entity, other.position, otherDelta,
                                other.radius, other.halfHeight, hit)) {
            if (!have || hit.toi < best.toi) { best = hit; have = true; }
        }
    }
    return have;
}

// ---- SlideResolver.resolveHit :1229-1375 ----
struct SlideOptions { bool allowHorizontalGroundPass, adjustVelocity, useGroundSnapSkinForStatic, allowTriangleNormalGroundLike; }; // :1208-1222
static const SlideOptions kKinematicMove{false, true, true, true};
static const SlideOptions kAgentSeparation{true, false, false, false};
struct SlideHit { bool isStatic; CapsuleCastHit s; CapsuleCapsuleHit a; };
static bool resolveHit(V3& remaining, float len, const SlideHit& hit, const sge_controller_params& P,
                       const sge_controller_state& C, bool wasGrounded, bool wasGroundedNear, D3& velocity,
                       V3& position, bool hasCachedSideNormal, V3 cachedSideNormal, const SlideOptions& options = kKinematicMove) {
    if (options.allowHorizontalGroundPass && hit.isStatic && fabsf(remaining.y) < 1e-5f && hit.s.normal.y >= P.minGroundDot) { // :1240-1247
        position += remaining;
        remaining = V3{0, 0, 0};
        return true;
    }
    float contactSkin, hitToi;
    V3 slideNormal, hitTriNormal = V3{0, 0, 0};
    bool hitIsStatic = false, hitIsGroundLike = false;
    if (hit.isStatic) {
        hitToi = hit.s.toi;
        slideNormal = hit.s.normal;
        hitIsGroundLike = hit.s.triangleNormal.y >= P.minGroundDot;
        contactSkin = (options.useGroundSnapSkinForStatic && hitIsGroundLike) ? P.groundSnapSkin : P.skinWidth;
        hitTriNormal = hit.s.triangleNormal;
        hitIsStatic = true;
    } else {
        hitToi = hit.a.toi;
        slideNormal = hit.a.normal;
        contactSkin = 0;
    }
    if (hitIsStatic && slideNormal.y < P.minGroundDot && C.sideContactFrames > 0) {
        if (hasCachedSideNormal) {
            V3 cachedN = cachedSideNormal;
            if (dot(cachedN, slideNormal) < 0) cachedN = -cachedN;
            slideNormal = cachedN;
        } else {
            V3 cached = ld3(C.sideContactNormal);
            float cachedLen = length_squared(cached);
            if (cachedLen > 1e-6f) {
                V3 cachedN = cached / sqrtf(cachedLen);
                float dotC = dot(cachedN, slideNormal);
                if (fabsf(dotC) > 0.5f) slideNormal = dotC >= 0 ? cachedN : -cachedN;
            }
        }
    }
    if (slideNormal.y < P.minGroundDot) {
        if (hitIsStatic && hitIsGroundLike && options.allowTriangleNormalGroundLike) slideNormal = hitTriNormal;
        if (slideNormal.y < P.minGroundDot) {
            slideNormal.y = 0;
            float nLen = length(slideNormal);
            if (nLen > 1e-5f) {
                slideNormal = slideNormal / nLen;
            } else {
                position += remaining;
                remaining = V3{0, 0, 0};
                return true;
            }
        }
    }
    float into = dot(remaining, slideNormal);
    float intoEps = 1e-4f * len;
    float effectiveSkin;
    if (hitToi <= contactSkin && into < -intoEps) effectiveSkin = fmin_s(contactSkin, hitToi * 0.5f);
    else effectiveSkin = contactSkin;
    float stickyThreshold = contactSkin * 0.1f;
    if (hitToi <= stickyThreshold && into < -intoEps) {
        remaining -= slideNormal * into;
        return false;
    }
    if (into >= -intoEps) {
        if (wasGroundedNear && hitIsStatic && !hitIsGroundLike && remaining.y < 0) remaining.y = 0;
        position += remaining;
        remaining = V3{0, 0, 0};
        return true;
    }
    if (hitToi <= effectiveSkin && fabsf(into) <= intoEps) {
        position += remaining;
        remaining = V3{0, 0, 0};
        return true;
    }
    if (into >= 0) {
        position += remaining;
        remaining = V3{0, 0, 0};
        return true;
    }
    float rawMoveDist = fmax_s(hitToi - effectiveSkin, 0.0f);
    float moveDist = rawMoveDist;
    if (slideNormal.y >= P.minGroundDot && remaining.y < 0 && moveDist > P.groundSweepMaxStep) moveDist = P.groundSweepMaxStep;
    V3 dir = remaining / len;
    position += dir * moveDist;
    V3 leftover = remaining - dir * moveDist;
    leftover -= slideNormal * dot(leftover, slideNormal);
    if (wasGrounded && wasGroundedNear && leftover.y < 0) leftover.y = 0;
    float residual = dot(leftover, slideNormal);
    if (fabsf(residual) < 1e-5f) leftover -= slideNormal * residual;
    if (length_squared(leftover) < 1e-8f) {
        remaining = V3{0, 0, 0};
        return true;
    }
    remaining = leftover;
    if (options.adjustVelocity) {
        D3 snD = d3(slideNormal);
        double vInto = dot(velocity, snD);
        if (vInto < 0) velocity -= snD * vInto;
    }
    return false;
}

// ---- resolveKinematicSweep :1658-1765 ----
static void resolveKinematicSweep(int entity, V3& position, V3& remaining, D3& velocity, const sge_controller_params& P,
                                  sge_controller_state& C, bool wasGrounded, bool wasGroundedNear, bool selfSolid,
                                  float selfRadius, const std::vector<AgentSweepState>* agents,
                                  const CollisionQuery& query, float dt) {
    V3 baseMove = f3(velocity) * dt;
    float baseMoveLen = length(baseMove);
    bool haveLast = false;
    V3 lastSlideNormal = V3{0, 0, 0};
    for (int it = 0; it < P.maxSlideIterations; ++it) {
        float len = length(remaining);
        if (len < 1e-6f) break;
        CapsuleCastHit sHit = CapsuleCastHit{};
        bool haveStatic = query.capsuleCastCombined(position, remaining, P.radius, P.halfHeight, true, false, 0, P.collisionMask, sHit);
        if (haveStatic && sHit.normal.y < P.minGroundDot && C.sideContactFrames > 0) {
            V3 cached;
            if (cachedNormal(C, sHit.triangleIndex, cached)) {
                if (dot(cached, sHit.normal) < 0) cached = -cached;
                sHit.normal = cached;
            }
        }
        CapsuleCapsuleHit aHit = CapsuleCapsuleHit{0, V3{0, 0, 0}, -1};
        bool haveAgent = agents ? agentBestHit(position, remaining, len, baseMoveLen, dt, entity, selfSolid, selfRadius,
                                               P.halfHeight, *agents, aHit)
                                : false;
        if (haveStatic || haveAgent) {
            SlideHit hit; // HitSelector.selectBestHit :1378-1399
            if (haveStatic && haveAgent) {
                float staticSkin = sHit.normal.y >= P.minGroundDot ? P.groundSnapSkin : P.skinWidth;
                float staticStop = fmax_s(sHit.toi - staticSkin, 0.0f);
                float agentStop = fmax_s(aHit.toi, 0.0f);
                hit.isStatic = staticStop <= agentStop;
            } else {
                hit.isStatic = haveStatic;
            }
            hit.s = sHit; hit.a = aHit;
            V3 hitNormal = hit.isStatic ? sHit.normal : aHit.normal;
            bool hasCachedSide = false;
            V3 cachedSide = V3{0, 0, 0};
            if (hit.isStatic && sHit.normal.y < P.minGroundDot && C.sideContactFrames > 0)
                hasCachedSide = cachedNormal(C, sHit.triangleIndex, cachedSide);
            bool shouldBreak = resolveHit(remaining, len, hit, P, C, wasGrounded, wasGroundedNear, velocity, position,
                                          hasCachedSide, cachedSide);
            if (hit.isStatic && sHit.normal.y < P.minGroundDot) cacheRecord(C, sHit.triangleIndex, sHit.normal, true);
            if (haveLast) {
                float dotN = dot(lastSlideNormal, hitNormal);
                if (fabsf(dotN) < 0.98f) {
                    V3 axis = cross(lastSlideNormal, hitNormal);
                    float axisLen = length(axis);
                    if (axisLen > 1e-5f) {
                        V3 axisN = axis / axisLen;
                        remaining = axisN * dot(remaining, axisN);
                    }
                }
            }
            lastSlideNormal = hitNormal;
            haveLast = true;
            if (shouldBreak) break;
        } else {
            position += remaining;
            remaining = V3{0, 0, 0};
            break;
        }
    }
}

// ---- GroundProbe.resolve :826-943 ----
struct GroundContactState { bool grounded, groundedNear; V3 normal; sge_surface_material material; int triangleIndex; float distance; };
struct GroundProbeResult { GroundContactState state; bool canSnap, nearGround, hasHit; CapsuleCastHit hit; };

static GroundProbeResult groundProbe(V3 position, D3 velocity, const sge_controller_params& P, const CollisionQuery& query,
                                     bool wasGroundedNear, V3 prevNormal) {
    GroundProbeResult R;
    R.state = GroundContactState{false, false, V3{0, 1, 0}, sge_surface_material{0.8f, 0.6f, 0}, -1, 3.40282347e+38f};
    R.canSnap = false; R.nearGround = false; R.hasHit = false;
    V3 down = V3{0, -1, 0};
    V3 snapDelta = down * P.snapDistance;
    CapsuleCastHit centerHit;
    bool haveCenter = false;
    if (P.snapDistance > 0)
        haveCenter = query.capsuleCastCombined(position, snapDelta, P.radius, P.halfHeight, false, true, P.minGroundDot, P.collisionMask, centerHit);
    if (P.fallProbeDistance > 0) {
        V3 fallDelta = down * P.fallProbeDistance;
        CapsuleCastHit fallHit;
        if (query.capsuleCastCombined(position, fallDelta, P.radius, P.halfHeight, false, true, P.minGroundDot, P.collisionMask, fallHit))
            R.state.distance = fallHit.toi;
    }
    if (!haveCenter || !(centerHit.toi <= P.snapDistance)) return R;

    float baseCenterY = position.y - P.halfHeight;
    float bottomY = baseCenterY - P.radius;
    float groundTol = fmax_s(P.skinWidth, P.groundSnapSkin);
    bool validGroundPoint = centerHit.position.y <= bottomY + groundTol;
    float groundNearThreshold = fmax_s(P.groundSnapSkin, P.skinWidth);
    bool nearGround = centerHit.toi <= groundNearThreshold;
    R.state.groundedNear = nearGround;
    R.state.distance = centerHit.toi;
    bool groundGateVel = velocity.y <= 0;
    double vInto = dot(velocity, d3(centerHit.normal));
    bool groundGateSpeed = vInto >= -(double)P.groundSnapMaxSpeed;
    bool groundGateToi = centerHit.toi <= P.groundSnapMaxToi;
    bool canSnap = validGroundPoint && groundGateVel && (nearGround || groundGateSpeed || groundGateToi);
    if (wasGroundedNear && centerHit.toi <= P.snapDistance) canSnap = validGroundPoint;

    if (validGroundPoint && (nearGround || canSnap)) {
        R.state.grounded = true;
        R.state.material = centerHit.material;
        R.state.triangleIndex = centerHit.triangleIndex;
        V3 normalSum = centerHit.triangleNormal;
        const float flatDot = 0.98f;
        if (centerHit.triangleNormal.y < flatDot && (wasGroundedNear || nearGround)) {
            float offset = P.radius * 0.6f;
            float sx[4] = {offset, -offset, 0, 0}, sz[4] = {0, 0, offset, -offset};
            float combineTol = fmax_s(fmax_s(P.groundSnapSkin, P.skinWidth), 0.05f);
            for (int k = 0; k < 4; ++k) {
                V3 samplePos = position + V3{sx[k], 0, sz[k]};
                CapsuleCastHit hit;
                if (query.capsuleCastCombined(samplePos, snapDelta, P.radius, P.halfHeight, false, true, P.minGroundDot, P.collisionMask, hit) &&
                    hit.toi <= centerHit.toi + combineTol) {
                    if (dot(hit.triangleNormal, centerHit.triangleNormal) > 0.98f) normalSum += hit.triangleNormal;
                }
            }
        }
        float nLen = length(normalSum);
        R.state.normal = nLen > 1e-6f ? normalSum / nLen : centerHit.triangleNormal;
    }
    if (R.state.grounded && wasGroundedNear) {
        float dotN = dot(prevNormal, R.state.normal);
        if (dotN > 0.9f) {
            const float blend = 0.2f;
            R.state.normal = normalize(prevNormal * (1 - blend) + R.state.normal * blend);
        }
    }
    if (R.state.grounded && R.state.material.flattenGround) R.state.normal = V3{0, 1, 0};
    R.canSnap = canSnap; R.nearGround = nearGround; R.hasHit = true; R.hit = centerHit;
    return R;
}

// GroundSnap.apply :945-963
static void groundSnap(V3& position, D3& velocity, const sge_controller_params& P, const GroundProbeResult& R) {
    if (!R.canSnap || !R.hasHit) return;
    V3 down = V3{0, -1, 0};
    float rawMove = fmax_s(R.hit.toi - P.groundSnapSkin, 0.0f);
    float moveDist = rawMove;
    if (R.nearGround && moveDist > P.groundSnapMaxStep) moveDist = P.groundSnapMaxStep;
    position += down * moveDist;
    D3 nD = d3(R.hit.normal);
    double vIntoSnap = dot(velocity, nD);
    if (vIntoSnap < 0) velocity -= nD * vIntoSnap;
}

// SlopeFriction.apply :965-1021
static void slopeFriction(D3& velocity, sge_controller_state& C, V3 gravity, float dt, const GroundContactState& st) {
    if (!st.grounded) { C.flags &= ~SGE_CTRL_GROUND_SLIDING; return; }
    V3 normal = normalize(st.normal);
    if (normal.y > 0.98f) {
        C.groundTransitionFrames = 0;
        C.flags &= ~SGE_CTRL_GROUND_SLIDING;
        return;
    }
    if (C.groundTransitionFrames > 0) {
        C.groundTransitionFrames -= 1;
        C.flags &= ~SGE_CTRL_GROUND_SLIDING;
        return;
    }
    float gN = dot(gravity, normal);
    V3 gTan = gravity - normal * gN;
    float gTanLen = length(gTan);
    const float slopeAccelEps = 0.5f;
    if (gTanLen > slopeAccelEps) {
        float gNMag = fabsf(gN);
        V3 gTanDir = gTan / gTanLen;
        D3 gTanDirD = d3(gTanDir), normalD = d3(normal);
        float stickLimit = st.material.muS * gNMag;
        bool enterSlide = gTanLen > stickLimit * 1.05f;
        bool exitSlide = gTanLen < stickLimit * 0.9f;
        bool sliding = (C.flags & SGE_CTRL_GROUND_SLIDING) != 0;
        if (sliding) { if (exitSlide) sliding = false; }
        else if (enterSlide) sliding = true;
        if (sliding) C.flags |= SGE_CTRL_GROUND_SLIDING; else C.flags &= ~SGE_CTRL_GROUND_SLIDING;
        if (!sliding && gTanLen <= stickLimit) {
            D3 v = velocity;
            D3 vTan = v - normalD * dot(v, normalD);
            double downhillSpeed = dot(vTan, gTanDirD);
            if (downhillSpeed > 0) velocity -= gTanDirD * downhillSpeed;
        } else {
            float slideAccelMag = fmax_s(gTanLen - st.material.muK * gNMag, 0.0f);
            if (slideAccelMag > 0) velocity += gTanDirD * (double)slideAccelMag * (double)dt;
        }
    }
}

// collectAgentStates :1592-1611 — the snapshot taken before the per-entity loop
void collect_agent_states(const World& w, std::vector<AgentSweepState>& agents, int& selfOffset) {
    agents.clear();
    selfOffset = 0;
    if (w.agentsImported) {
        selfOffset = w.agentSelfOffset;
        for (size_t i = 0; i < w.importedAgents.size(); ++i) {
            const sge_agent_state& a = w.importedAgents[i];
            if (a.radius < 0) continue;
            agents.push_back(AgentSweepState{(int)i, ld3(a.position), ld3(a.velocity), a.radius, a.halfHeight});
        }
    } else {
        for (int e = 0; e < (int)w.bodies.size(); ++e) {
            const sge_controller_params& P = w.params[e];
            if (!(P.agentFlags & SGE_AGENT_PRESENT) || !(P.agentFlags & SGE_AGENT_SOLID)) continue;
            float radius = (P.agentFlags & SGE_AGENT_RADIUS_OVERRIDE) ? P.agentRadiusOverride : P.radius;
            agents.push_back(AgentSweepState{e, f3(ldd3(w.bodies[e].position)), f3(ldd3(w.bodies[e].linearVelocity)), radius, P.halfHeight});
        }
    }
}

// PlatformCarry.computeDelta :644-732 over the step's platform list (meshWorldAABB is taken per platform, not per
// character: it depends on the platform alone)
static V3 platformCarryDelta(V3 position, const sge_controller_params& P, const std::vector<sge_platform_state>& platforms) {
    if (platforms.empty()) return V3{0, 0, 0};
    float capsuleHalf = P.halfHeight + P.radius;
    float baseY = position.y - capsuleHalf;
    V3 capMin = V3{position.x - P.radius, position.y - capsuleHalf, position.z - P.radius};
    V3 capMax = V3{position.x + P.radius, position.y + capsuleHalf, position.z + P.radius};
    float sideTol = fmax_s(P.skinWidth, P.groundSnapSkin);
    V3 bestCarry = V3{0, 0, 0}, pushDelta = V3{0, 0, 0};
    for (const sge_platform_state& pf : platforms) {
        if (!pf.kinematic) continue;
        V3 pDelta = V3{pf.delta[0], pf.delta[1], pf.delta[2]};
        if (length_squared(pDelta) < 1e-8f) continue;
        if (!pf.hasAABB) continue;
        V3 amin = V3{pf.aabbMin[0], pf.aabbMin[1], pf.aabbMin[2]}, amax = V3{pf.aabbMax[0], pf.aabbMax[1], pf.aabbMax[2]};
        V3 tol = V3{sideTol, sideTol, sideTol};
        V3 expandedMin = amin - tol, expandedMax = amax + tol;
        bool overlap = capMin.x <= expandedMax.x && capMax.x >= expandedMin.x && capMin.y <= expandedMax.y &&
                       capMax.y >= expandedMin.y && capMin.z <= expandedMax.z && capMax.z >= expandedMin.z;
        if (!overlap) continue;
        bool withinXZ = position.x >= amin.x - P.radius && position.x <= amax.x + P.radius &&
                        position.z >= amin.z - P.radius && position.z <= amax.z + P.radius;
        float topY = amax.y;
        float topTol = P.snapDistance + fmax_s(P.skinWidth, P.groundSnapSkin) + 0.05f;
        bool onTop = withinXZ && baseY >= topY - topTol && baseY <= topY + topTol;
        if (onTop) {
            if (length_squared(pDelta) > length_squared(bestCarry)) bestCarry = pDelta;
        } else {
            float yMin = amin.y - capsuleHalf, yMax = amax.y + capsuleHalf;
            if (position.y >= yMin && position.y <= yMax) {
                bool outsideX = position.x < amin.x - P.radius || position.x > amax.x + P.radius;
                bool outsideZ = position.z < amin.z - P.radius || position.z > amax.z + P.radius;
                if (!outsideX && !outsideZ) continue;
                float cx = fmax_s(amin.x, fmin_s(position.x, amax.x));
                float cz = fmax_s(amin.z, fmin_s(position.z, amax.z));
                float dx = position.x - cx, dz = position.z - cz;
                float sideDistSq = dx * dx + dz * dz;
                float sidePushTol = P.radius + sideTol;
                if (sideDistSq <= sidePushTol * sidePushTol) {
                    float dirLen = sqrtf(fmax_s(sideDistSq, 0.0f));
                    if (dirLen > 1e-5f) {
                        V3 dir = V3{dx / dirLen, 0, dz / dirLen};
                        float moveToward = dot(V3{pDelta.x, 0, pDelta.z}, dir);
                        if (moveToward > 0) pushDelta += V3{pDelta.x, 0, pDelta.z};
                    }
                }
            }
        }
    }
    if (length_squared(bestCarry) > 1e-8f) return bestCarry;
    if (length_squared(pushDelta) > 1e-8f) return pushDelta;
    return V3{0, 0, 0};
}

// VelocityGate.apply :1037-1051
static V3 velocityGate(D3& velocity, bool wasGrounded, bool wasGroundedNear, float dt) {
    if (wasGrounded && wasGroundedNear && velocity.y < 0) velocity.y = 0;
    D3 remD = velocity * (double)dt;
    if (wasGrounded && wasGroundedNear && remD.y < 0) remD.y = 0;
    return f3(remD);
}

// KinematicMoveStopSystem.fixedUpdate :1823-1902
void kinematic_move_fixed_update(World& w, int first, int count, float dt, V3 gravity,
                                 const std::vector<AgentSweepState>* agentsPtr, int selfOffset) {
    const CollisionQuery& query = w.query;
    const bool useAgents = agentsPtr != nullptr;
    for (int e = first; e < first + count; ++e) {
        sge_body_state& body = w.bodies[e];
        if (body.bodyType == SGE_BODY_STATIC) continue;
        const sge_controller_params& P = w.params[e];
        sge_controller_state& C = w.controllers[e];
        D3 velocity = ldd3(body.linearVelocity);
        V3 position = f3(ldd3(body.position));
        cacheDecay(C);
        bool selfSolid = (P.agentFlags & SGE_AGENT_PRESENT) && (P.agentFlags & SGE_AGENT_SOLID);
        float selfRadius = ((P.agentFlags & SGE_AGENT_PRESENT) && (P.agentFlags & SGE_AGENT_RADIUS_OVERRIDE)) ? P.agentRadiusOverride : P.radius;
        // applyPlatformDelta :1619-1633
        {
            V3 platformDelta = platformCarryDelta(position, P, w.platforms);
            if (length_squared(platformDelta) > 1e-8f) position += platformDelta;
        }
        bool wasGrounded = (C.flags & SGE_CTRL_GROUNDED) != 0;
        bool wasGroundedNear = (C.flags & SGE_CTRL_GROUNDED_NEAR) != 0;
        V3 remaining = velocityGate(velocity, wasGrounded, wasGroundedNear, dt);
        // applyPreSweepDepenetration :1635-1656
        V3 depenNormal;
        if (depenetrationResolve(position, velocity, P, C, query, depenNormal, w.sideContactCacheOnly)) {
            float into = dot(remaining, depenNormal);
            if (into < 0) remaining -= depenNormal * into;
        }
        resolveKinematicSweep(selfOffset + e, position, remaining, velocity, P, C, wasGrounded, wasGroundedNear, selfSolid,
                              selfRadius, useAgents ? agentsPtr : nullptr, query, dt);
        // resolveGroundContact :1767-1800
        GroundProbeResult probe = groundProbe(position, velocity, P, query, wasGroundedNear, ld3(C.groundNormal));
        GroundContactState gs = probe.state;
        groundSnap(position, velocity, P, probe);
        if (gs.grounded) {
            float normalUpDelta = gs.normal.y - C.groundNormal[1];
            if (gs.triangleIndex != C.groundTriangleIndex && normalUpDelta > 0.02f) C.groundTransitionFrames = 3;
        }
        slopeFriction(velocity, C, gravity, dt, gs);
        // writeBack :1802-1821
        std3(body.position, d3(position));
        std3(body.linearVelocity, velocity);
        C.flags &= ~(SGE_CTRL_GROUNDED | SGE_CTRL_GROUNDED_NEAR);
        if (gs.grounded) C.flags |= SGE_CTRL_GROUNDED;
        if (gs.groundedNear) C.flags |= SGE_CTRL_GROUNDED_NEAR;
        st3(C.groundNormal, gs.grounded ? gs.normal : V3{0, 1, 0});
        C.groundDistance = gs.distance;
        if (gs.grounded) C.groundTriangleIndex = gs.triangleIndex;
    }
}



// ---- AgentSeparationSystem (Systems.swift:1906-2210) ----
// The reference iterates `world.query(...)`, i.e. Swift Dictionary order, which is hash-seed dependent; the canonical order
// here is the character index. Everything else follows the Swift line by line: the agent list (solid agents;
// an entity without AgentCollisionComponent counts as a default one, :2171), the per-iteration grid rebuild (:2192-2199), the
// resolver's stale copy `a` of agent i beside the live agents[i] / agents[j] (:1952-2043), and the post-process (:2047-2140).
namespace {
struct SepAgent { int entity; V3 position, velocity; float radius, halfHeight, invWeight; };
struct CellKey { long long x, z; bool operator<(const CellKey& o) const { return x != o.x ? x < o.x : z < o.z; } };
CellKey cellCoord(V3 pos, float cellSize) { // :1939-1943
    return CellKey{(long long)floorf(pos.x / cellSize), (long long)floorf(pos.z / cellSize)};
}
}

void agent_separation_fixed_update(World& w, int iterations, float separationMargin, float heightMargin) {
    const CollisionQuery& query = w.query;
    const int N = (int)w.bodies.size();
    if (N <= 1) return; // guard entities.count > 1 (:2153)
    if (iterations < 1) iterations = 1; // :2146
    std::vector<SepAgent> agents;
    std::vector<V3> originalPositions;
    float maxRadius = 0;
    for (int e = 0; e < N; ++e) { // :2166-2187
        const sge_controller_params& P = w.params[e];
        const bool present = (P.agentFlags & SGE_AGENT_PRESENT) != 0;
        const bool solid = present ? (P.agentFlags & SGE_AGENT_SOLID) != 0 : true; // aStore[e] ?? AgentCollisionComponent()
        if (!solid) continue;
        const float radius = (present && (P.agentFlags & SGE_AGENT_RADIUS_OVERRIDE)) ? P.agentRadiusOverride : P.radius;
        const float massWeight = present ? P.agentMassWeight : 1.0f;
        const float invWeight = massWeight > 0 ? 1.0f / massWeight : 0.0f;
        maxRadius = fmax_s(maxRadius, radius);
        V3 pos = f3(ldd3(w.bodies[e].position)), vel = f3(ldd3(w.bodies[e].linearVelocity));
        agents.push_back(SepAgent{e, pos, vel, radius, P.halfHeight, invWeight});
        originalPositions.push_back(pos);
    }
    if (agents.size() <= 1) return; // :2189
    const float cellSize = fmax_s(maxRadius * 2 + separationMargin, 0.001f);
    const int n = (int)agents.size();
    for (int it = 0; it < iterations; ++it) {
        std::map<CellKey, std::vector<int>> cells; // rebuild :1930-1936 (lists in agent order)
        for (int i = 0; i < n; ++i) cells[cellCoord(agents[i].position, cellSize)].push_back(i);
        for (int i = 0; i < n; ++i) { // AgentSeparationResolver.resolve :1947-2046
            const SepAgent a = agents[i];
            const sge_controller_params& Pa = w.params[a.entity];
            const CellKey cell = cellCoord(a.position, cellSize);
            for (int dz = -1; dz <= 1; ++dz) {
                for (int dx = -1; dx <= 1; ++dx) {
                    auto found = cells.find(CellKey{cell.x + dx, cell.z + dz});
                    if (found == cells.end()) continue;
                    for (int j : found->second) {
                        if (!(j > i)) continue;
                        const SepAgent b = agents[j];
                        const sge_controller_params& Pb = w.params[b.entity];
                        float aMin = a.position.y - a.halfHeight, aMax = a.position.y + a.halfHeight;
                        float bMin = b.position.y - b.halfHeight, bMax = b.position.y + b.halfHeight;
                        float ddx = a.position.x - b.position.x, ddz = a.position.z - b.position.z;
                        float distSq = ddx * ddx + ddz * ddz;
                        float skinAllowance = fmin_s(Pa.skinWidth, Pb.skinWidth);
                        float margin = fmin_s(separationMargin, skinAllowance);
                        float minDist = a.radius + b.radius + margin;
                        bool heightSeparated = aMax < bMin - heightMargin || aMin > bMax + heightMargin;
                        if (heightSeparated) continue;
                        if (distSq >= minDist * minDist) continue;
                        float dist = sqrtf(fmax_s(distSq, 1e-8f));
                        float nx = ddx / dist, nz = ddz / dist;
                        float penetration = minDist - dist;
                        float wSum = a.invWeight + b.invWeight;
                        if (wSum <= 0) continue;
                        float corr = penetration / wSum;
                        V3 moveA{nx * corr * a.invWeight, 0, nz * corr * a.invWeight};
                        V3 moveB{-nx * corr * b.invWeight, 0, -nz * corr * b.invWeight};
                        V3 relV = a.velocity - b.velocity;
                        float vn = relV.x * nx + relV.z * nz;
                        if (vn < 0) {
                            float impulse = -vn;
                            float scaleA = a.invWeight / wSum, scaleB = b.invWeight / wSum;
                            agents[i].velocity.x += nx * impulse * scaleA;
                            agents[i].velocity.z += nz * impulse * scaleA;
                            agents[j].velocity.x -= nx * impulse * scaleB;
                            agents[j].velocity.z -= nz * impulse * scaleB;
                        }
                        { // `if let query` :2003
                            const float eps = 1e-6f;
                            bool blockedA = false, blockedB = false;
                            CapsuleCastHit hit;
                            if (length(moveA) > eps &&
                                query.capsuleCastCombined(agents[i].position, moveA, a.radius, a.halfHeight, true, false, 0, Pa.collisionMask, hit) &&
                                hit.toi <= Pa.skinWidth && hit.normal.y < Pa.minGroundDot)
                                blockedA = true;
                            if (length(moveB) > eps &&
                                query.capsuleCastCombined(agents[j].position, moveB, b.radius, b.halfHeight, true, false, 0, Pb.collisionMask, hit) &&
                                hit.toi <= Pb.skinWidth && hit.normal.y < Pb.minGroundDot)
                                blockedB = true;
                            if (blockedA && !blockedB) {
                                moveA = V3{0, 0, 0};
                                moveB = V3{-nx * penetration, 0, -nz * penetration};
                            } else if (blockedB && !blockedA) {
                                moveB = V3{0, 0, 0};
                                moveA = V3{nx * penetration, 0, nz * penetration};
                            } else if (blockedA && blockedB) {
                                continue;
                            }
                        }
                        agents[i].position += moveA;
                        agents[j].position += moveB;
                    }
                }
            }
        }
    }
    for (int idx = 0; idx < n; ++idx) { // :2201-2215 + AgentSeparationPostProcessor.apply :2048-2139
        const SepAgent& agent = agents[idx];
        sge_body_state& body = w.bodies[agent.entity];
        const sge_controller_params& P = w.params[agent.entity];
        sge_controller_state& C = w.controllers[agent.entity];
        const V3 start = originalPositions[idx];
        V3 position = agent.position;
        D3 bodyVelocity = ldd3(body.linearVelocity);
        V3 delta = position - start;
        float len = length(delta);
        bool moved = false;
        if (len > 1e-6f) {
            moved = true;
            V3 remaining = delta;
            position = start;
            for (int s = 0; s < 2; ++s) { // slideIterations
                float segLen = length(remaining);
                if (segLen < 1e-6f) break;
                CapsuleCastHit hit;
                if (query.capsuleCastCombined(position, remaining, agent.radius, agent.halfHeight, true, false, 0, P.collisionMask, hit)) {
                    SlideHit sh;
                    sh.isStatic = true; sh.s = hit; sh.a = CapsuleCapsuleHit{0, V3{0, 0, 0}, -1};
                    bool done = resolveHit(remaining, segLen, sh, P, C, false, false, bodyVelocity, position, false, V3{0, 0, 0}, kAgentSeparation);
                    if (done) break;
                } else {
                    position += remaining;
                    remaining = V3{0, 0, 0};
                    break;
                }
            }
        }
        if (moved && bodyVelocity.y <= 0) { // :2108-2136
            if (P.snapDistance > 0) {
                V3 down{0, -1, 0};
                CapsuleCastHit hit;
                if (query.capsuleCastCombined(position, down * P.snapDistance, agent.radius, agent.halfHeight, false, true, P.minGroundDot, P.collisionMask, hit) &&
                    hit.toi <= P.snapDistance) {
                    float rawMove = fmax_s(hit.toi - P.groundSnapSkin, 0.0f);
                    float moveDist = fmin_s(rawMove, P.groundSnapMaxStep);
                    position += down * moveDist;
                    C.flags |= SGE_CTRL_GROUNDED;
                    if (hit.toi <= fmax_s(P.groundSnapSkin, P.skinWidth)) C.flags |= SGE_CTRL_GROUNDED_NEAR; else C.flags &= ~(uint32_t)SGE_CTRL_GROUNDED_NEAR;
                    V3 gn = hit.material.flattenGround ? V3{0, 1, 0} : hit.triangleNormal;
                    C.groundNormal[0] = gn.x; C.groundNormal[1] = gn.y; C.groundNormal[2] = gn.z;
                    C.groundTriangleIndex = hit.triangleIndex;
                }
            }
        }
        D3 pd = d3(position), vd = d3(agent.velocity);
        body.position[0] = pd.x; body.position[1] = pd.y; body.position[2] = pd.z;
        body.linearVelocity[0] = vd.x; body.linearVelocity[1] = vd.y; body.linearVelocity[2] = vd.z;
    }
}

} // namespace sgeo

// ---- probes: the pieces of the move system the tick only reaches through a whole step, callable one at a time ---------------
// (tests/test_independent_pins.py checks each against an independent float64 answer; same functions the tick runs, no copies)
extern "C" {

// capsuleCapsuleSweep (Systems.swift:1505-1590) for `n` independent pairs: in[i] = from[3], delta[3], radius, halfHeight,
// otherPos[3], otherDelta[3], otherRadius, otherHalfHeight (16 floats); out[i] = hit (0/1), toi, normal[3] (5 floats)
int sgeo_probe_capsule_capsule_sweep(const float* in, int32_t n, float* out) {
    using namespace sgeo;
    for (int i = 0; i < n; ++i) {
        const float* p = in + (size_t)i * 16;
        CapsuleCapsuleHit h{0, V3{0, 0, 0}, 0};
        const bool hit = capsuleCapsuleSweep(ld3(p), ld3(p + 3), p[6], p[7], 1, ld3(p + 8), ld3(p + 11), p[14], p[15], h);
        float* o = out + (size_t)i * 5;
        o[0] = hit ? 1.0f : 0.0f; o[1] = hit ? h.toi : 0.0f; o[2] = h.normal.x; o[3] = h.normal.y; o[4] = h.normal.z;
    }
    return SGE_OK;
}

// AgentSweepSolver.bestHit (Systems.swift:1053-1091) over a snapshot of `count` agents (entity = index); self = `selfEntity`.
// out = hit, toi, normal[3], other (6 floats)
int sgeo_probe_agent_best_hit(const float position[3], const float remaining[3], float remainingLen, float baseMoveLen, float dt,
                              int32_t selfEntity, int32_t selfSolid, float selfRadius, float halfHeight,
                              const sge_agent_state* agents, int32_t count, float* out) {
    using namespace sgeo;
    std::vector<AgentSweepState> list;
    for (int i = 0; i < count; ++i)
        if (agents[i].radius >= 0) list.push_back(AgentSweepState{i, ld3(agents[i].position), ld3(agents[i].velocity), agents[i].radius, agents[i].halfHeight});
    CapsuleCapsuleHit best{0, V3{0, 0, 0}, -1};
    const bool hit = agentBestHit(ld3(position), ld3(remaining), remainingLen, baseMoveLen, dt, selfEntity, selfSolid != 0, selfRadius, halfHeight, list, best);
    out[0] = hit ? 1.0f : 0.0f; out[1] = hit ? best.toi : 0.0f; out[2] = best.normal.x; out[3] = best.normal.y; out[4] = best.normal.z;
    out[5] = hit ? (float)best.other : -1.0f;
    return SGE_OK;
}

// VelocityGate.apply (Systems.swift:1037-1051): velocity in/out (double[3]), remaining out (float[3])
int sgeo_probe_velocity_gate(double velocity[3], int32_t wasGrounded, int32_t wasGroundedNear, float dt, float remaining[3]) {
    using namespace sgeo;
    D3 v = ldd3(velocity);
    st3(remaining, velocityGate(v, wasGrounded != 0, wasGroundedNear != 0, dt));
    std3(velocity, v);
    return SGE_OK;
}

// GroundSnap.apply (Systems.swift:945-963): position / velocity in/out; the probe result as canSnap, hasHit, nearGround, hit.toi, hit.normal
int sgeo_probe_ground_snap(float position[3], double velocity[3], float groundSnapSkin, float groundSnapMaxStep, int32_t canSnap,
                           int32_t hasHit, int32_t nearGround, float toi, const float normal[3]) {
    using namespace sgeo;
    sge_controller_params P{};
    P.groundSnapSkin = groundSnapSkin; P.groundSnapMaxStep = groundSnapMaxStep;
    GroundProbeResult R{};
    R.canSnap = canSnap != 0; R.hasHit = hasHit != 0; R.nearGround = nearGround != 0;
    R.hit.toi = toi; R.hit.normal = ld3(normal);
    V3 pos = ld3(position);
    D3 vel = ldd3(velocity);
    groundSnap(pos, vel, P, R);
    st3(position, pos);
    std3(velocity, vel);
    return SGE_OK;
}

// SlopeFriction.apply (Systems.swift:965-1021): velocity in/out, the controller's groundSliding flag and groundTransitionFrames in/out
int sgeo_probe_slope_friction(double velocity[3], int32_t* groundSliding, int32_t* groundTransitionFrames, const float gravity[3], float dt,
                              int32_t grounded, const float normal[3], float muS, float muK) {
    using namespace sgeo;
    sge_controller_state C{};
    C.flags = *groundSliding ? (uint32_t)SGE_CTRL_GROUND_SLIDING : 0u;
    C.groundTransitionFrames = *groundTransitionFrames;
    GroundContactState st{};
    st.grounded = grounded != 0; st.normal = ld3(normal); st.material = sge_surface_material{muS, muK, 0};
    D3 vel = ldd3(velocity);
    slopeFriction(vel, C, ld3(gravity), dt, st);
    std3(velocity, vel);
    *groundSliding = (C.flags & SGE_CTRL_GROUND_SLIDING) ? 1 : 0;
    *groundTransitionFrames = C.groundTransitionFrames;
    return SGE_OK;
}

} // extern "C"
