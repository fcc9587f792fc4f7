// Pose evaluation + bone palette for gfx950 — replaces, per fixed step,
//   LocomotionProfileSystem.fixedUpdate   Game/Systems.swift:279-407
//   ActionAnimationSystem.fixedUpdate     Game/Systems.swift:475-517
//   PoseStackSystem.fixedUpdate           Game/ProceduralPoseSystem.swift:13-406
//   the palette re-bind of RenderExtract  Game/Systems.swift:2519-2527 (folded into invBind at upload)
//   PhysicsWritebackSystem (rotation)     Game/Systems.swift:2259-2265
//
// Mapping: one wavefront (64 lanes) per character, lane = bone (bones 64.. take
// a second pass).  The reference's per-bone String-keyed dictionary lookups become
// a dense [profile][bone][axis][k] coefficient table; the Fourier harmonics
// cos/sin(2*pi*k*p) depend only on the phase, so they are evaluated once per
// character and profile instead of once per axis.  The parent-before-child model
// product runs level by level through LDS (12 levels for the Y-Bot).  Scalar
// per-character state (clocks, blend weights) is computed redundantly by all lanes
// and written back by lane 0.
//
// Compiled with -ffp-contract=off: same operation order as the float32 oracle.
#include "sge_internal.hpp"

namespace sge {

constexpr int kWave = 64;
constexpr int kRow = 13; // floats per bone row of the LDS matrices (12 used)

// harmonics of one profile at one phase: cos / sin of 2 pi k p for k = 1 .. KMAX (KMAX = the largest order uploaded, 4 or 8)
template <int KMAX> struct Harmonics { float c[KMAX], s[KMAX]; };

// Animation.swift:68,73 — p = clamp(phase), angle = 2 * Float.pi * Float(k) * p
template <int KMAX>
__device__ __forceinline__ void harmonics(float phase, int order, Harmonics<KMAX>& h) {
    float p = smax(0.0f, smin(phase, 1.0f));
#pragma unroll
    for (int k = 1; k <= KMAX; ++k) {
        if (k <= order) {
            float angle = 2 * kSwiftPi * (float)k * p;
            sincosf(angle, &h.s[k - 1], &h.c[k - 1]);
        } else {
            h.c[k - 1] = 0.f; h.s[k - 1] = 0.f;
        }
    }
}

// Animation.swift:66-78 on coefficients already in registers. The reference's loop leaves at the first k with k > order or
// index + 1 >= count (index = 2k - 1): both conditions are monotone in k, so "term k is added" == (k <= order && 2k < count).
template <int KMAX>
__device__ __forceinline__ float evalAxis(const float (&cf)[2 * KMAX + 1], int cnt, int order, const Harmonics<KMAX>& h) {
    float result = cnt > 0 ? cf[0] : 0.0f;
#pragma unroll
    for (int k = 1; k <= KMAX; ++k) {
        const float next = result + (cf[2 * k - 1] * h.c[k - 1] + cf[2 * k] * h.s[k - 1]);
        result = (k <= order && 2 * k < cnt) ? next : result;
    }
    return result;
}

struct BoneEval { F3 t; Aff rot; bool present; };

__device__ __forceinline__ Aff loadAff12(const float* p) {
    return Aff{{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}, {p[9], p[10], p[11]}};
}
__device__ __forceinline__ void storeAff12(float* p, const Aff& a) {
    p[0] = a.c0.x; p[1] = a.c0.y; p[2] = a.c0.z; p[3] = a.c1.x; p[4] = a.c1.y; p[5] = a.c1.z;
    p[6] = a.c2.x; p[7] = a.c2.y; p[8] = a.c2.z; p[9] = a.c3.x; p[10] = a.c3.y; p[11] = a.c3.z;
}

// translation + rotation of bone i under one profile (ProceduralPoseSystem.swift:146-200 / 249-271).
// Written without data-dependent branches: every load of the bone (rest pose, pre-rotation, axis counts, the Fourier rows of the
// three rotation axes) is issued up front and the lanes of a pass go through ONE dependent memory latency instead of one per
// axis. An axis the profile does not have (or a bone it has no entry for) selects the default instead of skipping the evaluation:
// raw = rest and degrees = 0, which is exactly what the skipped form computes (delta = 0, R(0) = identity). The translation rows
// are read only by the root and, when some other bone of the profile carries translation, by everybody (pf.nonRootTranslation).
// The table is padded by one full row at its end, so reading 2 KMAX + 1 floats of a shorter row stays inside it.
template <int KMAX>
__device__ __forceinline__ BoneEval evalBone(const DevSkeleton& sk, const DevProfiles& pf, int prof, int i,
                                             const Harmonics<KMAX>& h, bool inPlace, const Aff& rootFix) {
    constexpr int NC = 2 * KMAX + 1;
    const int B = sk.boneCount;
    const uint8_t* ccp = pf.coeffCount + ((size_t)prof * B + i) * 6;
    const float* co = pf.coeffs + ((size_t)prof * B + i) * 6 * pf.stride;
    const bool present = pf.bonePresent[prof * B + i] != 0;
    int cc[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) cc[a] = ccp[a];
    float cr[3][NC];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int j = 0; j < NC; ++j) cr[a][j] = co[(3 + a) * pf.stride + j];
    const F3 restScaled{sk.restT[i * 3], sk.restT[i * 3 + 1], sk.restT[i * 3 + 2]};
    const F3 restRaw{sk.rawRestT[i * 3], sk.rawRestT[i * 3 + 1], sk.rawRestT[i * 3 + 2]};
    const Aff preRot = loadAff12(sk.preRot + i * 12);
    const int order = pf.order[prof];
    float raw[3] = {restRaw.x, restRaw.y, restRaw.z};
    if (i == 0 || ((pf.nonRootTranslation >> prof) & 1u)) {
        float ct[3][NC];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int j = 0; j < NC; ++j) ct[a][j] = co[a * pf.stride + j];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = evalAxis<KMAX>(ct[a], cc[a], order, h);
            raw[a] = (present && cc[a] != SGE_AXIS_ABSENT) ? v : raw[a];
        }
    }
    float deg[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = evalAxis<KMAX>(cr[a], cc[3 + a], order, h);
        deg[a] = (present && cc[3 + a] != SGE_AXIS_ABSENT) ? v : 0.0f;
    }
    F3 delta = F3{raw[0], raw[1], raw[2]} - restRaw;
    F3 t = restScaled + (delta * sk.unitScale);
    if (i == 0 && inPlace) { t.x = restScaled.x; t.z = restScaled.z; }
    Aff rot = affMul(preRot, rotationXYZDegrees(F3{deg[0], deg[1], deg[2]}));
    if (i == 0) rot = affMul(rootFix, rot);
    BoneEval r;
    r.t = t;
    r.rot = rot;
    r.present = present;
    return r;
}

// Run-time index into a four-entry array that lives in registers: written as selects, because a dynamic subscript makes the
// compiler spill the whole enclosing struct (the locomotion state, 100 B per lane) to scratch and fetch every access from there.
template <class T> __device__ __forceinline__ T at4(const T (&a)[4], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3])); }
template <class T> __device__ __forceinline__ void set4(T (&a)[4], int i, T v) {
    a[0] = i == 0 ? v : a[0]; a[1] = i == 1 ? v : a[1]; a[2] = i == 2 ? v : a[2]; a[3] = (i != 0 && i != 1 && i != 2) ? v : a[3];
}
__device__ __forceinline__ float cycleOf(const DevProfiles& pf, int p) { return smax(pf.cycleRaw[p], 0.001f); }

// Systems.swift:297-324
__device__ __forceinline__ int groundedNextState(int current, float speed, const sge_locomotion_state& L) {
    int g = current == SGE_LOCO_FALLING ? SGE_LOCO_IDLE : current;
    if (g == SGE_LOCO_IDLE) {
        if (speed >= L.runEnterSpeed) return SGE_LOCO_RUN;
        if (speed >= L.idleExitSpeed) return SGE_LOCO_WALK;
        return SGE_LOCO_IDLE;
    }
    if (g == SGE_LOCO_WALK) {
        if (speed >= L.runEnterSpeed) return SGE_LOCO_RUN;
        if (speed < L.idleEnterSpeed) return SGE_LOCO_IDLE;
        return SGE_LOCO_WALK;
    }
    if (g == SGE_LOCO_RUN) {
        if (speed < L.runExitSpeed) return speed < L.idleEnterSpeed ? SGE_LOCO_IDLE : SGE_LOCO_WALK;
        return SGE_LOCO_RUN;
    }
    return SGE_LOCO_FALLING;
}

#ifndef SGE_POSE_WAVES
#define SGE_POSE_WAVES 1
#endif
// Bones are visited by SLOT: slot s of pass s / 64 is bone sk.slotBone[s]. The host orders the slots so that the bones of the LAST,
// partly filled pass (bone 65 of the Y-Bot's 65) are ones no uploaded profile animates and whose parents sit in earlier passes
// (sge_api.hip: rebuildPoseSlots): with sk.lastPassStatic that pass evaluates nothing (pre-rotation and rest translation are the
// local matrix) and with sk.extraParentReady its model matrices are ONE product with the parent's finished model instead of a walk
// down the ancestor path. Before, the second pass repeated the whole per-bone chain — loads, six trig evaluations, the path walk —
// for a single lane: 40 % of the wavefront's time.
template <int KMAX>
__global__ __launch_bounds__(kWave, SGE_POSE_WAVES) void pose_kernel(PoseLaunch K) {
    // local and model matrices of this character's bones: sized by the launch for the skeleton's bone count
    // (a fixed SGE_MAX_BONES-sized array would cap the CU at 6 workgroups for a 65-bone rig)
    // (rows of kRow = 13 floats: with 12 the lane stride is a multiple of four banks and every per-lane row access a four-way bank
    // conflict — SQ_LDS_BANK_CONFLICT was 39 % of SQ_LDS_IDX_ACTIVE, profiles/r2_move_pmc_cheese.json)
    extern __shared__ float sPose[];
#ifdef SGE_POSE_SETPRIO // experiment: issue priority over the LBS wavefronts on the same SIMD
    __builtin_amdgcn_s_setprio(SGE_POSE_SETPRIO);
#endif
    float* const sLocal = sPose;
    float* const sModel = sPose + (size_t)K.sk.boneCount * kRow;
    const int e = K.first + blockIdx.x;
    const int lane = threadIdx.x;
    const DevSkeleton& sk = K.sk;
    const DevProfiles& pf = K.prof;
    const int B = sk.boneCount;
    const float dt = K.dt;
    // diagnostics (SGE_WAVE_PROF): cycle stamps of the phases, one row of 8 x u64 per character
    long long pS[6] = {0, 0, 0, 0, 0, 0};
#define SGE_POSE_STAMP(k) do { if (K.waveProf) pS[k] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    SGE_POSE_STAMP(0);
    // this lane's bones of the first two passes, requested before anything else (their latency hides behind the state machines)
    const int bone0 = lane < B ? sk.slotBone[lane] : 0;
    const int bone1 = kWave + lane < B ? sk.slotBone[kWave + lane] : 0;
    auto boneOfSlot = [&](int s) -> int { return s < kWave ? bone0 : (s < 2 * kWave ? bone1 : sk.slotBone[s]); };
    const int lastPassFirst = ((B - 1) / kWave) * kWave; // first slot of the last pass

    sge_locomotion_state L = K.crowd.locomotion[e];
    sge_action_state A = K.crowd.actions[e];
    const sge_body_state& body = K.crowd.bodies[e];
    const sge_controller_state& C = K.crowd.controllers[e];
    // what the stages read of the move stage's results: from the move stage's copy when the launch runs beside the NEXT step's move
    // stage (K.crowd.poseIn), from the body / controller themselves otherwise. transformRotation is this kernel's own.
    const PoseInput* const pin = K.crowd.poseIn ? K.crowd.poseIn + e : nullptr;
    const double* const velocityIn = pin ? pin->linearVelocity : body.linearVelocity;
    const float* const rotationIn = pin ? pin->rotation : body.rotation;
    const float* const groundNormalIn = pin ? pin->groundNormal : C.groundNormal;
    const float* const groundDistanceIn = pin ? &pin->groundDistance : &C.groundDistance;
    const uint32_t ctrlFlags = pin ? pin->flags : C.flags;
    const bool hasLoco = (L.flags & SGE_LOCO_PRESENT) != 0;
    const bool hasMotion = (L.flags & SGE_MOTION_PRESENT) != 0;

    // ---- LocomotionProfileSystem (Systems.swift:326-406) ----
    if ((K.stages & SGE_STAGE_LOCOMOTION) && hasLoco && hasMotion) {
        D3 hv{velocityIn[0], 0.0, velocityIn[2]};
        float speed = (float)length(hv);
        bool isAirborne = !(ctrlFlags & SGE_CTRL_GROUNDED_NEAR);
        int nextState;
        if (isAirborne) {
            bool highFall = *groundDistanceIn >= L.fallMinDropHeight;
            if (L.state == SGE_LOCO_FALLING || highFall) nextState = SGE_LOCO_FALLING;
            else nextState = groundedNextState(L.state, speed, L);
        } else {
            nextState = groundedNextState(L.state, speed, L);
        }
        if (nextState != L.state) {
            int fromState = L.state;
            float fromCycle = cycleOf(pf, at4(L.profile, fromState));
            float fromPhase = smax(0.0f, smin(at4(L.time, fromState) / fromCycle, 1.0f));
            float toCycle = cycleOf(pf, at4(L.profile, nextState));
            set4(L.time, nextState, fromPhase * toCycle);
            L.fromState = L.state;
            L.state = nextState;
            L.flags |= SGE_LOCO_IS_BLENDING;
            L.blendT = 0;
            if (nextState == SGE_LOCO_IDLE) L.idleInertia = 1.0f;
        }
        L.motionTime = at4(L.time, L.state);
    }

    // ---- ActionAnimationSystem (Systems.swift:482-516) ----
    if ((K.stages & SGE_STAGE_ACTION) && dt > 0 && (A.flags & SGE_ACTION_PRESENT) && (A.flags & SGE_ACTION_ACTIVE)) {
        float cycle = cycleOf(pf, A.profile);
        float capTime = cycle;
        if (A.flags & SGE_ACTION_HAS_DODGE) capTime = smax(smin(A.dodgeEnd, cycle), 0.001f);
        bool exiting = (A.flags & SGE_ACTION_EXITING) != 0;
        if (!exiting) {
            A.time += dt * A.playbackRate;
            if (A.flags & SGE_ACTION_LOOP) {
                A.time = fmodf(A.time, capTime);
            } else if (A.time >= capTime) {
                A.time = capTime;
                exiting = true;
            }
        }
        bool active = true;
        if (exiting) {
            float halfLife = smax(A.blendOutHalfLife, 0.001f);
            float decay = powf(0.5f, dt / halfLife);
            A.weight *= decay;
            if (A.weight <= 0.001f) { A.weight = 0; active = false; exiting = false; }
        } else {
            float blendIn = smax(A.blendInTime, 0.001f);
            A.weight = smin(A.weight + dt / blendIn, 1.0f);
        }
        A.flags &= ~(uint32_t)(SGE_ACTION_ACTIVE | SGE_ACTION_EXITING);
        if (active) A.flags |= SGE_ACTION_ACTIVE;
        if (exiting) A.flags |= SGE_ACTION_EXITING;
    }
    // the per-character state is written back as soon as it is final (not at the end of the kernel): 32 dwords per lane that need
    // not stay in registers across the bone loops
    if (lane == 0 && (K.stages & SGE_STAGE_ACTION)) K.crowd.actions[e] = A;

    SGE_POSE_STAMP(1);
    if (K.stages & SGE_STAGE_POSE) {
        const bool loop = (L.flags & SGE_MOTION_LOOP) != 0;
        const bool inPlace = (L.flags & SGE_MOTION_IN_PLACE) != 0;
        const Aff rootFix = loadAff12(sk.rootFix);
        float runLeanWeight = 0;
        // a bone of a static last pass: no profile animates it, its local matrix is pre-rotation + rest translation
        auto staticLocal = [&](int i) -> Aff {
            Aff loc = loadAff12(sk.preRot + i * 12);
            if (i == 0) loc = affMul(rootFix, loc);
            loc.c3 = F3{sk.restT[i * 3], sk.restT[i * 3 + 1], sk.restT[i * 3 + 2]};
            return loc;
        };

        if (hasLoco && hasMotion) { // ProceduralPoseSystem.swift:36-223
            float cyc[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) cyc[s] = cycleOf(pf, L.profile[s]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                L.time[s] += dt * L.playbackRate;
                L.time[s] = loop ? fmodf(L.time[s], cyc[s]) : smin(L.time[s], cyc[s]);
            }
            bool isBlending = (L.flags & SGE_LOCO_IS_BLENDING) != 0;
            if (isBlending) {
                if (L.state == SGE_LOCO_IDLE) {
                    float halfLife = smax(L.idleInertiaHalfLife, 0.001f);
                    float decay = powf(0.5f, dt / halfLife);
                    L.idleInertia *= decay;
                    if (L.idleInertia <= 0.001f) { L.idleInertia = 0; L.blendT = 1.0f; isBlending = false; }
                } else {
                    float blendDuration = smax(L.blendTime, 0.001f);
                    L.blendT = smin(L.blendT + dt / blendDuration, 1.0f);
                    if (L.blendT >= 1.0f) isBlending = false;
                }
            }
            if (isBlending) L.flags |= SGE_LOCO_IS_BLENDING; else L.flags &= ~(uint32_t)SGE_LOCO_IS_BLENDING;
            float phase4[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) phase4[s] = smax(0.0f, smin(L.time[s] / cyc[s], 1.0f));
            L.posePhase = at4(phase4, L.state & 3);
            if (lane == 0) K.crowd.locomotion[e] = L;

            const int fromState = (isBlending ? L.fromState : L.state) & 3;
            const int toState = L.state & 3;
            float weightTo = 1.0f;
            if (isBlending) {
                if (L.state == SGE_LOCO_IDLE) {
                    float inertia = smax(0.0f, smin(L.idleInertia, 1.0f));
                    weightTo = 1.0f - inertia;
                } else {
                    float t = smax(0.0f, smin(L.blendT, 1.0f));
                    weightTo = t * t * t * (t * (t * 6 - 15) + 10);
                }
            }
            float runWeight;
            if (isBlending) {
                if (L.state == SGE_LOCO_RUN) runWeight = weightTo;
                else if (L.fromState == SGE_LOCO_RUN) runWeight = 1.0f - weightTo;
                else runWeight = 0.0f;
            } else {
                runWeight = L.state == SGE_LOCO_RUN ? 1.0f : 0.0f;
            }
            runLeanWeight = runWeight;

            const int fromProf = at4(L.profile, fromState), toProf = at4(L.profile, toState);
            Harmonics<KMAX> hTo, hFrom;
            harmonics<KMAX>(at4(phase4, toState), pf.order[toProf], hTo);
            const bool sameEval = fromState == toState;
            if (!sameEval) harmonics<KMAX>(at4(phase4, fromState), pf.order[fromProf], hFrom);
            for (int s = lane; s < B; s += kWave) {
                const int i = boneOfSlot(s);
                if (sk.lastPassStatic && s >= lastPassFirst) {
                    // lerp(t, t) and slerp(q, q): the reference's matrix -> quaternion -> matrix round trip only re-rounds (~1e-7)
                    storeAff12(sLocal + i * kRow, staticLocal(i));
                    continue;
                }
                BoneEval to = evalBone<KMAX>(sk, pf, toProf, i, hTo, inPlace, rootFix);
                if (sameEval && !isBlending) {
                    // one profile, no blend: lerp(t, t) = t and slerp(q, q) = q. The reference still goes matrix ->
                    // quaternion -> slerp -> matrix, which only re-rounds the rotation (~1e-7); skip the round trip.
                    Aff loc = to.rot;
                    loc.c3 = to.t;
                    storeAff12(sLocal + i * kRow, loc);
                    continue;
                }
                // (the matrices die as soon as their quaternions exist: two bone evaluations never hold registers together)
                const Quat toQuat = quatFromRotation(to.rot);
                const F3 toT = to.t;
                Quat fromQuat = toQuat;
                F3 fromT = toT;
                float yawX = to.rot.c2.x, yawZ = to.rot.c2.z; // of the FROM rotation (:210)
                if (!sameEval) {
                    BoneEval from = evalBone<KMAX>(sk, pf, fromProf, i, hFrom, inPlace, rootFix);
                    fromQuat = quatFromRotation(from.rot);
                    fromT = from.t;
                    yawX = from.rot.c2.x; yawZ = from.rot.c2.z;
                }
                F3 t = fromT + (toT - fromT) * weightTo;
                Quat rotQuat;
                if (i == 0 && isBlending) { // yaw-stable root, :206-215
                    const float yaw = atan2f(yawX, yawZ);
                    Quat yawQuat = quatAngleAxis(yaw, F3{0, 1, 0});
                    Quat fromPR = quatMul(quatInverse(yawQuat), fromQuat);
                    Quat toPR = quatMul(quatInverse(yawQuat), toQuat);
                    rotQuat = quatMul(yawQuat, quatSlerp(fromPR, toPR, weightTo));
                } else {
                    rotQuat = quatSlerp(fromQuat, toQuat, weightTo);
                }
                Aff loc = rotationFromQuat(rotQuat);
                loc.c3 = t;
                storeAff12(sLocal + i * kRow, loc);
            }
        } else if (hasMotion) { // :224-276
            const int prof = L.motionProfile;
            float cycle = cycleOf(pf, prof);
            L.motionTime += dt * L.playbackRate;
            L.motionTime = loop ? fmodf(L.motionTime, cycle) : smin(L.motionTime, cycle);
            float phase = smax(0.0f, smin(L.motionTime / cycle, 1.0f));
            L.posePhase = phase;
            if (lane == 0) K.crowd.locomotion[e] = L;
            Harmonics<KMAX> h;
            harmonics<KMAX>(phase, pf.order[prof], h);
            for (int s = lane; s < B; s += kWave) {
                const int i = boneOfSlot(s);
                if (sk.lastPassStatic && s >= lastPassFirst) { // no entry in any profile: bindLocal (:265-270)
                    storeAff12(sLocal + i * kRow, loadAff12(sk.bindLocal + i * 12));
                    continue;
                }
                BoneEval ev = evalBone<KMAX>(sk, pf, prof, i, h, inPlace, rootFix);
                Aff loc = ev.rot;
                loc.c3 = ev.t;
                if (!ev.present) loc = loadAff12(sk.bindLocal + i * 12);
                storeAff12(sLocal + i * kRow, loc);
            }
        } else { // :277-284
            for (int i = lane; i < B; i += kWave) storeAff12(sLocal + i * kRow, loadAff12(sk.bindLocal + i * 12));
        }
        __syncthreads();
        SGE_POSE_STAMP(2);

        // ---- action layer :286-338 ----
        if ((A.flags & SGE_ACTION_PRESENT) && (A.flags & SGE_ACTION_ACTIVE) && A.weight > 0.001f) {
            const int prof = A.profile;
            float cycle = cycleOf(pf, prof);
            float phase = smax(0.0f, smin(A.time / cycle, 1.0f));
            Harmonics<KMAX> h;
            harmonics<KMAX>(phase, pf.order[prof], h);
            float wgt = smax(0.0f, smin(A.weight, 1.0f));
            float iw = 1 - wgt;
            runLeanWeight *= iw;
            const bool actInPlace = (A.flags & SGE_ACTION_IN_PLACE) != 0;
            for (int s = lane; s < B; s += kWave) {
                const int i = boneOfSlot(s);
                Aff act;
                if (sk.lastPassStatic && s >= lastPassFirst) {
                    act = loadAff12(sk.bindLocal + i * 12);
                } else {
                    BoneEval ev = evalBone<KMAX>(sk, pf, prof, i, h, actInPlace, rootFix);
                    act = ev.rot;
                    act.c3 = ev.t;
                    if (!ev.present) act = loadAff12(sk.bindLocal + i * 12);
                }
                Aff base = loadAff12(sLocal + i * kRow);
                F3 t = base.c3 + (act.c3 - base.c3) * wgt;
                Quat q = quatSlerp(quatFromRotation(base), quatFromRotation(act), wgt);
                Aff loc = rotationFromQuat(q);
                loc.c3 = t;
                storeAff12(sLocal + i * kRow, loc);
            }
            __syncthreads();
        }

        // ---- ground align + run lean :344-394 (uniform; lane 0 commits) ----
        if (sk.pelvisIndex >= 0) {
            Quat trot{body.transformRotation[0], body.transformRotation[1], body.transformRotation[2], body.transformRotation[3]};
            F3 forward = quatAct(trot, F3{0, 0, -1});
            F3 fh{forward.x, 0, forward.z};
            F3 forwardHoriz = lengthSq(fh) > 0.0001f ? normalize(fh) : F3{0, 0, -1};
            F3 groundNormal{groundNormalIn[0], groundNormalIn[1], groundNormalIn[2]};
            bool useTilt = (ctrlFlags & SGE_CTRL_GROUNDED_NEAR) != 0;
            Quat alignQuat;
            if (!useTilt) {
                alignQuat = quatAngleAxis(0, F3{0, 1, 0});
            } else {
                F3 up{0, 1, 0};
                F3 right = normalize(cross(up, forwardHoriz));
                F3 nProj = normalize(groundNormal - right * dot(groundNormal, right));
                F3 crossUp = cross(up, nProj);
                float angle = atan2f(dot(crossUp, right), dot(up, nProj)) * 0.33f;
                alignQuat = quatAngleAxis(angle, right);
            }
            if (lane == 0) {
                Aff p = affMul(rotationFromQuat(alignQuat), loadAff12(sLocal + sk.pelvisIndex * kRow));
                storeAff12(sLocal + sk.pelvisIndex * kRow, p);
            }
            __syncthreads();
            if (runLeanWeight > 0.001f && sk.leanIndex >= 0) {
                // model[leanIndex] needs only the ancestor chain of leanIndex (Skeleton.swift:189-203)
                Aff m = loadAff12(sLocal + sk.leanChain[0] * kRow), parentModel = m;
                for (int k = 1; k < sk.leanChainLen; ++k) {
                    parentModel = m;
                    m = affMul(m, loadAff12(sLocal + sk.leanChain[k] * kRow));
                }
                F3 rightWorld = normalize(m.c0);
                F3 rightLocal = rightWorld;
                if (sk.leanParent >= 0) {
                    Quat parentQuat = quatFromRotation(parentModel);
                    rightLocal = quatAct(quatInverse(parentQuat), rightWorld);
                }
                float leanAngle = radiansFromDegrees(10.0f) * runLeanWeight;
                Quat leanQuat = quatAngleAxis(leanAngle, rightLocal);
                __syncthreads();
                if (lane == 0) {
                    Aff l = affMul(rotationFromQuat(leanQuat), loadAff12(sLocal + sk.leanIndex * kRow));
                    storeAff12(sLocal + sk.leanIndex * kRow, l);
                }
                __syncthreads();
            }
        }

        SGE_POSE_STAMP(3);
        // ---- model transforms, then the palette :396-402 ----
        // model[i] = model[parent] * local[i] unrolls to local[root] * ... * local[i] multiplied left to right; every lane
        // walks its own bone's ancestor path in that order (the same sequence of products, hence the same bits, as the
        // reference's parent-before-child loop) — no level-by-level barriers. A bone of a later pass whose parent's model is
        // finished (sk.extraParentReady) takes the reference's own form instead: one product with model[parent].
        {
            const int stride = sk.maxDepth + 1;
            for (int s = lane; s - lane < B; s += kWave) { // (uniform trip count: the barrier below is inside the loop)
                if (s < B) {
                    const int i = boneOfSlot(s);
                    Aff mod;
                    if (s >= kWave && sk.extraParentReady) {
                        mod = affMul(loadAff12(sModel + sk.parent[i] * kRow), loadAff12(sLocal + i * kRow));
                    } else {
                        const int32_t* pth = sk.path + (size_t)i * stride;
                        const int dep = sk.depth[i];
                        mod = loadAff12(sLocal + pth[0] * kRow);
                        for (int k = 1; k <= dep; ++k) mod = affMul(mod, loadAff12(sLocal + pth[k] * kRow));
                    }
                    storeAff12(sModel + i * kRow, mod);
                }
                __syncthreads();
            }
        }
        SGE_POSE_STAMP(4);
        float* pal = K.crowd.palettes + ((size_t)e * B) * 16;
        for (int s = lane; s < B; s += kWave) {
            const int i = boneOfSlot(s);
            Aff M = loadAff12(sModel + i * kRow);
            const float* ib = sk.invBind + i * 16;
            float4* out = reinterpret_cast<float4*>(pal + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float b0 = ib[j * 4], b1 = ib[j * 4 + 1], b2 = ib[j * 4 + 2], b3 = ib[j * 4 + 3];
                F3 c = ((M.c0 * b0 + M.c1 * b1) + M.c2 * b2) + M.c3 * b3;
                float w = ((0.0f * b0 + 0.0f * b1) + 0.0f * b2) + 1.0f * b3;
                out[j] = make_float4(c.x, c.y, c.z, w);
            }
            if (K.crowd.poseModel) {
                float* dm = K.crowd.poseModel + ((size_t)e * B + i) * 16;
                float* dl = K.crowd.poseLocal + ((size_t)e * B + i) * 16;
                Aff Lc = loadAff12(sLocal + i * kRow);
                const F3 mc[4] = {M.c0, M.c1, M.c2, M.c3}, lc[4] = {Lc.c0, Lc.c1, Lc.c2, Lc.c3};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    reinterpret_cast<float4*>(dm)[j] = make_float4(mc[j].x, mc[j].y, mc[j].z, j == 3 ? 1.f : 0.f);
                    reinterpret_cast<float4*>(dl)[j] = make_float4(lc[j].x, lc[j].y, lc[j].z, j == 3 ? 1.f : 0.f);
                }
            }
        }
    }

    if (K.waveProf && lane == 0) { // state machines | locals | action + ground align + lean | model products | palette
        SGE_POSE_STAMP(5);
        unsigned long long* w = K.waveProf + (size_t)e * 8;
        w[0] = (unsigned long long)(pS[5] - pS[0]);
        for (int k = 1; k <= 5; ++k) w[k] = (unsigned long long)(pS[k] - pS[k - 1]);
        w[7] = (unsigned long long)pS[0];
    }
#undef SGE_POSE_STAMP
    if (lane == 0) {
        // (a pose stage with a motion profile has already stored the state, at the point where the clocks were advanced)
        const bool storedByPose = (K.stages & SGE_STAGE_POSE) && hasMotion;
        if ((K.stages & (SGE_STAGE_LOCOMOTION | SGE_STAGE_POSE)) && !storedByPose) K.crowd.locomotion[e] = L;
        if (K.stages & SGE_STAGE_WRITEBACK) {
            sge_body_state& b = K.crowd.bodies[e];
#pragma unroll
            for (int k = 0; k < 4; ++k) b.transformRotation[k] = rotationIn[k];
        }
    }
}

void launch_pose(const PoseLaunch& L, hipStream_t s) {
    if (L.count <= 0) return;
    const size_t lds = (size_t)L.sk.boneCount * kRow * 2 * sizeof(float);
    if (L.prof.maxOrder <= 4) hipLaunchKernelGGL(pose_kernel<4>, dim3(L.count), dim3(kWave), lds, s, L);
    else hipLaunchKernelGGL(pose_kernel<SGE_MAX_FOURIER_ORDER>, dim3(L.count), dim3(kWave), lds, s, L);
}

} // namespace sge
