// ORACLE — TEST INFRASTRUCTURE ONLY (see sge_oracle_math.h).
//
// CPU float32 restatement of the pose path:
//   MotionProfileEvaluator      Game/Animation.swift:65-89
//   PoseStackSystem.fixedUpdate Game/ProceduralPoseSystem.swift:13-406
//   LocomotionProfileSystem     Game/Systems.swift:276-408
//   ActionAnimationSystem       Game/Systems.swift:472-518
//   palette re-bind             Game/Systems.swift:2519-2527
#include "sge_oracle.h"

namespace sgeo {

// Animation.swift:66-78
static float evaluate(const float* coeffs, int count, float phase, int order) {
    if (count <= 0) return 0;
    float p = fmax_s(0.0f, fmin_s(phase, 1.0f));
    float result = coeffs[0];
    int index = 1;
    for (int k = 1; k <= order; ++k) {
        if (index + 1 >= count) break;
        float angle = 2 * SWIFT_FLOAT_PI * (float)k * p;
        result += coeffs[index] * cosf(angle) + coeffs[index + 1] * sinf(angle);
        index += 2;
    }
    return result;
}

// Animation.swift:80-88; channel 0 = translation, 1 = rotation
static V3 evaluateChannel(const MotionProfile& prof, int bone, int channel, float phase, V3 def) {
    float out[3] = {def.x, def.y, def.z};
    for (int a = 0; a < 3; ++a) {
        int slot = bone * 6 + channel * 3 + a;
        uint8_t cnt = prof.coeffCount[slot];
        if (cnt != SGE_AXIS_ABSENT) out[a] = evaluate(&prof.coeffs[(size_t)slot * SGE_MAX_COEFFS], cnt, phase, prof.order);
    }
    return V3{out[0], out[1], out[2]};
}

static inline float cycleOf(const MotionProfile& p) { return fmax_s(p.cycleDurationRaw, 0.001f); }
static inline V3 col3(const M4& m, int c) { return V3{m.c[c].x, m.c[c].y, m.c[c].z}; }

static void buildModelTransforms(const std::vector<int>& parent, const M4* local, M4* model, int n) { // Skeleton.swift:189-203
    for (int i = 0; i < n; ++i) {
        int p = parent[i];
        model[i] = p < 0 ? local[i] : mul(model[p], local[i]);
    }
}

// single-profile local matrix, ProceduralPoseSystem.swift:249-274 / 296-318
static M4 singleProfileLocal(const Skeleton& sk, const MotionProfile& prof, int i, float phase, bool inPlace) {
    V3 restScaled = sk.restTranslation[i];
    V3 restRaw = sk.rawRestTranslation[i];
    V3 animRaw = evaluateChannel(prof, i, 0, phase, restRaw);
    V3 delta = animRaw - restRaw;
    V3 t = restScaled + (delta * sk.unitScale);
    if (i == 0 && inPlace) { t.x = restScaled.x; t.z = restScaled.z; }
    V3 animR = evaluateChannel(prof, i, 1, phase, V3{0, 0, 0});
    M4 rot = mul(rotationXYZDegrees(sk.preRotationDegrees[i]), rotationXYZDegrees(animR));
    if (i == 0) rot = mul(sk.rootRotationFix, rot);
    return mul(matrix4x4_translation(t.x, t.y, t.z), rot);
}

void pose_fixed_update(World& w, int first, int count, float dt) {
    const Skeleton& sk = w.skeleton;
    const int B = sk.boneCount;
    std::vector<M4> actionLocal(B);
    for (int e = first; e < first + count; ++e) {
        M4* local = &w.local[(size_t)e * B];
        M4* model = &w.model[(size_t)e * B];
        M4* palette = &w.palette[(size_t)e * B];
        sge_locomotion_state& L = w.locomotion[e];
        float runLeanWeight = 0;
        const bool hasLoco = (L.flags & SGE_LOCO_PRESENT) != 0;
        const bool hasMotion = (L.flags & SGE_MOTION_PRESENT) != 0;
        const bool loop = (L.flags & SGE_MOTION_LOOP) != 0;
        const bool inPlace = (L.flags & SGE_MOTION_IN_PLACE) != 0;

        if (hasLoco && hasMotion) { // :36-223
            const MotionProfile* prof4[4] = {&w.profiles[L.profile[0]], &w.profiles[L.profile[1]],
                                             &w.profiles[L.profile[2]], &w.profiles[L.profile[3]]};
            float cyc[4];
            for (int s = 0; s < 4; ++s) cyc[s] = cycleOf(*prof4[s]);
            for (int s = 0; s < 4; ++s) L.time[s] += dt * L.playbackRate;
            if (loop) {
                for (int s = 0; s < 4; ++s) L.time[s] = fmodf(L.time[s], cyc[s]);
            } else {
                for (int s = 0; s < 4; ++s) L.time[s] = fmin_s(L.time[s], cyc[s]);
            }
            bool isBlending = (L.flags & SGE_LOCO_IS_BLENDING) != 0;
            if (isBlending) { // :58-75
                if (L.state == SGE_LOCO_IDLE) {
                    float halfLife = fmax_s(L.idleInertiaHalfLife, 0.001f);
                    float decay = powf(0.5f, dt / halfLife);
                    L.idleInertia *= decay;
                    if (L.idleInertia <= 0.001f) {
                        L.idleInertia = 0;
                        L.blendT = 1.0f;
                        isBlending = false;
                    }
                } else {
                    float blendDuration = fmax_s(L.blendTime, 0.001f);
                    L.blendT = fmin_s(L.blendT + dt / blendDuration, 1.0f);
                    if (L.blendT >= 1.0f) isBlending = false;
                }
            }
            if (isBlending) L.flags |= SGE_LOCO_IS_BLENDING; else L.flags &= ~SGE_LOCO_IS_BLENDING;

            float phase4[4];
            for (int s = 0; s < 4; ++s) phase4[s] = fmax_s(0.0f, fmin_s(L.time[s] / cyc[s], 1.0f));
            L.posePhase = phase4[L.state];

            const int fromState = isBlending ? L.fromState : L.state;
            const int toState = L.state;
            float weightTo = 1.0f; // :101-111
            if (isBlending) {
                if (L.state == SGE_LOCO_IDLE) {
                    float inertia = fmax_s(0.0f, fmin_s(L.idleInertia, 1.0f));
                    weightTo = 1.0f - inertia;
                } else {
                    float t = fmax_s(0.0f, fmin_s(L.blendT, 1.0f));
                    weightTo = t * t * t * (t * (t * 6 - 15) + 10);
                }
            }
            float runWeight; // :112-123
            if (isBlending) {
                if (L.state == SGE_LOCO_RUN) runWeight = weightTo;
                else if (L.fromState == SGE_LOCO_RUN) runWeight = 1.0f - weightTo;
                else runWeight = 0.0f;
            } else {
                runWeight = L.state == SGE_LOCO_RUN ? 1.0f : 0.0f;
            }
            runLeanWeight = runWeight;

            const MotionProfile& fromProfile = *prof4[fromState];
            const MotionProfile& toProfile = *prof4[toState];
            const float fromPhase = phase4[fromState], toPhase = phase4[toState];
            for (int i = 0; i < B; ++i) { // :144-221
                V3 restScaled = sk.restTranslation[i];
                V3 restRaw = sk.rawRestTranslation[i];
                bool fromBone = fromProfile.bonePresent[i] != 0;
                bool toBone = toProfile.bonePresent[i] != 0;
                V3 fromRaw = fromBone ? evaluateChannel(fromProfile, i, 0, fromPhase, restRaw) : restRaw;
                V3 toRaw = toBone ? evaluateChannel(toProfile, i, 0, toPhase, restRaw) : restRaw;
                V3 fromDelta = fromRaw - restRaw;
                V3 toDelta = toRaw - restRaw;
                V3 fromT = restScaled + (fromDelta * sk.unitScale);
                V3 toT = restScaled + (toDelta * sk.unitScale);
                if (i == 0 && inPlace) {
                    fromT.x = restScaled.x; fromT.z = restScaled.z;
                    toT.x = restScaled.x; toT.z = restScaled.z;
                }
                V3 fromR = fromBone ? evaluateChannel(fromProfile, i, 1, fromPhase, V3{0, 0, 0}) : V3{0, 0, 0};
                V3 toR = toBone ? evaluateChannel(toProfile, i, 1, toPhase, V3{0, 0, 0}) : V3{0, 0, 0};
                M4 pre = rotationXYZDegrees(sk.preRotationDegrees[i]);
                M4 fromRot = mul(pre, rotationXYZDegrees(fromR));
                M4 toRot = mul(pre, rotationXYZDegrees(toR));
                if (i == 0) {
                    fromRot = mul(sk.rootRotationFix, fromRot);
                    toRot = mul(sk.rootRotationFix, toRot);
                }
                V3 t = fromT + (toT - fromT) * weightTo;
                Q4 fromQuat = quat_from_matrix(fromRot);
                Q4 toQuat = quat_from_matrix(toRot);
                Q4 rotQuat;
                if (i == 0 && isBlending) { // :206-215 yaw-stable root
                    V3 zAxis = col3(fromRot, 2);
                    float yaw = atan2f(zAxis.x, zAxis.z);
                    Q4 yawQuat = quat_angle_axis(yaw, V3{0, 1, 0});
                    Q4 fromPR = q_mul(q_inverse(yawQuat), fromQuat);
                    Q4 toPR = q_mul(q_inverse(yawQuat), toQuat);
                    Q4 prQuat = q_slerp(fromPR, toPR, weightTo);
                    rotQuat = q_mul(yawQuat, prQuat);
                } else {
                    rotQuat = q_slerp(fromQuat, toQuat, weightTo);
                }
                local[i] = mul(matrix4x4_translation(t.x, t.y, t.z), matrix_from_quat(rotQuat));
            }
        } else if (hasMotion) { // :224-276
            const MotionProfile& prof = w.profiles[L.motionProfile];
            float cycle = cycleOf(prof);
            L.motionTime += dt * L.playbackRate;
            if (loop) L.motionTime = fmodf(L.motionTime, cycle);
            else L.motionTime = fmin_s(L.motionTime, cycle);
            float phase = fmax_s(0.0f, fmin_s(L.motionTime / cycle, 1.0f));
            L.posePhase = phase;
            for (int i = 0; i < B; ++i) local[i] = sk.bindLocal[i];
            for (int i = 0; i < B; ++i) {
                if (!prof.bonePresent[i]) continue;
                local[i] = singleProfileLocal(sk, prof, i, phase, inPlace);
            }
        } else { // :277-284
            for (int i = 0; i < B; ++i) local[i] = sk.bindLocal[i];
        }

        const sge_action_state& A = w.actions[e];
        if ((A.flags & SGE_ACTION_PRESENT) && (A.flags & SGE_ACTION_ACTIVE) && A.weight > 0.001f) { // :286-338
            const MotionProfile& prof = w.profiles[A.profile];
            float cycle = cycleOf(prof);
            float phase = fmax_s(0.0f, fmin_s(A.time / cycle, 1.0f));
            for (int i = 0; i < B; ++i) actionLocal[i] = sk.bindLocal[i];
            for (int i = 0; i < B; ++i) {
                if (!prof.bonePresent[i]) continue;
                actionLocal[i] = singleProfileLocal(sk, prof, i, phase, (A.flags & SGE_ACTION_IN_PLACE) != 0);
            }
            float wgt = fmax_s(0.0f, fmin_s(A.weight, 1.0f));
            float iw = 1 - wgt;
            runLeanWeight *= iw;
            for (int i = 0; i < B; ++i) {
                V3 baseT = col3(local[i], 3);
                V3 actionT = col3(actionLocal[i], 3);
                V3 t = baseT + (actionT - baseT) * wgt;
                Q4 baseQ = quat_from_matrix(local[i]);
                Q4 actionQ = quat_from_matrix(actionLocal[i]);
                Q4 q = q_slerp(baseQ, actionQ, wgt);
                local[i] = mul(matrix4x4_translation(t.x, t.y, t.z), matrix_from_quat(q));
            }
        }

        if (sk.pelvisIndex >= 0) { // :344-394
            const sge_body_state& body = w.bodies[e];
            const sge_controller_state& C = w.controllers[e];
            Q4 trot = Q4{body.transformRotation[0], body.transformRotation[1], body.transformRotation[2], body.transformRotation[3]};
            V3 forward = q_act(trot, V3{0, 0, -1});
            V3 fh = V3{forward.x, 0, forward.z};
            V3 forwardHoriz = length_squared(fh) > 0.0001f ? normalize(fh) : V3{0, 0, -1};
            V3 groundNormal = V3{C.groundNormal[0], C.groundNormal[1], C.groundNormal[2]};
            bool useTilt = (C.flags & SGE_CTRL_GROUNDED_NEAR) != 0;
            const float alignStrength = 0.33f;
            Q4 alignQuat;
            if (!useTilt) {
                alignQuat = quat_angle_axis(0, V3{0, 1, 0});
            } else {
                V3 up = V3{0, 1, 0};
                V3 right = normalize(cross(up, forwardHoriz));
                V3 nProj = normalize(groundNormal - right * dot(groundNormal, right));
                V3 crossUp = cross(up, nProj);
                float angle = atan2f(dot(crossUp, right), dot(up, nProj)) * alignStrength;
                alignQuat = quat_angle_axis(angle, right);
            }
            M4 alignMat = matrix_from_quat(alignQuat);
            local[sk.pelvisIndex] = mul(alignMat, local[sk.pelvisIndex]);

            if (runLeanWeight > 0.001f) {
                buildModelTransforms(sk.parent, local, model, B);
                int leanIndex = sk.leanIndex;
                if (leanIndex >= 0) {
                    const M4& boneModel = model[leanIndex];
                    V3 rightWorld = normalize(col3(boneModel, 0));
                    int parentIndex = sk.parent[leanIndex];
                    V3 rightLocal = rightWorld;
                    if (parentIndex >= 0) {
                        Q4 parentQuat = quat_from_matrix(model[parentIndex]);
                        rightLocal = q_act(q_inverse(parentQuat), rightWorld);
                    }
                    float leanAngle = radians_from_degrees(10.0f) * runLeanWeight;
                    Q4 leanQuat = quat_angle_axis(leanAngle, rightLocal);
                    local[leanIndex] = mul(matrix_from_quat(leanQuat), local[leanIndex]);
                }
            }
        }

        buildModelTransforms(sk.parent, local, model, B); // :396
        const bool rebind = (int)w.mesh.invBindModel.size() == B; // Systems.swift:2523
        for (int i = 0; i < B; ++i)
            palette[i] = mul(model[i], rebind ? w.mesh.invBindModel[i] : sk.invBindModel[i]);
    }
}

// Systems.swift:297-324
static int groundedNextState(int current, float speed, const sge_locomotion_state& L) {
    int groundedState = current == SGE_LOCO_FALLING ? SGE_LOCO_IDLE : current;
    switch (groundedState) {
    case SGE_LOCO_IDLE:
        if (speed >= L.runEnterSpeed) return SGE_LOCO_RUN;
        else if (speed >= L.idleExitSpeed) return SGE_LOCO_WALK;
        return SGE_LOCO_IDLE;
    case SGE_LOCO_WALK:
        if (speed >= L.runEnterSpeed) return SGE_LOCO_RUN;
        else if (speed < L.idleEnterSpeed) return SGE_LOCO_IDLE;
        return SGE_LOCO_WALK;
    case SGE_LOCO_RUN:
        if (speed < L.runExitSpeed) return speed < L.idleEnterSpeed ? SGE_LOCO_IDLE : SGE_LOCO_WALK;
        return SGE_LOCO_RUN;
    default:
        return SGE_LOCO_FALLING;
    }
}

// Systems.swift:279-407
void locomotion_fixed_update(World& w, int first, int count) {
    for (int e = first; e < first + count; ++e) {
        sge_locomotion_state& L = w.locomotion[e];
        if (!(L.flags & SGE_LOCO_PRESENT) || !(L.flags & SGE_MOTION_PRESENT)) continue;
        const sge_body_state& body = w.bodies[e];
        const sge_controller_state& C = w.controllers[e];
        D3 hv = D3{body.linearVelocity[0], 0, body.linearVelocity[2]};
        float speed = (float)length(hv);
        bool isAirborne = !(C.flags & SGE_CTRL_GROUNDED_NEAR);
        int nextState;
        if (isAirborne) {
            bool highFall = C.groundDistance >= L.fallMinDropHeight;
            if (L.state == SGE_LOCO_FALLING || highFall) nextState = SGE_LOCO_FALLING;
            else nextState = groundedNextState(L.state, speed, L);
        } else {
            nextState = groundedNextState(L.state, speed, L);
        }
        if (nextState != L.state) {
            int fromState = L.state;
            float fromCycle = cycleOf(w.profiles[L.profile[fromState]]);
            float fromTime = L.time[fromState];
            float fromPhase = fmax_s(0.0f, fmin_s(fromTime / fromCycle, 1.0f));
            float toCycle = cycleOf(w.profiles[L.profile[nextState]]);
            L.time[nextState] = fromPhase * toCycle;
            L.fromState = L.state;
            L.state = nextState;
            L.flags |= SGE_LOCO_IS_BLENDING;
            L.blendT = 0;
            if (nextState == SGE_LOCO_IDLE) L.idleInertia = 1.0f;
        }
        L.motionTime = L.time[L.state];
    }
}

// Systems.swift:475-517
void action_fixed_update(World& w, int first, int count, float dt) {
    if (!(dt > 0)) return;
    for (int e = first; e < first + count; ++e) {
        sge_action_state& A = w.actions[e];
        if (!(A.flags & SGE_ACTION_PRESENT) || !(A.flags & SGE_ACTION_ACTIVE)) continue;
        float cycle = cycleOf(w.profiles[A.profile]);
        float capTime = cycle;
        if (A.flags & SGE_ACTION_HAS_DODGE) capTime = fmax_s(fmin_s(A.dodgeEnd, cycle), 0.001f);
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
            float halfLife = fmax_s(A.blendOutHalfLife, 0.001f);
            float decay = powf(0.5f, dt / halfLife);
            A.weight *= decay;
            if (A.weight <= 0.001f) {
                A.weight = 0;
                active = false;
                exiting = false;
            }
        } else {
            float blendIn = fmax_s(A.blendInTime, 0.001f);
            A.weight = fmin_s(A.weight + dt / blendIn, 1.0f);
        }
        A.flags &= ~(SGE_ACTION_ACTIVE | SGE_ACTION_EXITING);
        if (active) A.flags |= SGE_ACTION_ACTIVE;
        if (exiting) A.flags |= SGE_ACTION_EXITING;
    }
}

} // namespace sgeo
