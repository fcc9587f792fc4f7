// Header-only C++ host mirror of the reference's Swift surfaces for this path, forwarding to the
// C ABI of include/sge_amd.h.  The reference is compiled Swift with no toolchain in this image, so
// the host side above the ABI is C++ (and a ctypes/numpy driver for tests); names, argument meaning
// and the nil/optional error convention follow the Swift originals so that call sites read alike:
//
//   sge::CollisionQuery            Game/CollisionQuery.swift:54-160
//   sge::RTSkinningEncoder         Game/RTSkinningEncoder.swift:10-57 (+ RTSkinningJob, RTGeometryCache.swift:43-52)
//   sge::KinematicMoveStopSystem   Game/Systems.swift:1402-1415, 1823-1902  (FixedStepSystem, Systems.swift:15-17)
//   sge::PoseStackSystem           Game/ProceduralPoseSystem.swift:10-13
//   sge::LocomotionProfileSystem   Game/Systems.swift:276-279
//   sge::ActionAnimationSystem     Game/Systems.swift:472-475
//   sge::EntityWorld, sge::GPUCrowd, sge::GPUCharacterStepSystem
//                                  the World <-> GPU bridge for a host that keeps its component stores (World.swift:64-75):
//                                  C++ twin of host/swift/GPUCrowd.swift + GPUCharacterStepSystem, over std::unordered_map stores
//
// `World` here is the crowd owned by one GPU context: the reference's per-entity dictionary stores
// (World.swift:64-75) become the context's resident arrays, so `fixedUpdate(world, dt)` is one batched call.
#pragma once
#include <algorithm>
#include <array>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "../../include/sge_amd.h"

namespace sge {

struct float3 { float x, y, z; };

class Error : public std::runtime_error {
public:
    explicit Error(const std::string& what) : std::runtime_error(what + ": " + sge_last_error()) {}
};
inline void check(int rc, const char* what) { if (rc != SGE_OK) throw Error(what); }
constexpr uint32_t bit(bool on, uint32_t flag) { return on ? flag : 0u; }

// The crowd resident on one GPU.
class World {
public:
    // Mirrors the reference's failable initialisers: returns nullptr when no gfx950 device exists.
    static std::unique_ptr<World> make(int deviceIndex = 0) {
        sge_context* c = sge_context_create(deviceIndex);
        if (!c) return nullptr;
        return std::unique_ptr<World>(new World(c));
    }
    ~World() { sge_context_destroy(ctx_); }
    World(const World&) = delete;
    World& operator=(const World&) = delete;

    sge_context* context() const { return ctx_; }
    void uploadSkeleton(const sge_skeleton_desc& d) { check(sge_skeleton_upload(ctx_, &d), "sge_skeleton_upload"); boneCount_ = d.boneCount; }
    void uploadMotionProfiles(const std::vector<sge_motion_profile_desc>& p) { check(sge_motion_profiles_upload(ctx_, p.data(), (int32_t)p.size()), "sge_motion_profiles_upload"); }
    void uploadSkinnedMesh(const sge_skinned_mesh_desc& d) { check(sge_skinned_mesh_upload(ctx_, &d), "sge_skinned_mesh_upload"); vertexCount_ = d.vertexCount; }
    void resize(int32_t n) { check(sge_characters_resize(ctx_, n), "sge_characters_resize"); count_ = n; }
    void upload(int32_t first, int32_t count, const sge_body_state* b, const sge_controller_params* p,
                const sge_controller_state* c, const sge_move_intent* i, const sge_locomotion_state* l,
                const sge_action_state* a) { check(sge_characters_upload(ctx_, first, count, b, p, c, i, l, a), "sge_characters_upload"); }
    void download(int32_t first, int32_t count, sge_body_state* b, sge_controller_params* p, sge_controller_state* c,
                  sge_move_intent* i, sge_locomotion_state* l, sge_action_state* a) { check(sge_characters_download(ctx_, first, count, b, p, c, i, l, a), "sge_characters_download"); }
    void tick(float dt, uint32_t stages, float3 gravity = {0, -98.0f, 0}) {
        sge_tick_desc d{dt, {gravity.x, gravity.y, gravity.z}, stages, 0, 0, 0};
        check(sge_tick(ctx_, &d), "sge_tick");
    }
    void synchronize() { check(sge_synchronize(ctx_), "sge_synchronize"); }
    int32_t count() const { return count_; }
    int32_t boneCount() const { return boneCount_; }
    int32_t vertexCount() const { return vertexCount_; }

private:
    explicit World(sge_context* c) : ctx_(c) {}
    sge_context* ctx_;
    int32_t count_ = 0, boneCount_ = 0, vertexCount_ = 0;
};

// ---- FixedStepSystem conformers (Systems.swift:15-17) ------------------------------------------
struct FixedStepSystem {
    virtual ~FixedStepSystem() = default;
    virtual void fixedUpdate(World& world, float dt) = 0;
};

// PhysicsIntentSystem + GravitySystem + KinematicMoveStopSystem in the reference's order
// (DemoScene.swift:62-68); init(gravity:) as Systems.swift:1407.
// init(gravity:contactCachePolicy:) — the two policies the reference defines (Systems.swift:1102-1157); a ContactCachePolicy is host
// code, so anything else cannot run inside the kernels and has no value here
enum class ContactCachePolicy { Default, SideContactOnly };
class KinematicMoveStopSystem : public FixedStepSystem {
public:
    explicit KinematicMoveStopSystem(float3 gravity = {0, -98.0f, 0}, bool applyIntentAndGravity = true,
                                     ContactCachePolicy contactCachePolicy = ContactCachePolicy::Default)
        : gravity_(gravity), pre_(applyIntentAndGravity), policy_(contactCachePolicy) {}
    void fixedUpdate(World& world, float dt) override {
        world.tick(dt, (pre_ ? (SGE_STAGE_INTENT | SGE_STAGE_GRAVITY) : 0u) | SGE_STAGE_MOVE |
                       bit(policy_ == ContactCachePolicy::SideContactOnly, SGE_STAGE_SIDE_CONTACT_CACHE), gravity_);
    }
    // the platform entities of world.query(PhysicsBody, Transform, StaticMesh, KinematicPlatform) (Systems.swift:1832-1835)
    // as PlatformCarry reads them; call once per step after the platform motion system ran
    static void setPlatforms(World& world, const std::vector<sge_platform_state>& platforms) {
        check(sge_platforms_upload(world.context(), platforms.data(), (int32_t)platforms.size()), "sge_platforms_upload");
    }
private:
    float3 gravity_;
    bool pre_;
    ContactCachePolicy policy_;
};
struct LocomotionProfileSystem : FixedStepSystem {
    void fixedUpdate(World& world, float dt) override { world.tick(dt, SGE_STAGE_LOCOMOTION); }
};
struct ActionAnimationSystem : FixedStepSystem {
    void fixedUpdate(World& world, float dt) override { world.tick(dt, SGE_STAGE_ACTION); }
};
struct PoseStackSystem : FixedStepSystem {
    void fixedUpdate(World& world, float dt) override { world.tick(dt, SGE_STAGE_POSE); }
};

// ---- CollisionQuery (CollisionQuery.swift:54-160) ------------------------------------------------
struct RaycastHit { float distance; float3 position, normal; int triangleIndex; sge_surface_material material; };
struct CapsuleCastHit { float toi; float3 position, normal, triangleNormal; int triangleIndex; sge_surface_material material; };
struct CapsuleOverlapHit { float depth; float3 position, normal, triangleNormal; int triangleIndex; sge_surface_material material; };

class CollisionQuery {
public:
    // init(world:activeEntityIDs:) — builds both triangle sets + BVHs from the collidable entities, partitioned the way
    // StaticTriMesh.partitionEntities does: bodies that are not .static go to the dynamic set (:886-900)
    CollisionQuery(World& world, const std::vector<sge_static_mesh_entity>& staticCollidables,
                   const std::vector<sge_static_mesh_entity>& dynamicCollidables = {}) : world_(world) {
        check(sge_collision_rebuild_static(world.context(), staticCollidables.data(), (int32_t)staticCollidables.size()), "sge_collision_rebuild_static");
        check(sge_collision_rebuild_dynamic(world.context(), dynamicCollidables.data(), (int32_t)dynamicCollidables.size()), "sge_collision_rebuild_dynamic");
    }
    // updateStaticTransforms / updateDynamicTransforms(world:entities:activeEntityIDs:) — `entities` index the arrays the
    // sets were built from; modelMatrices = their TransformComponent.modelMatrix, column-major
    void updateStaticTransforms(const std::vector<int32_t>& entities, const std::vector<std::array<float, 16>>& modelMatrices) {
        update(SGE_SET_STATIC, entities, modelMatrices);
    }
    void updateDynamicTransforms(const std::vector<int32_t>& entities, const std::vector<std::array<float, 16>>& modelMatrices) {
        update(SGE_SET_DYNAMIC, entities, modelMatrices);
    }
    std::optional<RaycastHit> raycast(float3 origin, float3 direction, float maxDistance, uint32_t mask = 0xFFFFFFFFu) {
        sge_ray_query q{{origin.x, origin.y, origin.z}, {direction.x, direction.y, direction.z}, maxDistance, mask};
        sge_raycast_hit h{};
        check(sge_raycast_batch(world_.context(), &q, 1, &h), "sge_raycast_batch");
        if (!h.hit) return std::nullopt;
        return RaycastHit{h.distance, {h.position[0], h.position[1], h.position[2]}, {h.normal[0], h.normal[1], h.normal[2]}, h.triangleIndex, h.material};
    }
    std::optional<CapsuleCastHit> capsuleCast(float3 from, float3 delta, float radius, float halfHeight, uint32_t mask = 0xFFFFFFFFu) {
        return cast(from, delta, radius, halfHeight, SGE_CAST, 0.0f, mask);
    }
    std::optional<CapsuleCastHit> capsuleCastBlocking(float3 from, float3 delta, float radius, float halfHeight, uint32_t mask = 0xFFFFFFFFu) {
        return cast(from, delta, radius, halfHeight, SGE_CAST_BLOCKING, 0.0f, mask);
    }
    std::optional<CapsuleCastHit> capsuleCastGround(float3 from, float3 delta, float radius, float halfHeight, float minNormalY, uint32_t mask = 0xFFFFFFFFu) {
        return cast(from, delta, radius, halfHeight, SGE_CAST_GROUND, minNormalY, mask);
    }
    std::optional<CapsuleOverlapHit> capsuleOverlap(float3 from, float radius, float halfHeight, uint32_t mask = 0xFFFFFFFFu) {
        sge_capsule_query q{{from.x, from.y, from.z}, {0, 0, 0}, radius, halfHeight, 0.0f, mask, SGE_CAST};
        sge_capsule_overlap_hit h{};
        int32_t found = 0;
        check(sge_capsule_overlap_batch(world_.context(), &q, 1, &h, &found), "sge_capsule_overlap_batch");
        if (!found) return std::nullopt;
        return CapsuleOverlapHit{h.depth, {h.position[0], h.position[1], h.position[2]}, {h.normal[0], h.normal[1], h.normal[2]},
                                 {h.triangleNormal[0], h.triangleNormal[1], h.triangleNormal[2]}, h.triangleIndex, h.material};
    }
    std::vector<CapsuleOverlapHit> capsuleOverlapAll(float3 from, float radius, float halfHeight, int maxHits = 8, uint32_t mask = 0xFFFFFFFFu) {
        maxHits = maxHits < 1 ? 1 : (maxHits > SGE_MAX_OVERLAP_HITS ? SGE_MAX_OVERLAP_HITS : maxHits); // max(1, maxHits), :157
        sge_capsule_query q{{from.x, from.y, from.z}, {0, 0, 0}, radius, halfHeight, 0.0f, mask, SGE_CAST};
        std::array<sge_capsule_overlap_hit, SGE_MAX_OVERLAP_HITS> out{};
        int32_t n = 0;
        check(sge_capsule_overlap_all_batch(world_.context(), &q, 1, maxHits, out.data(), &n), "sge_capsule_overlap_all_batch");
        std::vector<CapsuleOverlapHit> hits;
        for (int k = 0; k < n; ++k)
            hits.push_back({out[k].depth, {out[k].position[0], out[k].position[1], out[k].position[2]},
                            {out[k].normal[0], out[k].normal[1], out[k].normal[2]},
                            {out[k].triangleNormal[0], out[k].triangleNormal[1], out[k].triangleNormal[2]},
                            out[k].triangleIndex, out[k].material});
        return hits;
    }
    // batched forms for callers that have many probes per frame
    void capsuleCastBatch(const std::vector<sge_capsule_query>& q, std::vector<sge_capsule_cast_hit>& out) {
        out.resize(q.size());
        check(sge_capsule_cast_batch(world_.context(), q.data(), (int32_t)q.size(), out.data()), "sge_capsule_cast_batch");
    }

private:
    void update(int32_t set, const std::vector<int32_t>& entities, const std::vector<std::array<float, 16>>& m) {
        if (entities.size() != m.size()) throw std::invalid_argument("one model matrix per entity");
        check(sge_collision_update_transforms(world_.context(), set, entities.data(), m.empty() ? nullptr : m[0].data(), (int32_t)entities.size()),
              "sge_collision_update_transforms");
    }
    std::optional<CapsuleCastHit> cast(float3 from, float3 delta, float radius, float halfHeight, uint32_t mode, float minNormalY, uint32_t mask) {
        sge_capsule_query q{{from.x, from.y, from.z}, {delta.x, delta.y, delta.z}, radius, halfHeight, minNormalY, mask, mode};
        sge_capsule_cast_hit h{};
        check(sge_capsule_cast_batch(world_.context(), &q, 1, &h), "sge_capsule_cast_batch");
        if (!h.hit) return std::nullopt; // Swift nil
        return CapsuleCastHit{h.toi, {h.position[0], h.position[1], h.position[2]}, {h.normal[0], h.normal[1], h.normal[2]},
                              {h.triangleNormal[0], h.triangleNormal[1], h.triangleNormal[2]}, h.triangleIndex, h.material};
    }
    World& world_;
};

// ---- RTSkinningEncoder (RTSkinningEncoder.swift:10-57) ----------------------------------------------
using RTSkinningJob = sge_skinning_job; // sourcePositions ... paletteBuffer as device pointers, vertexCount, dstBaseVertex

class RTSkinningEncoder {
public:
    // init?(device:) — nil when the kernel is unavailable (here: no context)
    static std::optional<RTSkinningEncoder> make(World* world) {
        if (!world) return std::nullopt;
        return RTSkinningEncoder(*world);
    }
    // encode(commandBuffer:outputBuffer:outputNormalBuffer:outputTangentBuffer:jobs:) — asynchronous on the
    // context's stream (the command buffer); returns without work when jobs is empty (:32-35)
    void encode(void* outputBuffer, void* outputNormalBuffer, void* outputTangentBuffer, const std::vector<RTSkinningJob>& jobs,
                int outLayout = SGE_LAYOUT_PADDED16) {
        if (jobs.empty()) return;
        check(sge_skinning_encode(world_->context(), outputBuffer, outputNormalBuffer, outputTangentBuffer, outLayout, jobs.data(),
                                  (int32_t)jobs.size()), "sge_skinning_encode");
    }
    // the crowd path: one job per character over the shared source mesh, palettes from the pose stage
    void encodeCrowd(float dt = 0.0f) { world_->tick(dt, SGE_STAGE_SKIN); }
    // What the reference gets from enqueueing on the same command buffer (RayTracingScene.swift:35-43): `consumer` (a hipStream_t;
    // nullptr = the context's stream) sees the skinned streams of every encode / tick so far. Needed under SGE_OPT_OVERLAP_SKIN,
    // where the skin stage runs on the context's second stream; harmless otherwise.
    void waitForSkinning(void* consumer = nullptr) { check(sge_skin_wait(world_->context(), consumer), "sge_skin_wait"); }
    // ... and the reverse: what `consumer` holds so far completes before the next skin launch overwrites the streams
    void skinningConsumed(void* consumer = nullptr) { check(sge_skin_consumed(world_->context(), consumer), "sge_skin_consumed"); }

private:
    explicit RTSkinningEncoder(World& w) : world_(&w) {}
    World* world_;
};

// ---- RTAccelerationBuilder, dynamic slices (RTAccelerationBuilder.swift:75-145) --------------------------
// The skinned items' primitive acceleration structures: built once (usage .refit), refitted every frame from the
// skinned vertex buffer. Static slices and the TLAS stay with the renderer.
using RTRay = sge_blas_ray;
using RTHit = sge_blas_hit;

class RTAccelerationBuilder {
public:
    explicit RTAccelerationBuilder(World& w) : world_(&w) {}
    // state.dynamicChanged branch (:75-112): encoder.build over the item's slice of the dynamic index buffer
    void buildDynamic(const std::vector<uint32_t>& indices) {
        check(sge_blas_build(world_->context(), indices.data(), (int32_t)indices.size()), "sge_blas_build");
    }
    // the item's uvs in the dynamic UV buffer (RTGeometryCache.swift:277-283); hits then carry interp_uv
    void setUVs(const std::vector<float>& uvs) { check(sge_blas_set_uvs(world_->context(), uvs.data(), (int32_t)(uvs.size() / 2)), "sge_blas_set_uvs"); }
    // steady-state branch (:113-145): encoder.refit(... options: .vertexData) per dynamic slice; asynchronous
    void refit(int first, int count) { check(sge_blas_refit(world_->context(), first, count), "sge_blas_refit"); }
    // instance descriptors (:168-185): transformationMatrix = item.modelMatrix, column-major 4x4 per character
    void setInstances(int first, int count, const float* modelMatrices) {
        check(sge_blas_instances_upload(world_->context(), first, count, modelMatrices), "sge_blas_instances_upload");
    }
    // isect.intersect(ray, accel) against one instance + the kernel's reads at the hit (RayTracing.metalinc:242-300)
    std::vector<RTHit> intersect(const std::vector<RTRay>& rays) {
        std::vector<RTHit> hits(rays.size());
        check(sge_blas_intersect_batch(world_->context(), rays.data(), (int32_t)rays.size(), hits.data()), "sge_blas_intersect_batch");
        return hits;
    }
    sge_blas_info info() const {
        sge_blas_info i{};
        check(sge_blas_info_get(world_->context(), &i), "sge_blas_info_get");
        return i;
    }

private:
    World* world_;
};

// ---- World <-> GPU bridge (C++ twin of host/swift/GPUCrowd.swift) -----------------------------------------
// The reference keeps every component in a per-type dictionary, `world.store(T.self)[e]` (World.swift:64-75), and its systems copy
// value-type components in and out per entity per step (Systems.swift:1802-1821 writeBack; :279-407; :475-517; :2249-2267). A host
// that swaps the GPU path in keeps those stores: other systems (input, camera, gameplay) read them. The components below carry the
// fields of Components.swift that cross the boundary; EntityWorld is the stores; GPUCrowd moves them to and from the context with
// the pinned, event-ordered calls (sge_state_*), once per fixed step, no host synchronisation of the context.
using Entity = uint32_t; // Entity.id (World.swift:10-13)
enum class BodyType { Static = SGE_BODY_STATIC, Kinematic = SGE_BODY_KINEMATIC, Dynamic = SGE_BODY_DYNAMIC };
struct PhysicsBodyComponent {        // Components.swift:549-598
    double position[3] = {0, 0, 0}, linearVelocity[3] = {0, 0, 0};
    float rotation[4] = {0, 0, 0, 1};
    BodyType bodyType = BodyType::Dynamic;
};
struct TransformComponent {          // Components.swift:14-45
    float translation[3] = {0, 0, 0};
    float rotation[4] = {0, 0, 0, 1};
};
struct CharacterControllerComponent { // Components.swift:353-431 (defaults :380-404, CharacterFactory.swift:88-91)
    float radius = 1.5f, halfHeight = 1.0f, skinWidth = 0.3f, groundSnapSkin = 0.05f, snapDistance = 0.8f, fallProbeDistance = 200.0f;
    float groundSnapMaxSpeed = 8.0f, groundSnapMaxToi = 0.2f, groundSnapMaxStep = 0.1f, groundSweepMaxStep = 0.1f;
    int maxSlideIterations = 4;
    float minGroundDot = 0.5f;
    uint32_t collisionMask = 0xFFFFFFFFu;
    bool grounded = false, groundedNear = false, groundSliding = false;
    float groundNormal[3] = {0, 1, 0};
    int groundTriangleIndex = -1;
    float groundDistance = 3.402823466e38f;
    float sideContactNormal[3] = {0, 0, 0};
    int sideContactFrames = 0;
    std::vector<int> contactManifoldTriangles;
    std::vector<std::array<float, 3>> contactManifoldNormals;
    int contactManifoldFrames = 0, groundTransitionFrames = 0;
};
struct AgentCollisionComponent { float massWeight = 1.0f; bool isSolid = true; std::optional<float> radiusOverride; }; // :433-445
struct MoveIntentComponent { float desiredVelocity[3] = {0, 0, 0}; float desiredFacingYaw = 0; bool hasFacingYaw = false; };     // :600-618
struct MovementComponent { float maxAcceleration = 20.0f, maxDeceleration = 36.0f; };                                            // :684-702
struct LocomotionProfileComponent {  // Components.swift:203-293; profiles are rows of World::uploadMotionProfiles
    int idleProfile = 0, walkProfile = 0, runProfile = 0, fallProfile = 0;
    float idleTime = 0, walkTime = 0, runTime = 0, fallTime = 0;
    float idleEnterSpeed = 0.15f, idleExitSpeed = 0.3f, runEnterSpeed = 6.0f, runExitSpeed = 5.0f, fallMinDropHeight = 0.5f;
    float blendTime = 0.2f, blendT = 1.0f, idleInertiaHalfLife = 0.18f, idleInertia = 0.0f;
    int fromState = SGE_LOCO_IDLE, state = SGE_LOCO_IDLE;
    bool isBlending = false;
};
struct MotionProfileComponent { int profile = 0; float time = 0, playbackRate = 1.0f; bool loop = true, inPlace = true; };
struct ActionAnimationComponent {    // Components.swift:620-653
    int profile = 0;
    float time = 0, playbackRate = 1.0f, weight = 0, blendInTime = 0.1f, blendOutHalfLife = 0.1f;
    bool active = false, loop = false, inPlace = true, exiting = false;
};
template <class T> using Store = std::unordered_map<Entity, T>;
struct EntityWorld {                 // World.store(T.self), World.swift:64-75
    Store<PhysicsBodyComponent> bodies;
    Store<TransformComponent> transforms;
    Store<CharacterControllerComponent> controllers;
    Store<AgentCollisionComponent> agents;
    Store<MoveIntentComponent> intents;
    Store<MovementComponent> movements;
    Store<LocomotionProfileComponent> locomotion;
    Store<MotionProfileComponent> motion;
    Store<ActionAnimationComponent> actions;
};

class GPUCrowd {
public:
    explicit GPUCrowd(World& w) : world_(w) {}
    // index in the GPU arrays = position here: sorted by entity id, the canonical order (World.query returns Dictionary order)
    const std::vector<Entity>& entities() const { return entities_; }

    // Structural change (characters created / destroyed): entity <-> index tables + every component, once.
    void rebuild(const EntityWorld& ew) {
        entities_.clear();
        for (const auto& kv : ew.bodies) if (ew.controllers.count(kv.first)) entities_.push_back(kv.first);
        std::sort(entities_.begin(), entities_.end());
        indexOf_.clear();
        for (size_t i = 0; i < entities_.size(); ++i) indexOf_[entities_[i]] = (int32_t)i;
        const int32_t n = (int32_t)entities_.size();
        std::vector<sge_body_state> b(n);
        std::vector<sge_controller_params> p(n);
        std::vector<sge_controller_state> c(n);
        std::vector<sge_move_intent> in(n);
        std::vector<sge_locomotion_state> l(n);
        std::vector<sge_action_state> a(n);
        for (int32_t i = 0; i < n; ++i) {
            const Entity e = entities_[i];
            encodeBody(ew, e, b[i]);
            encodeController(ew, e, p[i], c[i]);
            in[i] = encodeIntent(ew, e);
            encodeAnimation(ew, e, l[i], a[i]);
        }
        world_.resize(n);
        if (n) world_.upload(0, n, b.data(), p.data(), c.data(), in.data(), l.data(), a.data());
        pending_ = -1;
    }

    // Before the step: what other systems wrote into the World since the last one. Intents go every step (PhysicsIntentSystem's
    // input); bodies / controllers only for entities their writers flagged (teleports, jump / dodge edits of the velocity). Pinned
    // staging, copies enqueued in front of the next tick, no host synchronisation.
    void pushDirtyState(const EntityWorld& ew, const std::unordered_set<Entity>& dirtyBodies = {}) {
        const int32_t n = (int32_t)entities_.size();
        if (n == 0) return;
        sge_state_view v{};
        check(sge_state_push_begin(world_.context(), SGE_STATE_INTENTS, 0, n, &v), "sge_state_push_begin");
        for (int32_t i = 0; i < n; ++i) v.intents[i] = encodeIntent(ew, entities_[i]);
        check(sge_state_push_commit(world_.context()), "sge_state_push_commit");
        for (Entity e : dirtyBodies) {
            auto it = indexOf_.find(e);
            if (it == indexOf_.end()) continue;
            check(sge_state_push_begin(world_.context(), SGE_STATE_BODIES | SGE_STATE_CONTROLLERS, it->second, 1, &v), "sge_state_push_begin");
            sge_controller_params unused{};
            encodeBody(ew, e, v.bodies[0]);
            encodeController(ew, e, unused, v.controllers[0]);
            check(sge_state_push_commit(world_.context()), "sge_state_push_commit");
        }
    }

    // Right behind the step's sge_tick: the snapshot of this step starts its way to pinned host memory. Returns at once.
    void beginPull(uint32_t which = SGE_STATE_WORLD) {
        if (entities_.empty()) return;
        check(sge_state_pull_async(world_.context(), which, 0, 0, &pending_), "sge_state_pull_async");
    }
    // After the step: what the reference's systems would have written into the World (KinematicMoveStopSystem.writeBack
    // Systems.swift:1802-1821, LocomotionProfileSystem :279-407, ActionAnimationSystem :475-517, PhysicsWritebackSystem :2249-2267).
    // Waits for the pull begun last (its copy only: not the skin launch, not later ticks) and decodes it.
    void pullBack(EntityWorld& ew) {
        if (pending_ < 0) return;
        sge_state_view v{};
        check(sge_state_wait(world_.context(), pending_, &v), "sge_state_wait");
        pending_ = -1;
        for (int32_t i = 0; i < v.count; ++i) {
            const Entity e = entities_[(size_t)(v.first + i)];
            if (v.bodies) {
                const sge_body_state& b = v.bodies[i];
                auto pb = ew.bodies.find(e);
                if (pb != ew.bodies.end()) {
                    std::memcpy(pb->second.position, b.position, sizeof(b.position));
                    std::memcpy(pb->second.linearVelocity, b.linearVelocity, sizeof(b.linearVelocity));
                    std::memcpy(pb->second.rotation, b.rotation, sizeof(b.rotation));
                }
                auto pt = ew.transforms.find(e); // PhysicsWritebackSystem: TransformComponent from the body
                if (pt != ew.transforms.end()) {
                    for (int k = 0; k < 3; ++k) pt->second.translation[k] = (float)b.position[k];
                    std::memcpy(pt->second.rotation, b.transformRotation, sizeof(b.transformRotation));
                }
            }
            if (v.controllers) {
                auto pc = ew.controllers.find(e);
                if (pc != ew.controllers.end()) decodeController(v.controllers[i], pc->second);
            }
            if (v.locomotion) {
                const sge_locomotion_state& l = v.locomotion[i];
                auto pl = ew.locomotion.find(e);
                if (pl != ew.locomotion.end()) {
                    LocomotionProfileComponent& c = pl->second;
                    c.idleTime = l.time[0]; c.walkTime = l.time[1]; c.runTime = l.time[2]; c.fallTime = l.time[3];
                    c.blendT = l.blendT; c.idleInertia = l.idleInertia; c.fromState = l.fromState; c.state = l.state;
                    c.isBlending = (l.flags & SGE_LOCO_IS_BLENDING) != 0;
                }
                auto pm = ew.motion.find(e);
                if (pm != ew.motion.end()) pm->second.time = l.motionTime;
            }
            if (v.actions) {
                auto pa = ew.actions.find(e);
                if (pa != ew.actions.end()) {
                    pa->second.time = v.actions[i].time; pa->second.weight = v.actions[i].weight;
                    pa->second.active = (v.actions[i].flags & SGE_ACTION_ACTIVE) != 0;
                    pa->second.exiting = (v.actions[i].flags & SGE_ACTION_EXITING) != 0;
                }
            }
        }
    }
    std::optional<int32_t> index(Entity e) const {
        auto it = indexOf_.find(e);
        if (it == indexOf_.end()) return std::nullopt;
        return it->second;
    }

private:
    static void encodeBody(const EntityWorld& ew, Entity e, sge_body_state& b) {
        const PhysicsBodyComponent& body = ew.bodies.at(e);
        b = sge_body_state{};
        std::memcpy(b.position, body.position, sizeof(b.position));
        std::memcpy(b.linearVelocity, body.linearVelocity, sizeof(b.linearVelocity));
        std::memcpy(b.rotation, body.rotation, sizeof(b.rotation));
        auto t = ew.transforms.find(e); // what PoseStackSystem reads one step stale (ProceduralPoseSystem.swift:345)
        std::memcpy(b.transformRotation, t != ew.transforms.end() ? t->second.rotation : body.rotation, sizeof(b.transformRotation));
        b.bodyType = (uint32_t)body.bodyType;
    }
    static void encodeController(const EntityWorld& ew, Entity e, sge_controller_params& p, sge_controller_state& s) {
        const CharacterControllerComponent& c = ew.controllers.at(e);
        p = sge_controller_params{};
        p.radius = c.radius; p.halfHeight = c.halfHeight; p.skinWidth = c.skinWidth; p.groundSnapSkin = c.groundSnapSkin;
        p.snapDistance = c.snapDistance; p.fallProbeDistance = c.fallProbeDistance; p.groundSnapMaxSpeed = c.groundSnapMaxSpeed;
        p.groundSnapMaxToi = c.groundSnapMaxToi; p.groundSnapMaxStep = c.groundSnapMaxStep; p.groundSweepMaxStep = c.groundSweepMaxStep;
        p.maxSlideIterations = c.maxSlideIterations; p.minGroundDot = c.minGroundDot; p.collisionMask = c.collisionMask;
        auto ag = ew.agents.find(e);
        if (ag != ew.agents.end()) {
            p.agentFlags = SGE_AGENT_PRESENT | bit(ag->second.isSolid, SGE_AGENT_SOLID) | bit(ag->second.radiusOverride.has_value(), SGE_AGENT_RADIUS_OVERRIDE);
            p.agentRadiusOverride = ag->second.radiusOverride.value_or(0.0f);
            p.agentMassWeight = ag->second.massWeight;
        } else p.agentMassWeight = 1.0f;
        s = sge_controller_state{};
        std::memcpy(s.groundNormal, c.groundNormal, sizeof(s.groundNormal));
        s.groundTriangleIndex = c.groundTriangleIndex;
        std::memcpy(s.sideContactNormal, c.sideContactNormal, sizeof(s.sideContactNormal));
        s.sideContactFrames = c.sideContactFrames;
        const size_t m = std::min({c.contactManifoldTriangles.size(), c.contactManifoldNormals.size(), (size_t)SGE_MANIFOLD_MAX});
        for (size_t k = 0; k < m; ++k) {
            s.manifoldTriangles[k] = c.contactManifoldTriangles[k];
            for (int a = 0; a < 3; ++a) s.manifoldNormals[k][a] = c.contactManifoldNormals[k][(size_t)a];
        }
        s.manifoldCount = (int32_t)m; s.manifoldFrames = c.contactManifoldFrames; s.groundTransitionFrames = c.groundTransitionFrames;
        s.flags = bit(c.grounded, SGE_CTRL_GROUNDED) | bit(c.groundedNear, SGE_CTRL_GROUNDED_NEAR) | bit(c.groundSliding, SGE_CTRL_GROUND_SLIDING);
        s.groundDistance = c.groundDistance;
    }
    static void decodeController(const sge_controller_state& s, CharacterControllerComponent& c) {
        std::memcpy(c.groundNormal, s.groundNormal, sizeof(s.groundNormal));
        c.groundTriangleIndex = s.groundTriangleIndex;
        std::memcpy(c.sideContactNormal, s.sideContactNormal, sizeof(s.sideContactNormal));
        c.sideContactFrames = s.sideContactFrames;
        const int m = std::max(0, std::min(s.manifoldCount, (int32_t)SGE_MANIFOLD_MAX));
        c.contactManifoldTriangles.assign(s.manifoldTriangles, s.manifoldTriangles + m);
        c.contactManifoldNormals.resize((size_t)m);
        for (int k = 0; k < m; ++k) c.contactManifoldNormals[(size_t)k] = {s.manifoldNormals[k][0], s.manifoldNormals[k][1], s.manifoldNormals[k][2]};
        c.contactManifoldFrames = s.manifoldFrames; c.groundTransitionFrames = s.groundTransitionFrames;
        c.grounded = (s.flags & SGE_CTRL_GROUNDED) != 0; c.groundedNear = (s.flags & SGE_CTRL_GROUNDED_NEAR) != 0;
        c.groundSliding = (s.flags & SGE_CTRL_GROUND_SLIDING) != 0;
        c.groundDistance = s.groundDistance;
    }
    // MoveIntentComponent + MovementComponent as PhysicsIntentSystem consumes them (Systems.swift:205-250)
    static sge_move_intent encodeIntent(const EntityWorld& ew, Entity e) {
        sge_move_intent out{};
        auto it = ew.intents.find(e);
        if (it == ew.intents.end()) return out;
        std::memcpy(out.desiredVelocity, it->second.desiredVelocity, sizeof(out.desiredVelocity));
        out.desiredFacingYaw = it->second.desiredFacingYaw;
        out.flags = SGE_INTENT_PRESENT | bit(it->second.hasFacingYaw, SGE_INTENT_HAS_FACING_YAW);
        auto mv = ew.movements.find(e);
        if (mv != ew.movements.end()) { out.maxAcceleration = mv->second.maxAcceleration; out.maxDeceleration = mv->second.maxDeceleration; }
        return out;
    }
    static void encodeAnimation(const EntityWorld& ew, Entity e, sge_locomotion_state& l, sge_action_state& a) {
        l = sge_locomotion_state{};
        a = sge_action_state{};
        auto lc = ew.locomotion.find(e);
        if (lc != ew.locomotion.end()) {
            const LocomotionProfileComponent& c = lc->second;
            l.profile[0] = c.idleProfile; l.profile[1] = c.walkProfile; l.profile[2] = c.runProfile; l.profile[3] = c.fallProfile;
            l.time[0] = c.idleTime; l.time[1] = c.walkTime; l.time[2] = c.runTime; l.time[3] = c.fallTime;
            l.idleEnterSpeed = c.idleEnterSpeed; l.idleExitSpeed = c.idleExitSpeed; l.runEnterSpeed = c.runEnterSpeed; l.runExitSpeed = c.runExitSpeed;
            l.fallMinDropHeight = c.fallMinDropHeight; l.blendTime = c.blendTime; l.blendT = c.blendT;
            l.idleInertiaHalfLife = c.idleInertiaHalfLife; l.idleInertia = c.idleInertia; l.fromState = c.fromState; l.state = c.state;
            l.flags |= SGE_LOCO_PRESENT | bit(c.isBlending, SGE_LOCO_IS_BLENDING);
        }
        auto mp = ew.motion.find(e);
        if (mp != ew.motion.end()) {
            l.flags |= SGE_MOTION_PRESENT | bit(mp->second.loop, SGE_MOTION_LOOP) | bit(mp->second.inPlace, SGE_MOTION_IN_PLACE);
            l.motionTime = mp->second.time; l.playbackRate = mp->second.playbackRate; l.motionProfile = mp->second.profile;
        }
        auto ac = ew.actions.find(e);
        if (ac != ew.actions.end()) {
            const ActionAnimationComponent& c = ac->second;
            a.profile = c.profile; a.time = c.time; a.playbackRate = c.playbackRate; a.weight = c.weight;
            a.blendInTime = c.blendInTime; a.blendOutHalfLife = c.blendOutHalfLife;
            a.flags = SGE_ACTION_PRESENT | bit(c.active, SGE_ACTION_ACTIVE) | bit(c.loop, SGE_ACTION_LOOP) |
                      bit(c.inPlace, SGE_ACTION_IN_PLACE) | bit(c.exiting, SGE_ACTION_EXITING);
        }
    }
    World& world_;
    std::vector<Entity> entities_;
    std::unordered_map<Entity, int32_t> indexOf_;
    int32_t pending_ = -1;
};

// All character stages of one fixed step as ONE tick, in the order of DemoScene.swift:57-75, with the World kept in step: put it
// where physicsIntentSystem stands. `lagged`: the World holds step n - 1 when fixedUpdate(n) returns and nothing is waited for
// (tick n + 1 is enqueued while pull n is still on its way); otherwise it holds step n (the host waits for move(n) + pose(n) + the
// copy, never for skin(n)).
class GPUCharacterStepSystem {
public:
    GPUCharacterStepSystem(World& world, GPUCrowd& crowd, EntityWorld& stores, uint32_t stages = SGE_STAGE_ALL, bool lagged = false)
        : world_(world), crowd_(crowd), stores_(stores), stages_(stages), lagged_(lagged) {
        check(sge_context_set_option(world.context(), SGE_OPT_OVERLAP_SKIN, 1), "sge_context_set_option");
    }
    float3 gravity{0, -98.0f, 0};
    void fixedUpdate(float dt, const std::unordered_set<Entity>& dirtyBodies = {}) {
        crowd_.pushDirtyState(stores_, dirtyBodies);
        world_.tick(dt, stages_, gravity);
        if (lagged_) crowd_.pullBack(stores_); // the previous step's, long landed
        crowd_.beginPull();
        if (!lagged_) crowd_.pullBack(stores_);
    }
    void finish() { crowd_.pullBack(stores_); }
private:
    World& world_;
    GPUCrowd& crowd_;
    EntityWorld& stores_;
    uint32_t stages_;
    bool lagged_;
};

} // namespace sge
