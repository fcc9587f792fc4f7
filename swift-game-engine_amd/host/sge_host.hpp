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
//
// `World` here is the crowd owned by one GPU context: the reference's per-entity dictionary stores
// (World.swift:64-75) become the context's resident arrays, so `fixedUpdate(world, dt)` is one batched call.
#pragma once
#include <array>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/sge_amd.h"

namespace sge {

struct float3 { float x, y, z; };

class Error : public std::runtime_error {
public:
    explicit Error(const std::string& what) : std::runtime_error(what + ": " + sge_last_error()) {}
};
inline void check(int rc, const char* what) { if (rc != SGE_OK) throw Error(what); }

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
                       (policy_ == ContactCachePolicy::SideContactOnly ? SGE_STAGE_SIDE_CONTACT_CACHE : 0u), gravity_);
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

} // namespace sge
