// ORACLE — TEST INFRASTRUCTURE ONLY (see sge_oracle_math.h for the rules).
//
// Internal declarations of the CPU restatement. The POD layouts come from the
// product's PUBLIC header so tests can feed both sides the same bytes; nothing
// in the product depends on oracle/.
#pragma once
#include <vector>
#include <cstdint>
#include "../include/sge_amd.h"
#include "sge_oracle_math.h"

namespace sgeo {

// ---- Skeleton (Game/Skeleton.swift:127-173) ----
struct Skeleton {
    int boneCount = 0;
    std::vector<int> parent;
    std::vector<M4> bindLocal, invBindModel;
    std::vector<V3> restTranslation, rawRestTranslation, preRotationDegrees;
    M4 rootRotationFix = m4_identity();
    float unitScale = 1.0f;
    int pelvisIndex = -1, leanIndex = -1;
};

// ---- MotionProfile flattened per skeleton bone (Game/Animation.swift:11-53) ----
struct MotionProfile {
    int order = 0;
    float cycleDurationRaw = 0; // phase?.cycleDuration ?? duration
    std::vector<uint8_t> bonePresent;   // [B]
    std::vector<uint8_t> coeffCount;    // [B][6]
    std::vector<float> coeffs;          // [B][6][SGE_MAX_COEFFS]
};

struct SkinnedMesh {
    int vertexCount = 0;
    std::vector<float> positions, normals, tangents, weights;
    std::vector<uint16_t> indices;
    std::vector<M4> invBindModel; // optional
};

// ---- Collision (Game/CollisionQuery.swift) ----
struct AABB { V3 min, max; };
struct BVHNode { AABB bounds; int left, right, start, count, parent; };
struct BVH {
    std::vector<BVHNode> nodes;
    std::vector<int> triOrder, triLeaf;
    int root = -1;
    void build(const std::vector<AABB>& triangleAABBs);
    void refit(const std::vector<int>& updatedTriangles, const std::vector<AABB>& triangleAABBs); // :528-575
};
struct MeshSlice { int vertexBegin = 0, vertexEnd = 0, indexBegin = 0, indexEnd = 0, triBegin = 0, triEnd = 0; bool valid = false; }; // :480-485
struct TriangleMeshSet {
    std::vector<V3> positions;
    std::vector<uint32_t> indices;
    std::vector<AABB> triangleAABBs;
    std::vector<sge_surface_material> triangleMaterials;
    std::vector<uint32_t> triangleLayers;
    BVH bvh;
    bool hasBVH = false;
    std::vector<MeshSlice> slices;   // by entity index of the rebuild call (the reference keys them by Entity)
    std::vector<V3> localPositions;  // collisionMesh.streams.positions of every entity, same indexing as `positions`
    void rebuild(const sge_static_mesh_entity* ents, int count);
    // updateTransforms (:419-462): new model matrices for `n` entities of the last rebuild; returns the updated triangles
    std::vector<int> updateTransforms(const int32_t* entities, const float* modelMatrices, int n);
};
struct QueryStats { long long candidates = 0, sweeps = 0, iterations = 0, queries = 0; };

struct CapsuleCastHit { float toi; V3 position, normal, triangleNormal; int triangleIndex; sge_surface_material material; };
struct CapsuleOverlapHit { float depth; V3 position, normal, triangleNormal; int triangleIndex; sge_surface_material material; };

// Per-thread counters (the reference's CollisionQueryStats side channel); merged by the tick driver.
QueryStats& thread_stats();
QueryStats take_thread_stats();

struct RaycastHit { float distance; V3 position, normal; int triangleIndex; sge_surface_material material; };

struct CollisionQuery {
    TriangleMeshSet staticSet, dynamicSet; // StaticTriMesh.staticSet / dynamicSet (:710-711)
    bool raycast(V3 origin, V3 direction, float maxDistance, uint32_t mask, RaycastHit& out) const; // :768-785
    bool capsuleCastCombined(V3 from, V3 delta, float radius, float halfHeight, bool blockingOnly,
                             bool hasMinNormalY, float minNormalY, uint32_t mask, CapsuleCastHit& out) const;
    int capsuleOverlapAll(V3 from, float radius, float halfHeight, int maxHits, uint32_t mask,
                          CapsuleOverlapHit* out) const;
    bool capsuleOverlap(V3 from, float radius, float halfHeight, uint32_t mask, CapsuleOverlapHit& out) const;
};

struct AgentSweepState { int entity; V3 position, velocity; float radius, halfHeight; };

struct World {
    Skeleton skeleton;
    std::vector<MotionProfile> profiles;
    SkinnedMesh mesh;
    CollisionQuery query;
    // characters (AoS per component, same PODs as the product ABI)
    std::vector<sge_body_state> bodies;
    std::vector<sge_controller_params> params;
    std::vector<sge_controller_state> controllers;
    std::vector<sge_move_intent> intents;
    std::vector<sge_locomotion_state> locomotion;
    std::vector<sge_action_state> actions;
    // PoseComponent
    std::vector<M4> local, model, palette; // [N][B]
    // skinned outputs, packed
    std::vector<float> outPositions, outNormals, outTangents;
    // kinematic platforms of this step (PlatformCarry.computeDelta inputs, Systems.swift:644-732)
    std::vector<sge_platform_state> platforms;
    // the skinned item's slice of dynamicIndexBuffer + per-character instance matrices (RTAccelerationBuilder.swift:75-185)
    std::vector<uint32_t> blasIndices;
    std::vector<float> blasInstances; // [N][16], missing rows = identity
    std::vector<float> blasUVs;       // [V][2] or empty
    // agents (start-of-step snapshot over ALL ranks' characters)
    std::vector<sge_agent_state> importedAgents;
    int agentSelfOffset = 0;
    bool agentsImported = false;
    // AgentSeparationSystem.init defaults (Systems.swift:2146-2152)
    int separationIterations = 2;
    float separationMargin = 0.2f, separationHeightMargin = 0.1f;
    // contactCachePolicy of the move system for the current tick (SGE_STAGE_SIDE_CONTACT_CACHE): SideContactOnlyCachePolicy
    bool sideContactCacheOnly = false;
};

// pose.cpp
void pose_fixed_update(World& w, int first, int count, float dt);
void locomotion_fixed_update(World& w, int first, int count);
void action_fixed_update(World& w, int first, int count, float dt);
// skin.cpp
void skin_characters(World& w, int first, int count);
void skinning_kernel(int vertexCount, const float* pos, const float* nrm, const float* tan,
                     const uint16_t* idx, const float* wts, const M4* palette,
                     float* outPos, float* outNrm, float* outTan, int dstBaseVertex);
// blas.cpp: scan over every triangle of the index buffer (RayTracing.metalinc:242-296)
void blas_intersect(const World& w, const sge_blas_ray& ray, sge_blas_hit& hit);
// move.cpp
void intent_fixed_update(World& w, int first, int count, float dt);
void gravity_fixed_update(World& w, int first, int count, float dt, V3 gravity);
void collect_agent_states(const World& w, std::vector<AgentSweepState>& agents, int& selfOffset);
void kinematic_move_fixed_update(World& w, int first, int count, float dt, V3 gravity,
                                 const std::vector<AgentSweepState>* agents, int selfOffset);
void writeback_fixed_update(World& w, int first, int count);
// AgentSeparationSystem (Systems.swift:1906-2210) over the whole crowd, canonical order = character index
void agent_separation_fixed_update(World& w, int iterations, float separationMargin, float heightMargin);

} // namespace sgeo
