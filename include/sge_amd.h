/*
 * sge_amd.h — C ABI of the MI355X (gfx950) character-update path.
 *
 * This is the drop-in boundary for ONE hot path of kelian343/swift-game-engine:
 * MotionProfile pose evaluation + bone palette, 4-weight linear-blend skinning,
 * and capsule-CCD move-and-slide against the static and dynamic triangle sets of
 * CollisionQuery (kinematic-platform carry included), followed by what the
 * renderer does next with the skinned vertices: the per-frame refit of the skinned
 * items' acceleration structures and the ray-hit reads (last section).  The reference
 * has no FFI layer; each entry point below names the Swift surface it replaces
 * (paths relative to the reference checkout).  A Swift host binds these through
 * a module map; tests and bench.py bind them with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes; no C++/torch types; all structs are POD with the
 *     exact layouts below (static_asserted in the implementation);
 *   - matrices are column-major float[16] (simd float4x4 memory order);
 *     quaternions are float[4] = (ix, iy, iz, r) (simd_quatf memory order);
 *   - every function returns SGE_OK (0) or an SGE_ERR_* code; create returns NULL
 *     on failure, mirroring the reference's `init?` convention
 *     (Game/RTSkinningEncoder.swift:14);
 *   - host pointers unless a parameter is named d_* (device pointer);
 *   - work is enqueued on the context's HIP stream and completes asynchronously,
 *     like the reference's MTLCommandBuffer encode (RTSkinningEncoder.swift:27);
 *     call sge_synchronize() (or a *_download, which synchronises) to wait;
 *   - one context per GPU, one caller thread at a time (the reference runs on the
 *     main actor, Game.xcodeproj/project.pbxproj:420).
 *
 * There is NO CPU fallback: without a visible gfx950 device sge_context_create
 * fails and every other call needs a context.
 */
#ifndef SGE_AMD_H
#define SGE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: SGE_OPT_OVERLAP_SKIN takes effect on a caller-provided stream only with the value 2 (1 keeps version 1's contract there: the
 *    option is ignored and work enqueued behind sge_tick sees the skinned streams); sge_skin_wait / sge_skin_consumed /
 *    sge_crowd_palette_buffers, SGE_STAGE_SIDE_CONTACT_CACHE; the asynchronous World write-back (sge_state_*). */
#define SGE_ABI_VERSION 2

typedef struct sge_context sge_context;

enum {
    SGE_OK = 0,
    SGE_ERR_INVALID = 1,   /* bad argument / shape mismatch */
    SGE_ERR_DEVICE = 2,    /* HIP error (message via sge_last_error) */
    SGE_ERR_STATE = 3,     /* required upload missing */
    SGE_ERR_CAPACITY = 4,  /* a fixed device-side capacity was exceeded */
    SGE_ERR_NOT_READY = 5  /* sge_state_poll: the pull has not landed yet (not an error) */
};

#define SGE_MAX_BONES 256
#define SGE_MAX_PROFILES 16
#define SGE_MAX_FOURIER_ORDER 8
#define SGE_MAX_COEFFS (1 + 2 * SGE_MAX_FOURIER_ORDER)
#define SGE_MAX_OVERLAP_HITS 8
#define SGE_MAX_PLATFORMS 64 /* kinematic platforms per step (the demo scene has two) */
#define SGE_MANIFOLD_MAX 4 /* ContactManifoldCache.maxCount, Game/Systems.swift:1160 */

/* ------------------------------------------------------------------------- */
/* Data contracts (Game/Components.swift)                                     */
/* ------------------------------------------------------------------------- */

/* SurfaceMaterial, Components.swift:704-716 */
typedef struct sge_surface_material {
    float muS;
    float muK;
    uint32_t flattenGround;
} sge_surface_material;

/* BodyType, Components.swift:543-547 */
enum { SGE_BODY_STATIC = 0, SGE_BODY_KINEMATIC = 1, SGE_BODY_DYNAMIC = 2 };

/* PhysicsBodyComponent (Components.swift:549-598) plus the entity's
 * TransformComponent.rotation as last written by PhysicsWritebackSystem
 * (Systems.swift:2249-2267), which PoseStackSystem reads one step stale
 * (ProceduralPoseSystem.swift:345). 96 bytes. */
typedef struct sge_body_state {
    double position[3];
    double linearVelocity[3];
    float rotation[4];
    float transformRotation[4];
    uint32_t bodyType;
    uint32_t _pad[3];
} sge_body_state;

enum {
    SGE_AGENT_PRESENT = 1u << 0,        /* entity has an AgentCollisionComponent */
    SGE_AGENT_SOLID = 1u << 1,          /* .isSolid */
    SGE_AGENT_RADIUS_OVERRIDE = 1u << 2 /* .radiusOverride != nil */
};

/* The constant part of CharacterControllerComponent (Components.swift:353-431)
 * and AgentCollisionComponent (Components.swift:433-445). 64 bytes. */
typedef struct sge_controller_params {
    float radius;
    float halfHeight;
    float skinWidth;
    float groundSnapSkin;
    float snapDistance;
    float fallProbeDistance;
    float groundSnapMaxSpeed;
    float groundSnapMaxToi;
    float groundSnapMaxStep;
    float groundSweepMaxStep;
    int32_t maxSlideIterations;
    float minGroundDot;
    uint32_t collisionMask;
    uint32_t agentFlags;
    float agentRadiusOverride;
    float agentMassWeight;
} sge_controller_params;

enum {
    SGE_CTRL_GROUNDED = 1u << 0,
    SGE_CTRL_GROUNDED_NEAR = 1u << 1,
    SGE_CTRL_GROUND_SLIDING = 1u << 2
};

/* The per-step mutable part of CharacterControllerComponent. 128 bytes. */
typedef struct sge_controller_state {
    float groundNormal[3];
    int32_t groundTriangleIndex;
    float sideContactNormal[3];
    int32_t sideContactFrames;
    int32_t manifoldTriangles[SGE_MANIFOLD_MAX];
    float manifoldNormals[SGE_MANIFOLD_MAX][3];
    int32_t manifoldCount;
    int32_t manifoldFrames;
    int32_t groundTransitionFrames;
    uint32_t flags; /* SGE_CTRL_* */
    float groundDistance;
    uint32_t _pad[3];
} sge_controller_state;

enum {
    SGE_INTENT_PRESENT = 1u << 0,     /* entity has a MoveIntentComponent */
    SGE_INTENT_HAS_FACING_YAW = 1u << 1,
    SGE_INTENT_DODGE_ACTIVE = 1u << 2 /* DodgeActionComponent.active */
};

/* MoveIntentComponent + MovementComponent (Components.swift:600-618, 684-702)
 * as consumed by PhysicsIntentSystem (Systems.swift:205-250). 32 bytes. */
typedef struct sge_move_intent {
    float desiredVelocity[3];
    float desiredFacingYaw;
    uint32_t flags; /* SGE_INTENT_* */
    float maxAcceleration;
    float maxDeceleration;
    uint32_t _pad;
} sge_move_intent;

/* LocomotionState, Components.swift:223-228 */
enum { SGE_LOCO_IDLE = 0, SGE_LOCO_WALK = 1, SGE_LOCO_RUN = 2, SGE_LOCO_FALLING = 3 };

enum {
    SGE_LOCO_IS_BLENDING = 1u << 0,
    SGE_LOCO_PRESENT = 1u << 1,       /* entity has a LocomotionProfileComponent */
    SGE_MOTION_PRESENT = 1u << 2,     /* entity has a MotionProfileComponent */
    SGE_MOTION_LOOP = 1u << 3,
    SGE_MOTION_IN_PLACE = 1u << 4
};

/* LocomotionProfileComponent + MotionProfileComponent
 * (Components.swift:203-293). Profiles are indices into the table uploaded by
 * sge_motion_profiles_upload. 96 bytes. */
typedef struct sge_locomotion_state {
    int32_t profile[4]; /* idle, walk, run, fall */
    float time[4];      /* idleTime, walkTime, runTime, fallTime */
    float idleEnterSpeed;
    float idleExitSpeed;
    float runEnterSpeed;
    float runExitSpeed;
    float fallMinDropHeight;
    float blendTime;
    float blendT;
    float idleInertiaHalfLife;
    float idleInertia;
    int32_t fromState;
    int32_t state;
    uint32_t flags; /* SGE_LOCO_* | SGE_MOTION_* */
    float motionTime;     /* MotionProfileComponent.time */
    float playbackRate;   /* MotionProfileComponent.playbackRate */
    int32_t motionProfile; /* MotionProfileComponent.profile */
    float posePhase;      /* out: PoseComponent.phase */
} sge_locomotion_state;

enum {
    SGE_ACTION_PRESENT = 1u << 0,
    SGE_ACTION_ACTIVE = 1u << 1,
    SGE_ACTION_LOOP = 1u << 2,
    SGE_ACTION_IN_PLACE = 1u << 3,
    SGE_ACTION_EXITING = 1u << 4,
    SGE_ACTION_HAS_DODGE = 1u << 5 /* entity has a DodgeActionComponent */
};

/* ActionAnimationComponent (Components.swift:620-653); dodgeEnd is
 * `dodge.endTime > 0 ? dodge.endTime : dodge.duration` (Systems.swift:488). 32 bytes. */
typedef struct sge_action_state {
    int32_t profile;
    float time;
    float playbackRate;
    float weight;
    float blendInTime;
    float blendOutHalfLife;
    float dodgeEnd;
    uint32_t flags; /* SGE_ACTION_* */
} sge_action_state;

/* ------------------------------------------------------------------------- */
/* Context                                                                    */
/* ------------------------------------------------------------------------- */

/* Creates the per-GPU context (device memory owner, HIP stream, kernels).
 * Returns NULL when no gfx950 device `device_index` exists. */
sge_context* sge_context_create(int device_index);
void sge_context_destroy(sge_context* ctx);
/* Last error text of this thread ("" if none). */
const char* sge_last_error(void);
int sge_abi_version(void);
/* Use the caller's hipStream_t (e.g. torch's current stream) instead of the
 * context-owned one. NULL restores the owned stream. */
int sge_context_set_stream(sge_context* ctx, void* hip_stream);
/* The hipStream_t the context currently enqueues on (its own, or the caller's), so that a host can order its own work — an RCCL
 * collective, a renderer's pass — behind sge_* calls without a host synchronisation. */
int sge_context_get_stream(sge_context* ctx, void** hip_stream);
/* Blocks until everything enqueued on the context's stream has completed. */
int sge_synchronize(sge_context* ctx);

enum {
    SGE_OPT_STORE_POSE_DEBUG = 1, /* also keep PoseComponent.local/.model per character */
    SGE_OPT_SKIN_LAYOUT = 2,      /* SGE_LAYOUT_* for the skinned output streams */
    SGE_OPT_PROFILE = 3,          /* 1: bracket every kernel with HIP events */
    SGE_OPT_HEAVY_THRESHOLD = 5,  /* distance evaluations in a character's previous step above which its slide / ground
                                   * pass runs in the 8-wavefront kernel (default 4000; 0: every character that swept
                                   * anything; < 0: always the one-wave kernel). Scheduling only: results are identical. */
    SGE_OPT_PLACEMENT_PROBES = 6, /* how many candidate placements of the skinned output streams are timed when they are
                                   * (re)allocated; the fastest is kept (default 8, stops early at 6.5 TB/s; <= 1: take the
                                   * first). Takes effect at the next sge_characters_resize / layout change. */
    SGE_OPT_FUSE_BLAS_REFIT = 7,  /* 1 (default): a tick with both SGE_STAGE_SKIN and SGE_STAGE_BLAS_REFIT folds the refit into the LBS
                                   * kernel (the boxes are reduced from the positions while they are still on chip), except
                                   * under SGE_OPT_OVERLAP_SKIN, where the fused kernel would keep the next step's collision
                                   * kernels off the chip; 2: always; 0: two launches, the refit reads the positions back */
    SGE_OPT_OVERLAP_SKIN = 4      /* 1 or 2: the skin stage of sge_tick runs on the context's second stream, so that skin(n) overlaps
                                   * move(n+1) + pose(n+1) (the benchmarked schedule). On the context's own stream 1 and 2 mean the
                                   * same. On a caller-provided stream (sge_context_set_stream) the value 1 is IGNORED, as in ABI
                                   * version 1: stages run back to back on the caller's stream, whatever the caller enqueues behind
                                   * sge_tick sees the skinned streams, and the palette pointer of sge_crowd_buffers does not move.
                                   * The value 2 is the explicit opt-in there: the skin launch is ordered behind the pose stage on
                                   * the caller's stream by events, and what is enqueued on the caller's stream behind sge_tick is
                                   * then NOT ordered behind the skin launch by itself: a consumer of the skinned streams calls
                                   * sge_skin_wait first and sge_skin_consumed after its last read, and queries the palette pointer
                                   * again after every tick (sge_blas_refit*, sge_skinning_encode, the downloads and
                                   * sge_synchronize join by themselves).
                                   * A whole-crowd tick with move + pose + skin stages also runs its POSE launch on a stream of
                                   * the context's own, beside the next tick's move stage: palettes, locomotion / action states and
                                   * transformRotation of tick n are complete where its skin launch is — sge_skin_wait orders a
                                   * consumer behind both, sge_characters_upload / _download join by themselves. */
};
enum {
    SGE_LAYOUT_PACKED = 0,  /* positions/normals float[3] (12 B), tangents float[4] */
    SGE_LAYOUT_PADDED16 = 1 /* Metal `device float3*` stride: 16 B per element */
};
int sge_context_set_option(sge_context* ctx, int option, int value);

/* ------------------------------------------------------------------------- */
/* Skeleton — Game/Skeleton.swift, Game/SkeletonLoader.swift                  */
/* ------------------------------------------------------------------------- */

/* Host helper: SkeletonLoader.buildSkeleton (SkeletonLoader.swift:28-87) and
 * Skeleton.init's invBindModel (Skeleton.swift:155-156).  Outputs are
 * caller-allocated: restTranslation [B][3], bindLocal [B][16],
 * invBindModel [B][16], rootRotationFix [16]. zero_root = resolved RootRule. */
int sge_skeleton_build(int32_t bone_count, const int32_t* parent,
                       const float* raw_translations, const float* pre_rotation_degrees,
                       const float root_fix_degrees[3], float unit_scale, int zero_root,
                       float* rest_translation, float* bind_local, float* inv_bind_model,
                       float* root_rotation_fix);

typedef struct sge_skeleton_desc {
    int32_t boneCount;
    const int32_t* parent;           /* [B], parent index < child index, root = -1 */
    const float* bindLocal;          /* [B][16] */
    const float* invBindModel;       /* [B][16] */
    const float* restTranslation;    /* [B][3] */
    const float* rawRestTranslation; /* [B][3] */
    const float* preRotationDegrees; /* [B][3] */
    float rootRotationFix[16];
    float unitScale;
    int32_t pelvisIndex; /* skeleton.semantic(.pelvis) or -1 */
    int32_t leanIndex;   /* chest ?? spine3 ?? spine2 ?? spine1 or -1 (ProceduralPoseSystem.swift:371-374) */
} sge_skeleton_desc;

/* Replaces constructing `Skeleton` for the GPU path (Skeleton.swift:127-173). */
int sge_skeleton_upload(sge_context* ctx, const sge_skeleton_desc* desc);

/* ------------------------------------------------------------------------- */
/* Motion profiles — Game/Animation.swift                                     */
/* ------------------------------------------------------------------------- */

#define SGE_AXIS_ABSENT 255

/* One MotionProfile (Animation.swift:11-53) flattened per skeleton bone.
 * Axis order: translation x,y,z then rotation x,y,z. */
typedef struct sge_motion_profile_desc {
    int32_t order;       /* MotionProfile.order */
    float cycleDuration; /* phase?.cycleDuration ?? duration */
    const uint8_t* bonePresent; /* [B]: 1 when profile.bones[skeleton.names[i]] exists */
    const uint8_t* coeffCount;  /* [B][6]: array length, SGE_AXIS_ABSENT when the axis is nil */
    const float* coeffs;        /* [B][6][SGE_MAX_COEFFS] = [a0,a1,b1,...] */
} sge_motion_profile_desc;

/* Replaces MotionProfileLoader.load + the per-bone dictionary lookups of
 * ProceduralPoseSystem.swift:153-154 with a dense device table. */
int sge_motion_profiles_upload(sge_context* ctx, const sge_motion_profile_desc* profiles, int32_t count);

/* ------------------------------------------------------------------------- */
/* Skinned mesh + skinning — Game/RTSkinningEncoder.swift,                    */
/* Game/RTGeometryCache.swift:43-52,492-576, Game/RayTracing.metalinc:732-776 */
/* ------------------------------------------------------------------------- */

/* Host helper: MeshTangents.compute (MeshTangents.swift:10-83). Exactly one of
 * indices16 / indices32 is non-NULL. tangents out [V][4]. */
int sge_mesh_tangents_compute(int32_t vertex_count, const float* positions, const float* normals,
                              const float* uvs, const uint16_t* indices16, const uint32_t* indices32,
                              int32_t index_count, float* tangents);

typedef struct sge_skinned_mesh_desc {
    int32_t vertexCount;
    const float* positions;      /* [V][3] */
    const float* normals;        /* [V][3] */
    const float* tangents;       /* [V][4] */
    const uint16_t* boneIndices; /* [V][4] */
    const float* boneWeights;    /* [V][4] */
    const float* invBindModel;   /* optional [invBindCount][16]: SkinnedMeshDescriptor.invBindModel */
    int32_t invBindCount;
} sge_skinned_mesh_desc;

/* Uploads the crowd's shared source mesh once (the reference caches source
 * buffers per unique mesh, RTGeometryCache.swift:499-554). When invBindModel is
 * given with invBindCount == boneCount, the pose stage emits
 * model[i] * mesh.invBindModel[i] (Systems.swift:2519-2527) as the palette. */
int sge_skinned_mesh_upload(sge_context* ctx, const sge_skinned_mesh_desc* desc);

/* RTSkinningJob (RTGeometryCache.swift:43-52) with MTLBuffers as device pointers. */
typedef struct sge_skinning_job {
    const void* d_sourcePositions;   /* layout per sourceLayout */
    const void* d_sourceNormals;
    const void* d_sourceTangents;    /* float4 */
    const void* d_sourceBoneIndices; /* ushort4 */
    const void* d_sourceBoneWeights; /* float4 */
    const void* d_palette;           /* float4x4[paletteCount] */
    int32_t paletteCount;
    int32_t vertexCount;
    int32_t dstBaseVertex;
    int32_t sourceLayout; /* SGE_LAYOUT_* of positions/normals */
} sge_skinning_job;

/* RTSkinningEncoder.encode (RTSkinningEncoder.swift:27-56): one skinningKernel
 * dispatch per job into the shared output streams. Asynchronous. */
int sge_skinning_encode(sge_context* ctx, void* d_outPositions, void* d_outNormals,
                        void* d_outTangents, int32_t out_layout,
                        const sge_skinning_job* jobs, int32_t job_count);

/* Device-pointer accessors for the context-owned crowd buffers (for building
 * sge_skinning_job lists or handing the streams to a downstream consumer).
 * d_palettes is the buffer the NEWEST pose stage wrote. Without SGE_OPT_OVERLAP_SKIN it never changes. With it the palettes
 * alternate between two buffers (pose(n+1) writes one while skin(n) still reads the other): the pointer is valid until the next
 * whole-crowd pose stage, so query it again after every tick (RTGeometryCache.makeSkinningJob builds a fresh palette buffer per
 * job per frame as well, RTGeometryCache.swift:556-566), or take both with sge_crowd_palette_buffers.
 * The output streams never move between ticks; under SGE_OPT_OVERLAP_SKIN a consumer reads them after sge_skin_wait. */
int sge_crowd_buffers(sge_context* ctx, void** d_palettes, void** d_outPositions,
                      void** d_outNormals, void** d_outTangents);
/* Both palette buffers ([2]; they do not move until sge_characters_resize) and the index of the one holding the newest pose. */
int sge_crowd_palette_buffers(sge_context* ctx, void** d_palettes, int32_t* latest);
/* Orders `consumer_stream` (a hipStream_t; NULL: the context's own current stream) behind every skin launch and every other kernel
 * the context has enqueued so far, without a host synchronisation: hipStreamWaitEvent on the newest skin launch's event and on the
 * newest pose launch's, if that ran on the pose stream (+ a marker on the main stream when the consumer is a different stream). The equivalent of enqueueing behind
 * RTSkinningEncoder.encode on the same MTLCommandBuffer (RTSkinningEncoder.swift:27-56; consumed at RayTracingScene.swift:35-43). */
int sge_skin_wait(sge_context* ctx, void* consumer_stream);
/* The reverse ordering, for a consumer that reads the skinned streams asynchronously on a stream of its own: everything
 * `consumer_stream` holds so far completes before the context's NEXT skin / refit launch overwrites the streams (an event on the
 * consumer's stream that the context's main, skin and pose streams wait for: the pose launch two ticks on rewrites the palette
 * buffer the consumer's jobs may name). Not needed when the consumer runs on the context's stream in serial order,
 * or synchronises with the host before the next sge_tick. One command buffer per frame gives the reference both orderings
 * (Renderer.swift:159, 224). */
int sge_skin_consumed(sge_context* ctx, void* consumer_stream);
int sge_skinned_mesh_buffers(sge_context* ctx, void** d_positions, void** d_normals,
                             void** d_tangents, void** d_boneIndices, void** d_boneWeights);

/* ------------------------------------------------------------------------- */
/* Collision world — Game/CollisionQuery.swift                                */
/* ------------------------------------------------------------------------- */

/* One collidable StaticMeshComponent (Components.swift:323-351) with its
 * TransformComponent.modelMatrix; mesh = collisionMesh ?? mesh
 * (CollisionQuery.swift:344). */
typedef struct sge_static_mesh_entity {
    const float* positions; /* [vertexCount][3], mesh-local */
    int32_t vertexCount;
    const uint32_t* indices;
    int32_t indexCount;
    float modelMatrix[16];
    sge_surface_material material;
    const sge_surface_material* triangleMaterials; /* optional, used when triangleMaterialCount == indexCount/3 */
    int32_t triangleMaterialCount;
    uint32_t collisionLayer;
} sge_static_mesh_entity;

/* CollisionQuery.init / TriangleMeshSet.rebuild + BVH.build
 * (CollisionQuery.swift:331-417, 577-670) for the static set. */
int sge_collision_rebuild_static(sge_context* ctx, const sge_static_mesh_entity* entities, int32_t count);

/* The dynamic triangle set: collidable StaticMeshComponents whose PhysicsBodyComponent is not .static
 * (StaticTriMesh.partitionEntities, CollisionQuery.swift:886-900; rebuildDynamic :744-751). Every query
 * consults the static set first, then this one; its triangle indices are reported offset by the static
 * set's triangle count (:796-801 and the other combined queries). count = 0 empties the set. */
int sge_collision_rebuild_dynamic(sge_context* ctx, const sge_static_mesh_entity* entities, int32_t count);

enum { SGE_SET_STATIC = 0, SGE_SET_DYNAMIC = 1 };
/* CollisionQuery.updateStaticTransforms / updateDynamicTransforms (CollisionQuery.swift:69-83):
 * TriangleMeshSet.updateTransforms (:419-462) re-poses the listed entities of the last rebuild of `set`
 * (entity_indices index that call's array; entities that kept no triangle are skipped) with new
 * TransformComponent.modelMatrix values [n][16], recomputes their triangle AABBs and refits the BVH
 * (BVH.refit :528-575) — topology, triangle order and the area filter of the rebuild are kept. */
int sge_collision_update_transforms(sge_context* ctx, int32_t set, const int32_t* entity_indices,
                                    const float* model_matrices, int32_t n);

/* Introspection for parity tests: sizes, then copies of the host-side build. */
typedef struct sge_bvh_node {
    float boundsMin[3];
    float boundsMax[3];
    int32_t left, right, start, count, parent;
} sge_bvh_node;
int sge_collision_counts(sge_context* ctx, int32_t* vertex_count, int32_t* triangle_count, int32_t* node_count);
int sge_collision_copy(sge_context* ctx, float* positions, uint32_t* indices, float* triangle_aabbs,
                       sge_bvh_node* nodes, int32_t* tri_order, int32_t* tri_leaf);
/* the same two calls for either set (the two above = SGE_SET_STATIC) */
int sge_collision_counts_set(sge_context* ctx, int32_t set, int32_t* vertex_count, int32_t* triangle_count, int32_t* node_count);
int sge_collision_copy_set(sge_context* ctx, int32_t set, float* positions, uint32_t* indices, float* triangle_aabbs,
                           sge_bvh_node* nodes, int32_t* tri_order, int32_t* tri_leaf);

enum { SGE_CAST = 0, SGE_CAST_BLOCKING = 1, SGE_CAST_GROUND = 2 };

typedef struct sge_capsule_query {
    float from[3];
    float delta[3]; /* ignored by overlap queries */
    float radius;
    float halfHeight;
    float minNormalY; /* SGE_CAST_GROUND only */
    uint32_t mask;
    uint32_t mode; /* SGE_CAST* */
} sge_capsule_query;

/* CapsuleCastHit, CollisionQuery.swift:36-43 */
typedef struct sge_capsule_cast_hit {
    int32_t hit; /* 0 = nil */
    float toi;
    float position[3];
    float normal[3];
    float triangleNormal[3];
    int32_t triangleIndex;
    sge_surface_material material;
} sge_capsule_cast_hit;

/* CapsuleOverlapHit, CollisionQuery.swift:45-52 */
typedef struct sge_capsule_overlap_hit {
    float depth;
    float position[3];
    float normal[3];
    float triangleNormal[3];
    int32_t triangleIndex;
    sge_surface_material material;
} sge_capsule_overlap_hit;

/* CollisionQuery.capsuleCast / capsuleCastBlocking / capsuleCastGround
 * (CollisionQuery.swift:96-135), batched: out[i] answers queries[i]. Synchronous. */
int sge_capsule_cast_batch(sge_context* ctx, const sge_capsule_query* queries, int32_t count,
                           sge_capsule_cast_hit* out);
/* CollisionQuery.capsuleOverlapAll (CollisionQuery.swift:148-159), batched:
 * out[i*max_hits + k], k < out_counts[i]; max_hits in 1..SGE_MAX_OVERLAP_HITS. */
int sge_capsule_overlap_all_batch(sge_context* ctx, const sge_capsule_query* queries, int32_t count,
                                  int32_t max_hits, sge_capsule_overlap_hit* out, int32_t* out_counts);

/* CollisionQuery.capsuleOverlap (CollisionQuery.swift:137-146, 830-850, 1119-1199), batched: the deepest
 * overlapping triangle (first in visit order among equal depths); out_found[i] = 0 means nil. Synchronous. */
int sge_capsule_overlap_batch(sge_context* ctx, const sge_capsule_query* queries, int32_t count,
                              sge_capsule_overlap_hit* out, int32_t* out_found);

/* CollisionQuery.raycast(origin:direction:maxDistance:mask:) (CollisionQuery.swift:85-94, 768-785, 916-978) */
typedef struct sge_ray_query {
    float origin[3];
    float direction[3]; /* used as given (the reference does not normalise it) */
    float maxDistance;
    uint32_t mask;
} sge_ray_query;
/* RaycastHit, CollisionQuery.swift:28-34 */
typedef struct sge_raycast_hit {
    int32_t hit; /* 0 = nil */
    float distance;
    float position[3];
    float normal[3];
    int32_t triangleIndex;
    sge_surface_material material;
} sge_raycast_hit;
int sge_raycast_batch(sge_context* ctx, const sge_ray_query* queries, int32_t count, sge_raycast_hit* out);

/* Kinematic platforms — the inputs PlatformCarry.computeDelta (Systems.swift:644-732) reads per platform entity
 * (PhysicsBodyComponent + TransformComponent + StaticMeshComponent + KinematicPlatformComponent, :1832-1835), in
 * entity order: the body type, pDelta = positionF - prevPositionF, and meshWorldAABB(collisionMesh ?? mesh, transform)
 * (:627-642), which the reference recomputes per character and this ABI takes once per step. */
typedef struct sge_platform_state {
    float aabbMin[3];
    float aabbMax[3];
    float delta[3];
    uint32_t kinematic; /* bodyType == .kinematic */
    uint32_t hasAABB;   /* 0 when the mesh has no positions (meshWorldAABB returned nil) */
    uint32_t _pad;
} sge_platform_state;
/* meshWorldAABB (Systems.swift:627-642): host helper. Returns SGE_OK and writes min/max, or SGE_ERR_INVALID for an
 * empty mesh. */
int sge_mesh_world_aabb(const float* positions, int32_t vertex_count, const float model_matrix[16],
                        float out_min[3], float out_max[3]);
/* The platform list of the coming ticks (count = 0: none, PlatformCarry returns .zero at :651). */
int sge_platforms_upload(sge_context* ctx, const sge_platform_state* platforms, int32_t count);

/* ------------------------------------------------------------------------- */
/* Characters + the batched fixed step                                        */
/* ------------------------------------------------------------------------- */

/* Sets the crowd size on this GPU; (re)allocates state, palettes and the
 * skinned output streams (count * vertexCount vertices). */
int sge_characters_resize(sge_context* ctx, int32_t count);
/* Any pointer may be NULL (= leave / skip). Arrays are [count]. */
int sge_characters_upload(sge_context* ctx, int32_t first, int32_t count,
                          const sge_body_state* bodies, const sge_controller_params* params,
                          const sge_controller_state* controllers, const sge_move_intent* intents,
                          const sge_locomotion_state* locomotion, const sge_action_state* actions);
int sge_characters_download(sge_context* ctx, int32_t first, int32_t count,
                            sge_body_state* bodies, sge_controller_params* params,
                            sge_controller_state* controllers, sge_move_intent* intents,
                            sge_locomotion_state* locomotion, sge_action_state* actions);
/* PoseComponent.palette / .model / .local (Components.swift:188-201) as
 * [count][boneCount][16]; model/local need SGE_OPT_STORE_POSE_DEBUG. */
int sge_palettes_download(sge_context* ctx, int32_t first, int32_t count,
                          float* palette, float* model, float* local);
/* Skinned output streams for vertices [first_vertex, first_vertex+count) of the
 * crowd buffer, always returned packed: positions/normals [n][3], tangents [n][4]. */
int sge_skinned_download(sge_context* ctx, int64_t first_vertex, int64_t vertex_count,
                         float* positions, float* normals, float* tangents);

/* ---- Asynchronous World synchronisation ------------------------------------------------------------------------------------
 * The reference's systems read and write their components in process memory: KinematicMoveStopSystem.writeBack
 * (Systems.swift:1802-1821), the locomotion / action clocks (:279-407, :475-517) and PhysicsWritebackSystem (:2249-2267) store into
 * `world.store(T.self)[e]` (World.swift:64-75) every fixed step, and PhysicsIntentSystem reads MoveIntentComponent from there
 * (:205-250). A host that keeps its World has to see every step's results and feed every step's intents; sge_characters_download /
 * _upload do that with pageable memory and a host synchronisation of the whole context. The calls below are the pinned,
 * event-ordered form: nothing here waits for the skin launch of the step, and the next tick may be enqueued while a pull is in
 * flight (the arrays are first copied on the device, behind the kernels that wrote them and in front of the next step's, then
 * moved to pinned host memory on a stream of their own). */
enum {
    SGE_STATE_BODIES = 1u << 0,      /* sge_body_state */
    SGE_STATE_CONTROLLERS = 1u << 1, /* sge_controller_state */
    SGE_STATE_LOCOMOTION = 1u << 2,  /* sge_locomotion_state */
    SGE_STATE_ACTIONS = 1u << 3,     /* sge_action_state */
    SGE_STATE_INTENTS = 1u << 4,     /* sge_move_intent (push only: nothing on the device writes it) */
    SGE_STATE_WORLD = 0xFu           /* what GPUCrowd.pullBack decodes into the World */
};
/* Host pointers into context-owned pinned memory, [count] each, NULL for arrays that were not asked for. */
typedef struct sge_state_view {
    int32_t first, count;
    uint32_t which;
    int32_t ticket;
    sge_body_state* bodies;
    sge_controller_state* controllers;
    sge_locomotion_state* locomotion;
    sge_action_state* actions;
    sge_move_intent* intents;
} sge_state_view;
/* Snapshot of the selected arrays of characters [first, first + count) (count 0 = all) as they stand behind everything enqueued so
 * far — call it right after sge_tick(n) for "the World after step n": bodies (less transformRotation) and controllers as the move /
 * separation stage left them, locomotion / action states and transformRotation as the pose stage left them, wherever those stages ran.
 * Returns at once; *ticket names the pull. Two pulls may be in flight; the third reuses the first one's memory (and waits for it). */
int sge_state_pull_async(sge_context* ctx, uint32_t which, int32_t first, int32_t count, int32_t* ticket);
/* Blocks the host until pull `ticket` has landed (its copy only: not the skin launch, not later ticks) and fills `view`. The
 * pointers stay valid until the pull after next is enqueued. SGE_ERR_STATE for a ticket that has been overwritten. */
int sge_state_wait(sge_context* ctx, int32_t ticket, sge_state_view* view);
/* SGE_OK when pull `ticket` has landed, SGE_ERR_NOT_READY while it is in flight. */
int sge_state_poll(sge_context* ctx, int32_t ticket);
/* The other direction: pinned staging for what the host's systems wrote since the last step (intents every step; bodies /
 * controllers / locomotion / actions of teleported or edited characters). begin() hands out host arrays [count] for characters
 * [first, first + count) of the arrays in `which`; the caller fills them; commit() enqueues the copies on the context's stream in
 * front of the next tick and returns at once. One begin / commit pair at a time; two stagings alternate. */
int sge_state_push_begin(sge_context* ctx, uint32_t which, int32_t first, int32_t count, sge_state_view* staging);
int sge_state_push_commit(sge_context* ctx);

enum {
    SGE_STAGE_INTENT = 1u << 0,     /* PhysicsIntentSystem, Systems.swift:205-250 */
    SGE_STAGE_GRAVITY = 1u << 1,    /* GravitySystem, Systems.swift:596-620 */
    SGE_STAGE_MOVE = 1u << 2,       /* KinematicMoveStopSystem, Systems.swift:1823-1902 */
    SGE_STAGE_LOCOMOTION = 1u << 3, /* LocomotionProfileSystem, Systems.swift:279-407 */
    SGE_STAGE_ACTION = 1u << 4,     /* ActionAnimationSystem, Systems.swift:475-517 */
    SGE_STAGE_POSE = 1u << 5,       /* PoseStackSystem, ProceduralPoseSystem.swift:13-406 */
    SGE_STAGE_WRITEBACK = 1u << 6,  /* PhysicsWritebackSystem (rotation), Systems.swift:2249-2267 */
    SGE_STAGE_SKIN = 1u << 7,       /* RTSkinningEncoder.encode over the crowd */
    SGE_STAGE_AGENTS = 1u << 8,     /* capsule-capsule sweep vs the imported agent set, Systems.swift:1053-1091 */
    SGE_STAGE_BLAS_REFIT = 1u << 9, /* RTAccelerationBuilder dynamic-slice refit, RTAccelerationBuilder.swift:113-145 (not in SGE_STAGE_ALL) */
    SGE_STAGE_SEPARATION = 1u << 10, /* AgentSeparationSystem, Systems.swift:1906-2210: between the move stage and the locomotion stage,
                                      * as in DemoScene.swift:66-68 (not in SGE_STAGE_ALL; whole crowd only: first = 0, count = all).
                                      * With the crowd sharded over several contexts (configs[3] / [4]: sge_agents_import or
                                      * sge_agents_allgather named agents of other contexts) the stage returns SGE_ERR_STATE: the
                                      * pair loop is sequential over the whole crowd in index order, a shard cannot reproduce it and
                                      * a halo exchange would define another result. Gather the crowd into one context for it. */
    SGE_STAGE_SIDE_CONTACT_CACHE = 1u << 11, /* modifier of SGE_STAGE_MOVE, not a stage: the system's contactCachePolicy
                                      * (KinematicMoveStopSystem.init(gravity:contactCachePolicy:), Systems.swift:1402-1415) is
                                      * SideContactOnlyCachePolicy (:1136-1157: the depenetration pass records only side contacts)
                                      * instead of DefaultContactCachePolicy (:1102-1134). Any other ContactCachePolicy is host
                                      * code the kernels cannot run: the Swift / C++ mirrors refuse it. */
    SGE_STAGE_ALL_FIXED = 0x7Fu,
    SGE_STAGE_ALL = 0xFFu
};

typedef struct sge_tick_desc {
    float dt;         /* fixed step, TimeComponent.fixedDelta = 1/60 (Components.swift:526) */
    float gravity[3]; /* (0,-98,0), Systems.swift:599 */
    uint32_t stages;  /* SGE_STAGE_* in the reference's order (DemoScene.swift:57-75) */
    int32_t first;    /* character range; count 0 = all */
    int32_t count;
    uint32_t _pad;
} sge_tick_desc;

/* One fixed step of the hot path for the character range. Asynchronous. */
int sge_tick(sge_context* ctx, const sge_tick_desc* desc);

/* AgentSweepState (Systems.swift:1023-1029), 32 bytes: the start-of-step
 * snapshot collectAgentStates builds (Systems.swift:1592-1611). */
typedef struct sge_agent_state {
    float position[3];
    float radius;
    float velocity[3];
    float halfHeight;
} sge_agent_state;

/* Packs this GPU's characters into d_out[count] (device), non-solid or
 * agent-less characters get radius < 0. Asynchronous. */
int sge_agents_export(sge_context* ctx, void* d_out);
/* Declares the all-gathered agent set (device pointer, total entries) and the
 * index of this GPU's first character inside it; binned on device into an XZ
 * grid each step when SGE_STAGE_AGENTS is set. */
int sge_agents_import(sge_context* ctx, const void* d_all, int32_t total, int32_t self_offset);
/* The whole exchange of the character-vs-character config (collectAgentStates Systems.swift:1592-1611 taken on every GPU)
 * in one call, stream-ordered on the context's stream, no host synchronisation:
 *     sge_agents_export into the context's own buffer -> ncclAllGather(comm) -> sge_agents_import of the gathered set.
 * `nccl_comm` is the caller's ncclComm_t (one rank per GPU; RCCL is resolved at run time from librccl.so, this library does not link
 * it); rank / world_size are that communicator's; slot = max over ranks of their character counts (every rank contributes `slot`
 * 32-byte records, padded with radius < 0). Call between the tick with SGE_STAGE_INTENT|GRAVITY and the tick with the remaining stages
 * + SGE_STAGE_AGENTS, as the snapshot is taken after gravity (Systems.swift:1837-1841). world_size 1 needs no communicator (NULL). */
int sge_agents_allgather(sge_context* ctx, void* nccl_comm, int32_t rank, int32_t world_size, int32_t slot);

/* Accumulated HIP-event kernel time since the last reset (SGE_OPT_PROFILE). */
typedef struct sge_stage_times {
    double move_ms, pose_ms, skin_ms, agents_ms;
    int64_t move_launches, pose_launches, skin_launches, agents_launches;
} sge_stage_times;
int sge_profile_read(sge_context* ctx, sge_stage_times* out, int reset);

/* Device-side counters of the move stage since the last reset (diagnostics; the
 * reference's CollisionQueryStats, CollisionQuery.swift:280-290). */
typedef struct sge_move_stats {
    uint64_t queries;          /* BVH queries issued */
    uint64_t candidates;       /* capsuleCandidateCount */
    uint64_t sweepIterations;  /* capsuleSweepIterations */
    uint64_t overflow;         /* traversal-stack or candidate overflows (must stay 0) */
    uint64_t traversalSteps;   /* wave-wide node-expansion steps */
    uint64_t sweepTrips;       /* wave-wide trips of the sweep loop (one distance evaluation per active lane each) */
    uint64_t prunedPairs;      /* (cast, triangle) pairs of vertical casts skipped by the conservative XZ reject (no result changes) */
} sge_move_stats;
/* AgentSeparationSystem.init(iterations:separationMargin:heightMargin:) (Systems.swift:2146-2152; defaults 2, 0.2, 0.1).
   The stage resolves overlaps between the context's solid agents in CHARACTER-INDEX order — the reference iterates a Swift
   Dictionary, whose order is hash-seed dependent, so it has no canonical result of its own. A few dozen
   characters: one wavefront walks the pair loop with everything in LDS (it can hold SGE_MAX_SEPARATION_AGENTS, the reference's
   scale); more run the same loop as a dataflow over agents — every agent carries a counter of the loops that have passed it, so
   loops that share no agent run side by side and the result is the sequential loop's, bit for bit (sge_ccd.hip, "dataflow";
   faster from ~50 agents on: 1,024 agents 3-10 ms per step against 23-49). Capacity: the context's character count. */
#define SGE_MAX_SEPARATION_AGENTS 1024
int sge_separation_params(sge_context* ctx, int32_t iterations, float separation_margin, float height_margin);
/* Diagnostics of the crowd path, last pass of the last step: out[4] = listed agents, loops drawn, redo flags (bit 0: an agent had
   more than 1,024 neighbours to track, bit 1: an agent was pushed further than a grid cell, bit 2: a loop's bounded wait ran out —
   not expected —, bit 3: a loop had more than 64 pairs that change something; in every case the pass was redone by the one-wavefront
   form, same result; a pass that only raised bit 1 is first tried again on the device with candidates from 7 x 7 cells and counts
   as redone only if that fails too), cell size (float bits). */
int sge_debug_separation(sge_context* ctx, int32_t* out);
/* Which form the newest skin stage of sge_tick took (the rule is in DESIGN.md 3.5): quarters of a RESIDENT workgroup per CU (0: one
   workgroup per character, coming and going) and characters per work unit. bench.py names the kernel it prices from this. */
int sge_debug_skin_form(sge_context* ctx, int32_t* quarters, int32_t* chars_per_unit);
int sge_move_stats_read(sge_context* ctx, sge_move_stats* out, int reset);
/* Diagnostics (SGE_OPT_PROFILE): the HIP-event duration in ms of every skin launch since the last reset of sge_profile_read, oldest
   first. *count = how many there are; min(*count, cap) are written (at most 65,536 are kept). */
int sge_debug_skin_launch_times(sge_context* ctx, float* out_ms, int32_t cap, int32_t* count);
/* Diagnostics: the placement search of the skinned output streams at the last (re)allocation (SGE_OPT_PLACEMENT_PROBES): time of one
   three-stream store pass over the kept placement in ms (0: not probed) and how many placements were timed. */
int sge_debug_placement(sge_context* ctx, float* kept_ms, int32_t* tried);
/* Per-character share of CollisionQueryStats.capsuleSweepIterations (CollisionQuery.swift:280-318) for the LAST fixed step:
   distance evaluations each of characters [first, first + count) spent in its casts (what the scheduler balances on). */
int sge_move_cost_read(sge_context* ctx, int32_t first, int32_t count, int32_t* evaluations);
/* Diagnostics (environment SGE_WAVE_PROF=1): shader-clock cycles of every wavefront of the last grouped move launch,
   8 x uint64 per wavefront: total, gather (setup + traversal), sweep, consume, rounds, sweep trips, traversal steps, start clock. */
int sge_debug_wave_profile(sge_context* ctx, uint64_t* out, int32_t waves);
/* Diagnostics: the scheduling lists of the last move launch: lists[0 .. counts[0]) = the grouped launch's characters ordered by last
   step's cost, lists[count .. count + counts[1]) = the multi-wave launch's characters (lists holds 2 * count entries). */
int sge_debug_move_lists(sge_context* ctx, int32_t* lists, int32_t* counts);

/* ------------------------------------------------------------------------- */
/* Skinned-geometry acceleration structures — the step after skinning:        */
/* Game/RTAccelerationBuilder.swift:75-145 (one primitive acceleration        */
/* structure per skinned item: built once with usage .refit, refitted every   */
/* frame from the dynamic vertex buffer with options .vertexData), and the    */
/* raytraceKernel's reads of the skinned streams at a hit                     */
/* (Game/RayTracing.metalinc:242-296).                                        */
/*                                                                            */
/* Metal's acceleration structure is opaque; here it is a 64-wide BVH (one    */
/* wavefront tests one wide node, one entry per lane): triangles are grouped   */
/* into clusters of <= SGE_BLAS_CLUSTER in a spatial order, an entry is the    */
/* box of one cluster or of one child wide node. The clones of the crowd share */
/* one topology (they share the mesh); what a refit produces per character is  */
/* the entries' boxes, float[entryCount + 1][6] = (min xyz, max xyz), the last  */
/* row being the box of the whole character.                                   */
/* ------------------------------------------------------------------------- */
#define SGE_BLAS_WIDTH 64   /* entries per wide node */
#define SGE_BLAS_CLUSTER 64 /* triangles per leaf entry */

typedef struct sge_blas_info {
    int32_t triangleCount;
    int32_t clusterCount;   /* leaf entries */
    int32_t entryCount;     /* leaf + inner entries */
    int32_t wideCount;      /* wide nodes; node 0 is the root */
    int32_t levels;         /* depth of the wide tree (1 = the root holds only clusters) */
    int32_t incidenceCount; /* sum over vertices of the clusters they belong to */
} sge_blas_info;

/* Host helper (no context): the topology sge_blas_build derives from a mesh, for inspection and tests. Every
 * output pointer may be NULL; call once for `info`, then with arrays of these sizes:
 *   entry_link        [entryCount][2]  (a, b): a >= 0: child wide node, b = 0; a < 0: cluster of slots [~a, ~a + b)
 *   wide_first        [wideCount + 1]  entries of wide node w = [wide_first[w], wide_first[w + 1]), at most SGE_BLAS_WIDTH;
 *                                      children come after their parent
 *   wide_parent_entry [wideCount]      the entry whose box is the union of wide node w's entries (-1 for the root)
 *   slot_triangle     [triangleCount]  primitive id (triangle of the index buffer) at every slot
 *   vertex_entry_start[vertexCount+1], vertex_entries [incidenceCount]: the clusters every vertex belongs to (CSR) */
int sge_blas_topology(const float* positions, int32_t vertex_count, const uint32_t* indices, int32_t index_count,
                      sge_blas_info* info, int32_t* entry_link, int32_t* wide_first, int32_t* wide_parent_entry,
                      uint32_t* slot_triangle, int32_t* vertex_entry_start, int32_t* vertex_entries);

/* encoder.build (RTAccelerationBuilder.swift:75-112) for the crowd's shared skinned mesh: `indices` is the item's
 * slice of dynamicIndexBuffer (RTGeometryCache.swift:289-296; uint16 sources widened as there). The topology comes
 * from the uploaded source positions; needs sge_skinned_mesh_upload first. */
int sge_blas_build(sge_context* ctx, const uint32_t* indices, int32_t index_count);
int sge_blas_info_get(sge_context* ctx, sge_blas_info* info);
/* The item's slice of dynamicUVBuffer (RTGeometryCache.swift:277-283: the skinned item's uvs, copied once): float[V][2].
 * Optional; without it hits report uv = (0, 0). Call after sge_blas_build. */
int sge_blas_set_uvs(sge_context* ctx, const float* uvs, int32_t vertex_count);

/* encoder.refit(..., options: .vertexData) (RTAccelerationBuilder.swift:113-145) for characters [first, first+count)
 * over the context's skinned positions. Asynchronous; also runs as SGE_STAGE_BLAS_REFIT of sge_tick, after the skin
 * stage. With SGE_OPT_FUSE_BLAS_REFIT (the default) the skin stage of the same tick produces the boxes itself and the
 * skinned positions are not read back. */
int sge_blas_refit(sge_context* ctx, int32_t first, int32_t count);
/* The same over caller-owned device buffers: d_positions in `layout`, character k at vertex first_vertex + k * vertexCount;
 * d_bounds receives [count][entryCount + 1][6]. */
int sge_blas_refit_buffers(sge_context* ctx, const void* d_positions, int32_t layout, int64_t first_vertex,
                           int32_t count, void* d_bounds);
/* float [count][entryCount + 1][6] */
int sge_blas_bounds_download(sge_context* ctx, int32_t first, int32_t count, float* bounds);
int sge_blas_buffers(sge_context* ctx, void** d_bounds, void** d_indices);

/* MTLAccelerationStructureInstanceDescriptor.transformationMatrix = item.modelMatrix
 * (RTAccelerationBuilder.swift:168-185): column-major 4x4 per character; identity until uploaded. */
int sge_blas_instances_upload(sge_context* ctx, int32_t first, int32_t count, const float* model_matrices);

/* `ray` of the raytraceKernel (RayTracing.metalinc:231-235). 48 bytes. */
typedef struct sge_blas_ray {
    float origin[3];
    float minDistance;
    float direction[3];
    float maxDistance;
    int32_t instance; /* character index; < 0: every character (the TLAS level: closest hit over all instances, the smaller
                       * instance index winning ties) */
    int32_t _pad[3];
} sge_blas_ray;

/* What the kernel derives from a triangle hit (RayTracing.metalinc:246-296) before materials and lights:
 * primitive_id, distance, triangle_barycentric_coord, the geometric normal of the world-space triangle flipped
 * against the ray, the interpolated shading frame (nW, tW, bW) from the skinned normal / tangent streams, and interp_uv
 * (RayTracing.metalinc:106-119). 80 bytes. */
typedef struct sge_blas_hit {
    int32_t hit;       /* intersection_type::triangle */
    int32_t primitive; /* hit.primitive_id: triangle of the index buffer */
    int32_t instance;  /* hit.instance_id: character index (-1: no hit) */
    float distance;
    float bary[2];     /* weights of the triangle's second and third vertex */
    float geomNormal[3];
    float normal[3];
    float tangent[3];
    float bitangent[3];
    float uv[2];
} sge_blas_hit;

/* `isect.intersect(ray, accel)` (RayTracing.metalinc:242): closest hit of every ray against its instance's refitted
 * structure, or against all of them. Host arrays; synchronous. Ties on distance go to the smaller instance, then the
 * smaller primitive id. The instance level (the TLAS of RTAccelerationBuilder.swift:168-185, rebuilt per call like the reference's
 * per frame) is three scans of 64 boxes: the characters are sorted by the cell of an XZ grid over their world boxes' centres (the
 * grid of the character-vs-character sweeps), 64 consecutive characters of that order form a group, 64 groups a super-group —
 * 250,000 characters are 62 super-groups, one step. */
int sge_blas_intersect_batch(sge_context* ctx, const sge_blas_ray* rays, int32_t count, sge_blas_hit* hits);
/* The same with rays and hits in device memory (a renderer's ray buffers), enqueued on the context's stream: asynchronous.
 * `any_instance` != 0 when some ray may carry instance < 0 (the world boxes are then refreshed first). */
int sge_blas_intersect_device(sge_context* ctx, const void* d_rays, int32_t count, void* d_hits, int32_t any_instance);

/* HIP-event time of the refit launches since the last reset (SGE_OPT_PROFILE). */
int sge_blas_profile_read(sge_context* ctx, double* refit_ms, int64_t* refit_launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* SGE_AMD_H */
