"""ctypes mirror of include/sge_amd.h (POD layouts + prototypes).

The same struct classes describe the product library (libsge_amd.so, HIP) and —
in tests only — the CPU oracle, whose `sgeo_*` entry points share these layouts.
This module never loads anything under oracle/.
"""
import ctypes as C
import os

import numpy as np

SGE_OK, SGE_ERR_INVALID, SGE_ERR_DEVICE, SGE_ERR_STATE, SGE_ERR_CAPACITY, SGE_ERR_NOT_READY = 0, 1, 2, 3, 4, 5
SGE_ABI_VERSION = 2
SGE_MAX_COEFFS = 17
SGE_MAX_PLATFORMS = 64
SET_STATIC, SET_DYNAMIC = 0, 1
SGE_AXIS_ABSENT = 255
SGE_MAX_OVERLAP_HITS = 8
SGE_MANIFOLD_MAX = 4

# BodyType
BODY_STATIC, BODY_KINEMATIC, BODY_DYNAMIC = 0, 1, 2
# agent flags
AGENT_PRESENT, AGENT_SOLID, AGENT_RADIUS_OVERRIDE = 1, 2, 4
# controller flags
CTRL_GROUNDED, CTRL_GROUNDED_NEAR, CTRL_GROUND_SLIDING = 1, 2, 4
# intent flags
INTENT_PRESENT, INTENT_HAS_FACING_YAW, INTENT_DODGE_ACTIVE = 1, 2, 4
# locomotion
LOCO_IDLE, LOCO_WALK, LOCO_RUN, LOCO_FALLING = 0, 1, 2, 3
LOCO_IS_BLENDING, LOCO_PRESENT, MOTION_PRESENT, MOTION_LOOP, MOTION_IN_PLACE = 1, 2, 4, 8, 16
# action flags
ACTION_PRESENT, ACTION_ACTIVE, ACTION_LOOP, ACTION_IN_PLACE, ACTION_EXITING, ACTION_HAS_DODGE = 1, 2, 4, 8, 16, 32
# query modes
CAST, CAST_BLOCKING, CAST_GROUND = 0, 1, 2
# stages
STAGE_INTENT, STAGE_GRAVITY, STAGE_MOVE, STAGE_LOCOMOTION = 1, 2, 4, 8
STAGE_ACTION, STAGE_POSE, STAGE_WRITEBACK, STAGE_SKIN, STAGE_AGENTS = 16, 32, 64, 128, 256
STAGE_BLAS_REFIT = 512
STAGE_SEPARATION = 1024
STAGE_SIDE_CONTACT_CACHE = 1 << 11  # modifier of STAGE_MOVE: SideContactOnlyCachePolicy (Systems.swift:1136-1157)
STAGE_ALL_FIXED, STAGE_ALL = 0x7F, 0xFF
# options
OPT_STORE_POSE_DEBUG, OPT_SKIN_LAYOUT, OPT_PROFILE, OPT_OVERLAP_SKIN, OPT_HEAVY_THRESHOLD, OPT_PLACEMENT_PROBES = 1, 2, 3, 4, 5, 6
OPT_FUSE_BLAS_REFIT = 7
# sge_state_* array bits
STATE_BODIES, STATE_CONTROLLERS, STATE_LOCOMOTION, STATE_ACTIONS, STATE_INTENTS, STATE_WORLD = 1, 2, 4, 8, 16, 15
BLAS_WIDTH, BLAS_CLUSTER = 64, 64
LAYOUT_PACKED, LAYOUT_PADDED16 = 0, 1

f32, f64, i32, u32, u8, u16, i64, u64 = (C.c_float, C.c_double, C.c_int32, C.c_uint32, C.c_uint8,
                                           C.c_uint16, C.c_int64, C.c_uint64)
P = C.POINTER


class SurfaceMaterial(C.Structure):
    _fields_ = [("muS", f32), ("muK", f32), ("flattenGround", u32)]


class BodyState(C.Structure):
    _fields_ = [("position", f64 * 3), ("linearVelocity", f64 * 3), ("rotation", f32 * 4),
                ("transformRotation", f32 * 4), ("bodyType", u32), ("_pad", u32 * 3)]


class ControllerParams(C.Structure):
    _fields_ = [("radius", f32), ("halfHeight", f32), ("skinWidth", f32), ("groundSnapSkin", f32),
                ("snapDistance", f32), ("fallProbeDistance", f32), ("groundSnapMaxSpeed", f32),
                ("groundSnapMaxToi", f32), ("groundSnapMaxStep", f32), ("groundSweepMaxStep", f32),
                ("maxSlideIterations", i32), ("minGroundDot", f32), ("collisionMask", u32),
                ("agentFlags", u32), ("agentRadiusOverride", f32), ("agentMassWeight", f32)]


class ControllerState(C.Structure):
    _fields_ = [("groundNormal", f32 * 3), ("groundTriangleIndex", i32), ("sideContactNormal", f32 * 3),
                ("sideContactFrames", i32), ("manifoldTriangles", i32 * 4), ("manifoldNormals", (f32 * 3) * 4),
                ("manifoldCount", i32), ("manifoldFrames", i32), ("groundTransitionFrames", i32),
                ("flags", u32), ("groundDistance", f32), ("_pad", u32 * 3)]


class MoveIntent(C.Structure):
    _fields_ = [("desiredVelocity", f32 * 3), ("desiredFacingYaw", f32), ("flags", u32),
                ("maxAcceleration", f32), ("maxDeceleration", f32), ("_pad", u32)]


class LocomotionState(C.Structure):
    _fields_ = [("profile", i32 * 4), ("time", f32 * 4), ("idleEnterSpeed", f32), ("idleExitSpeed", f32),
                ("runEnterSpeed", f32), ("runExitSpeed", f32), ("fallMinDropHeight", f32), ("blendTime", f32),
                ("blendT", f32), ("idleInertiaHalfLife", f32), ("idleInertia", f32), ("fromState", i32),
                ("state", i32), ("flags", u32), ("motionTime", f32), ("playbackRate", f32),
                ("motionProfile", i32), ("posePhase", f32)]


class ActionState(C.Structure):
    _fields_ = [("profile", i32), ("time", f32), ("playbackRate", f32), ("weight", f32),
                ("blendInTime", f32), ("blendOutHalfLife", f32), ("dodgeEnd", f32), ("flags", u32)]


class StateView(C.Structure):
    _fields_ = [("first", i32), ("count", i32), ("which", u32), ("ticket", i32), ("bodies", C.c_void_p),
                ("controllers", C.c_void_p), ("locomotion", C.c_void_p), ("actions", C.c_void_p), ("intents", C.c_void_p)]


class SkeletonDesc(C.Structure):
    _fields_ = [("boneCount", i32), ("parent", P(i32)), ("bindLocal", P(f32)), ("invBindModel", P(f32)),
                ("restTranslation", P(f32)), ("rawRestTranslation", P(f32)), ("preRotationDegrees", P(f32)),
                ("rootRotationFix", f32 * 16), ("unitScale", f32), ("pelvisIndex", i32), ("leanIndex", i32)]


class MotionProfileDesc(C.Structure):
    _fields_ = [("order", i32), ("cycleDuration", f32), ("bonePresent", P(u8)), ("coeffCount", P(u8)),
                ("coeffs", P(f32))]


class SkinnedMeshDesc(C.Structure):
    _fields_ = [("vertexCount", i32), ("positions", P(f32)), ("normals", P(f32)), ("tangents", P(f32)),
                ("boneIndices", P(u16)), ("boneWeights", P(f32)), ("invBindModel", P(f32)), ("invBindCount", i32)]


class SkinningJob(C.Structure):
    _fields_ = [("d_sourcePositions", C.c_void_p), ("d_sourceNormals", C.c_void_p), ("d_sourceTangents", C.c_void_p),
                ("d_sourceBoneIndices", C.c_void_p), ("d_sourceBoneWeights", C.c_void_p), ("d_palette", C.c_void_p),
                ("paletteCount", i32), ("vertexCount", i32), ("dstBaseVertex", i32), ("sourceLayout", i32)]


class StaticMeshEntity(C.Structure):
    _fields_ = [("positions", P(f32)), ("vertexCount", i32), ("indices", P(u32)), ("indexCount", i32),
                ("modelMatrix", f32 * 16), ("material", SurfaceMaterial), ("triangleMaterials", P(SurfaceMaterial)),
                ("triangleMaterialCount", i32), ("collisionLayer", u32)]


class BVHNode(C.Structure):
    _fields_ = [("boundsMin", f32 * 3), ("boundsMax", f32 * 3), ("left", i32), ("right", i32), ("start", i32),
                ("count", i32), ("parent", i32)]


class CapsuleQuery(C.Structure):
    _fields_ = [("origin", f32 * 3), ("delta", f32 * 3), ("radius", f32), ("halfHeight", f32), ("minNormalY", f32),
                ("mask", u32), ("mode", u32)]


class CapsuleCastHit(C.Structure):
    _fields_ = [("hit", i32), ("toi", f32), ("position", f32 * 3), ("normal", f32 * 3), ("triangleNormal", f32 * 3),
                ("triangleIndex", i32), ("material", SurfaceMaterial)]


class CapsuleOverlapHit(C.Structure):
    _fields_ = [("depth", f32), ("position", f32 * 3), ("normal", f32 * 3), ("triangleNormal", f32 * 3),
                ("triangleIndex", i32), ("material", SurfaceMaterial)]


class RayQuery(C.Structure):
    _fields_ = [("origin", f32 * 3), ("direction", f32 * 3), ("maxDistance", f32), ("mask", u32)]


class RaycastHit(C.Structure):
    _fields_ = [("hit", i32), ("distance", f32), ("position", f32 * 3), ("normal", f32 * 3), ("triangleIndex", i32),
                ("material", SurfaceMaterial)]


class BlasInfo(C.Structure):
    _fields_ = [("triangleCount", i32), ("clusterCount", i32), ("entryCount", i32), ("wideCount", i32), ("levels", i32),
                ("incidenceCount", i32)]


class BlasRay(C.Structure):
    _fields_ = [("origin", f32 * 3), ("minDistance", f32), ("direction", f32 * 3), ("maxDistance", f32), ("instance", i32),
                ("_pad", i32 * 3)]


class BlasHit(C.Structure):
    _fields_ = [("hit", i32), ("primitive", i32), ("instance", i32), ("distance", f32), ("bary", f32 * 2), ("geomNormal", f32 * 3),
                ("normal", f32 * 3), ("tangent", f32 * 3), ("bitangent", f32 * 3), ("uv", f32 * 2)]


class PlatformState(C.Structure):
    _fields_ = [("aabbMin", f32 * 3), ("aabbMax", f32 * 3), ("delta", f32 * 3), ("kinematic", u32), ("hasAABB", u32),
                ("_pad", u32)]


class TickDesc(C.Structure):
    _fields_ = [("dt", f32), ("gravity", f32 * 3), ("stages", u32), ("first", i32), ("count", i32), ("_pad", u32)]


class AgentState(C.Structure):
    _fields_ = [("position", f32 * 3), ("radius", f32), ("velocity", f32 * 3), ("halfHeight", f32)]


class StageTimes(C.Structure):
    _fields_ = [("move_ms", f64), ("pose_ms", f64), ("skin_ms", f64), ("agents_ms", f64),
                ("move_launches", i64), ("pose_launches", i64), ("skin_launches", i64), ("agents_launches", i64)]


class MoveStats(C.Structure):
    _fields_ = [("queries", u64), ("candidates", u64), ("sweepIterations", u64), ("overflow", u64),
                ("traversalSteps", u64), ("sweepTrips", u64), ("prunedPairs", u64)]


# sizes the C side static_asserts as well
EXPECTED_SIZES = {BodyState: 96, ControllerParams: 64, ControllerState: 128, MoveIntent: 32,
                  LocomotionState: 96, ActionState: 32, AgentState: 32, CapsuleQuery: 44,
                  CapsuleCastHit: 60, CapsuleOverlapHit: 56, BVHNode: 44, TickDesc: 32, RayQuery: 32, RaycastHit: 48,
                  PlatformState: 48}
for _cls, _sz in EXPECTED_SIZES.items():
    assert C.sizeof(_cls) == _sz, (_cls.__name__, C.sizeof(_cls), _sz)

# numpy views of the per-character PODs (same memory layout)
body_dtype = np.dtype([("position", "<f8", 3), ("linearVelocity", "<f8", 3), ("rotation", "<f4", 4),
                       ("transformRotation", "<f4", 4), ("bodyType", "<u4"), ("_pad", "<u4", 3)])
params_dtype = np.dtype([(n, "<f4") for n in ("radius", "halfHeight", "skinWidth", "groundSnapSkin", "snapDistance",
                                               "fallProbeDistance", "groundSnapMaxSpeed", "groundSnapMaxToi",
                                               "groundSnapMaxStep", "groundSweepMaxStep")] +
                        [("maxSlideIterations", "<i4"), ("minGroundDot", "<f4"), ("collisionMask", "<u4"),
                         ("agentFlags", "<u4"), ("agentRadiusOverride", "<f4"), ("agentMassWeight", "<f4")])
controller_dtype = np.dtype([("groundNormal", "<f4", 3), ("groundTriangleIndex", "<i4"),
                             ("sideContactNormal", "<f4", 3), ("sideContactFrames", "<i4"),
                             ("manifoldTriangles", "<i4", 4), ("manifoldNormals", "<f4", (4, 3)),
                             ("manifoldCount", "<i4"), ("manifoldFrames", "<i4"), ("groundTransitionFrames", "<i4"),
                             ("flags", "<u4"), ("groundDistance", "<f4"), ("_pad", "<u4", 3)])
intent_dtype = np.dtype([("desiredVelocity", "<f4", 3), ("desiredFacingYaw", "<f4"), ("flags", "<u4"),
                         ("maxAcceleration", "<f4"), ("maxDeceleration", "<f4"), ("_pad", "<u4")])
locomotion_dtype = np.dtype([("profile", "<i4", 4), ("time", "<f4", 4), ("idleEnterSpeed", "<f4"),
                             ("idleExitSpeed", "<f4"), ("runEnterSpeed", "<f4"), ("runExitSpeed", "<f4"),
                             ("fallMinDropHeight", "<f4"), ("blendTime", "<f4"), ("blendT", "<f4"),
                             ("idleInertiaHalfLife", "<f4"), ("idleInertia", "<f4"), ("fromState", "<i4"),
                             ("state", "<i4"), ("flags", "<u4"), ("motionTime", "<f4"), ("playbackRate", "<f4"),
                             ("motionProfile", "<i4"), ("posePhase", "<f4")])
action_dtype = np.dtype([("profile", "<i4"), ("time", "<f4"), ("playbackRate", "<f4"), ("weight", "<f4"),
                         ("blendInTime", "<f4"), ("blendOutHalfLife", "<f4"), ("dodgeEnd", "<f4"), ("flags", "<u4")])
agent_dtype = np.dtype([("position", "<f4", 3), ("radius", "<f4"), ("velocity", "<f4", 3), ("halfHeight", "<f4")])
query_dtype = np.dtype([("origin", "<f4", 3), ("delta", "<f4", 3), ("radius", "<f4"), ("halfHeight", "<f4"),
                        ("minNormalY", "<f4"), ("mask", "<u4"), ("mode", "<u4")])
material_dtype = np.dtype([("muS", "<f4"), ("muK", "<f4"), ("flattenGround", "<u4")])
cast_hit_dtype = np.dtype([("hit", "<i4"), ("toi", "<f4"), ("position", "<f4", 3), ("normal", "<f4", 3),
                           ("triangleNormal", "<f4", 3), ("triangleIndex", "<i4"), ("material", material_dtype)])
ray_query_dtype = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("maxDistance", "<f4"), ("mask", "<u4")])
raycast_hit_dtype = np.dtype([("hit", "<i4"), ("distance", "<f4"), ("position", "<f4", 3), ("normal", "<f4", 3),
                              ("triangleIndex", "<i4"), ("material", material_dtype)])
platform_dtype = np.dtype([("aabbMin", "<f4", 3), ("aabbMax", "<f4", 3), ("delta", "<f4", 3), ("kinematic", "<u4"),
                           ("hasAABB", "<u4"), ("_pad", "<u4")])
blas_ray_dtype = np.dtype([("origin", "<f4", 3), ("minDistance", "<f4"), ("direction", "<f4", 3), ("maxDistance", "<f4"),
                           ("instance", "<i4"), ("_pad", "<i4", 3)])
blas_hit_dtype = np.dtype([("hit", "<i4"), ("primitive", "<i4"), ("instance", "<i4"), ("distance", "<f4"), ("bary", "<f4", 2),
                           ("geomNormal", "<f4", 3), ("normal", "<f4", 3), ("tangent", "<f4", 3), ("bitangent", "<f4", 3), ("uv", "<f4", 2)])
overlap_hit_dtype = np.dtype([("depth", "<f4"), ("position", "<f4", 3), ("normal", "<f4", 3),
                              ("triangleNormal", "<f4", 3), ("triangleIndex", "<i4"), ("material", material_dtype)])
bvh_node_dtype = np.dtype([("boundsMin", "<f4", 3), ("boundsMax", "<f4", 3), ("left", "<i4"), ("right", "<i4"),
                           ("start", "<i4"), ("count", "<i4"), ("parent", "<i4")])
for _dt, _cls in ((body_dtype, BodyState), (params_dtype, ControllerParams), (controller_dtype, ControllerState),
                  (intent_dtype, MoveIntent), (locomotion_dtype, LocomotionState), (action_dtype, ActionState),
                  (agent_dtype, AgentState), (query_dtype, CapsuleQuery), (cast_hit_dtype, CapsuleCastHit),
                  (overlap_hit_dtype, CapsuleOverlapHit), (bvh_node_dtype, BVHNode), (ray_query_dtype, RayQuery),
                  (raycast_hit_dtype, RaycastHit), (platform_dtype, PlatformState), (blas_ray_dtype, BlasRay),
                  (blas_hit_dtype, BlasHit)):
    assert _dt.itemsize == C.sizeof(_cls), (_cls.__name__, _dt.itemsize)

# Every symbol include/sge_amd.h declares: name -> (restype, argtypes)
VP = C.c_void_p
PROTOTYPES = {
    "sge_context_create": (VP, [C.c_int]),
    "sge_context_destroy": (None, [VP]),
    "sge_last_error": (C.c_char_p, []),
    "sge_abi_version": (C.c_int, []),
    "sge_context_set_stream": (C.c_int, [VP, VP]),
    "sge_synchronize": (C.c_int, [VP]),
    "sge_context_set_option": (C.c_int, [VP, C.c_int, C.c_int]),
    "sge_skeleton_build": (C.c_int, [i32, VP, VP, VP, VP, f32, C.c_int, VP, VP, VP, VP]),
    "sge_skeleton_upload": (C.c_int, [VP, P(SkeletonDesc)]),
    "sge_motion_profiles_upload": (C.c_int, [VP, P(MotionProfileDesc), i32]),
    "sge_mesh_tangents_compute": (C.c_int, [i32, VP, VP, VP, VP, VP, i32, VP]),
    "sge_skinned_mesh_upload": (C.c_int, [VP, P(SkinnedMeshDesc)]),
    "sge_skinning_encode": (C.c_int, [VP, VP, VP, VP, i32, P(SkinningJob), i32]),
    "sge_crowd_buffers": (C.c_int, [VP, P(VP), P(VP), P(VP), P(VP)]),
    "sge_crowd_palette_buffers": (C.c_int, [VP, P(VP), P(i32)]),
    "sge_skin_wait": (C.c_int, [VP, VP]),
    "sge_skin_consumed": (C.c_int, [VP, VP]),
    "sge_skinned_mesh_buffers": (C.c_int, [VP, P(VP), P(VP), P(VP), P(VP), P(VP)]),
    "sge_collision_rebuild_static": (C.c_int, [VP, P(StaticMeshEntity), i32]),
    "sge_collision_rebuild_dynamic": (C.c_int, [VP, P(StaticMeshEntity), i32]),
    "sge_collision_update_transforms": (C.c_int, [VP, i32, VP, VP, i32]),
    "sge_collision_counts": (C.c_int, [VP, P(i32), P(i32), P(i32)]),
    "sge_collision_copy": (C.c_int, [VP, VP, VP, VP, VP, VP, VP]),
    "sge_collision_counts_set": (C.c_int, [VP, i32, P(i32), P(i32), P(i32)]),
    "sge_collision_copy_set": (C.c_int, [VP, i32, VP, VP, VP, VP, VP, VP]),
    "sge_raycast_batch": (C.c_int, [VP, VP, i32, VP]),
    "sge_mesh_world_aabb": (C.c_int, [VP, i32, VP, VP, VP]),
    "sge_platforms_upload": (C.c_int, [VP, VP, i32]),
    "sge_capsule_cast_batch": (C.c_int, [VP, VP, i32, VP]),
    "sge_capsule_overlap_all_batch": (C.c_int, [VP, VP, i32, i32, VP, VP]),
    "sge_capsule_overlap_batch": (C.c_int, [VP, VP, i32, VP, VP]),
    "sge_characters_resize": (C.c_int, [VP, i32]),
    "sge_characters_upload": (C.c_int, [VP, i32, i32, VP, VP, VP, VP, VP, VP]),
    "sge_characters_download": (C.c_int, [VP, i32, i32, VP, VP, VP, VP, VP, VP]),
    "sge_palettes_download": (C.c_int, [VP, i32, i32, VP, VP, VP]),
    "sge_skinned_download": (C.c_int, [VP, i64, i64, VP, VP, VP]),
    "sge_tick": (C.c_int, [VP, P(TickDesc)]),
    "sge_state_pull_async": (C.c_int, [VP, u32, i32, i32, P(i32)]),
    "sge_state_wait": (C.c_int, [VP, i32, P(StateView)]),
    "sge_state_poll": (C.c_int, [VP, i32]),
    "sge_state_push_begin": (C.c_int, [VP, u32, i32, i32, P(StateView)]),
    "sge_state_push_commit": (C.c_int, [VP]),
    "sge_agents_export": (C.c_int, [VP, VP]),
    "sge_agents_import": (C.c_int, [VP, VP, i32, i32]),
    "sge_profile_read": (C.c_int, [VP, P(StageTimes), C.c_int]),
    "sge_separation_params": (C.c_int, [VP, C.c_int32, C.c_float, C.c_float]),
    "sge_move_stats_read": (C.c_int, [VP, P(MoveStats), C.c_int]),
    "sge_context_get_stream": (C.c_int, [VP, P(VP)]),
    "sge_agents_allgather": (C.c_int, [VP, VP, C.c_int32, C.c_int32, C.c_int32]),
    "sge_move_cost_read": (C.c_int, [VP, C.c_int32, C.c_int32, VP]),
    "sge_debug_wave_profile": (C.c_int, [VP, VP, C.c_int32]),
    "sge_debug_move_lists": (C.c_int, [VP, VP, VP]),
    "sge_debug_separation": (C.c_int, [VP, VP]),
    "sge_debug_skin_form": (C.c_int, [VP, P(i32), P(i32)]),
    "sge_debug_skin_launch_times": (C.c_int, [VP, VP, i32, P(i32)]),
    "sge_debug_placement": (C.c_int, [VP, P(f32), P(i32)]),
    "sge_blas_topology": (C.c_int, [VP, i32, VP, i32, P(BlasInfo), VP, VP, VP, VP, VP, VP]),
    "sge_blas_build": (C.c_int, [VP, VP, i32]),
    "sge_blas_info_get": (C.c_int, [VP, P(BlasInfo)]),
    "sge_blas_set_uvs": (C.c_int, [VP, VP, i32]),
    "sge_blas_refit": (C.c_int, [VP, i32, i32]),
    "sge_blas_refit_buffers": (C.c_int, [VP, VP, i32, i64, i32, VP]),
    "sge_blas_bounds_download": (C.c_int, [VP, i32, i32, VP]),
    "sge_blas_buffers": (C.c_int, [VP, P(VP), P(VP)]),
    "sge_blas_instances_upload": (C.c_int, [VP, i32, i32, VP]),
    "sge_blas_intersect_batch": (C.c_int, [VP, VP, i32, VP]),
    "sge_blas_intersect_device": (C.c_int, [VP, VP, i32, VP, i32]),
    "sge_blas_profile_read": (C.c_int, [VP, P(C.c_double), P(i64), C.c_int]),
}

LIB_NAME = "libsge_amd.so"


def library_path():
    # SGE_AMD_LIB selects another build of the SAME HIP library (e.g. a diagnostic -DSGE_CCD_TIMING build)
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("SGE_AMD_LIB", LIB_NAME))


def bind(lib, prefix="sge_", names=None):
    """Attach restype/argtypes for every declared symbol (raises AttributeError if one is missing)."""
    for name, (res, args) in PROTOTYPES.items():
        if names is not None and name not in names:
            continue
        fn = getattr(lib, prefix + name[len("sge_"):])
        fn.restype = res
        fn.argtypes = args
    return lib


def load_library(path=None):
    """Loads the HIP product library. No fallback: a missing build is an error."""
    path = path or library_path()
    # PyTorch wheels bundle their own copy of the HIP / HSA runtimes and open them by path: if this library (linked against
    # /opt/rocm's) is loaded first, a later `import torch` brings a second runtime into the process, which then finds no
    # GPU. Loading torch first makes both share one runtime (same sonames). torch is plumbing here, not a dependency.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    return bind(C.CDLL(path))


def ptr(a):
    """void* of a numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
