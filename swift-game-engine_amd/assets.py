"""Host-side asset preparation for the character-update path (numpy only, no GPU).

Mirrors the reference's loaders just far enough to feed the C ABI:
  - RigProfile.mixamo / resolve          Game/Skeleton.swift:44-90
  - resolveRootRule                      Game/SkeletonLoader.swift:141-158
  - controller / locomotion defaults     Game/Components.swift:230-293, 353-431,
                                         Game/CharacterFactory.swift:88-107
and generates the synthetic stand-ins for the assets missing from the reference
checkout (.MISSING_LARGE_BLOBS: YBot.skinned.json, 17-Cheese.static.json):
a ~14k-vertex 4-influence skinned mesh wrapped around the real Y-Bot skeleton,
and a 71,680-triangle static terrain (the triangle count of 17-Cheese.fbx).
"""
import os

import numpy as np

from . import abi

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# Skeleton.RigProfile.mixamo(), Skeleton.swift:63-89
MIXAMO_ALIASES = {
    "pelvis": ["mixamorig:Hips", "Hips", "pelvis"],
    "spine1": ["mixamorig:Spine", "Spine", "spine_01"],
    "spine2": ["mixamorig:Spine1", "Spine1", "spine_02"],
    "spine3": ["mixamorig:Spine2", "Spine2", "spine_03"],
    "neck": ["mixamorig:Neck", "Neck", "neck_01"],
    "head": ["mixamorig:Head", "Head"],
    "clavicleL": ["mixamorig:LeftShoulder", "LeftShoulder", "clavicle_l"],
    "upperarmL": ["mixamorig:LeftArm", "LeftArm", "upperarm_l"],
    "lowerarmL": ["mixamorig:LeftForeArm", "LeftForeArm", "lowerarm_l"],
    "handL": ["mixamorig:LeftHand", "LeftHand", "hand_l"],
    "clavicleR": ["mixamorig:RightShoulder", "RightShoulder", "clavicle_r"],
    "upperarmR": ["mixamorig:RightArm", "RightArm", "upperarm_r"],
    "lowerarmR": ["mixamorig:RightForeArm", "RightForeArm", "lowerarm_r"],
    "handR": ["mixamorig:RightHand", "RightHand", "hand_r"],
    "thighL": ["mixamorig:LeftUpLeg", "LeftUpLeg", "thigh_l"],
    "calfL": ["mixamorig:LeftLeg", "LeftLeg", "calf_l"],
    "footL": ["mixamorig:LeftFoot", "LeftFoot", "foot_l"],
    "ballL": ["mixamorig:LeftToeBase", "LeftToeBase", "ball_l"],
    "thighR": ["mixamorig:RightUpLeg", "RightUpLeg", "thigh_r"],
    "calfR": ["mixamorig:RightLeg", "RightLeg", "calf_r"],
    "footR": ["mixamorig:RightFoot", "RightFoot", "foot_r"],
    "ballR": ["mixamorig:RightToeBase", "RightToeBase", "ball_r"],
}


def resolve_semantic(names, aliases=MIXAMO_ALIASES):
    """RigProfile.resolve (Skeleton.swift:44-61): lower-cased name table, first matching alias wins."""
    table = {n.lower(): i for i, n in enumerate(names)}
    out = {}
    for semantic, lst in aliases.items():
        for alias in lst:
            if alias.lower() in table:
                out[semantic] = table[alias.lower()]
                break
    return out


def resolve_root_rule(rule, rig_profile_name):
    """SkeletonLoader.swift:141-158 -> True when the root rest translation is zeroed."""
    r = rule.lower()
    if r in ("zero", "zero_root", "zero-root"):
        return True
    if r in ("keep", "preserve"):
        return False
    if r == "auto":
        return rig_profile_name.lower() == "mixamo"
    return False


class YBotAssets:
    """The Y-Bot skeleton + five motion profiles as dense float32 tables (tests/golden/ybot_assets.npz)."""

    def __init__(self, path=None):
        z = np.load(path or os.path.join(GOLDEN_DIR, "ybot_assets.npz"))
        self.names = [str(n) for n in z["names"]]
        self.parent = z["parent"].astype(np.int32)
        self.translations = z["translations"].astype(np.float32)
        self.pre_rotation_degrees = z["preRotationDegrees"].astype(np.float32)
        self.unit_scale = float(z["unitScale"])
        self.root_fix_degrees = z["rootRotationFixDegrees"].astype(np.float32)
        self.zero_root = resolve_root_rule(str(z["rootRule"]), str(z["rigProfile"]))
        sem = resolve_semantic(self.names)
        self.pelvis_index = sem.get("pelvis", -1)
        # chest ?? spine3 ?? spine2 ?? spine1 (ProceduralPoseSystem.swift:371-374); mixamo() has no chest alias
        self.lean_index = sem.get("chest", sem.get("spine3", sem.get("spine2", sem.get("spine1", -1))))
        self.profile_names = [str(n) for n in z["profileNames"]]
        self.profiles = []
        for p in self.profile_names:
            self.profiles.append({
                "name": p,
                "order": int(z[f"{p}.order"]),
                "cycleDuration": float(z[f"{p}.cycleDuration"]),
                "sampleFps": int(z[f"{p}.sampleFps"]),
                "bonePresent": np.ascontiguousarray(z[f"{p}.bonePresent"], np.uint8),
                "coeffCount": np.ascontiguousarray(z[f"{p}.coeffCount"], np.uint8),
                "coeffs": np.ascontiguousarray(z[f"{p}.coeffs"], np.float32),
            })

    @property
    def bone_count(self):
        return len(self.names)

    def profile_index(self, name):
        return self.profile_names.index(name)


# --------------------------------------------------------------------------- #
# synthetic skinned mesh                                                       #
# --------------------------------------------------------------------------- #

def make_synthetic_skinned_mesh(parent, bind_model, rings=22, segments=10, radius=0.16, seed=7):
    """A tube of rings x segments vertices along every non-root bone (64 bones x 220 = 14,080 vertices
    for the Y-Bot), with 1-4 skin influences per vertex: the bone's parent and the bone itself blended
    along the tube, plus the grandparent near the parent end and a child near the far end.

    bind_model: [B][16] column-major bind model matrices. Returns a dict of packed float32/uint16 arrays
    plus uint32 indices and uvs (for sge_mesh_tangents_compute)."""
    rng = np.random.default_rng(seed)
    B = len(parent)
    pos_bone = bind_model.reshape(B, 4, 4)[:, 3, :3]  # column 3 = translation
    children = [[] for _ in range(B)]
    for i, p in enumerate(parent):
        if p >= 0:
            children[p].append(i)
    P, N, UV, IDX, W, TRI = [], [], [], [], [], []
    base = 0
    for b in range(B):
        p = parent[b]
        if p < 0:
            continue
        a0, a1 = pos_bone[p], pos_bone[b]
        d = a1 - a0
        L = float(np.linalg.norm(d))
        if L < 1e-4:
            axis = np.array([0.0, 1.0, 0.0])
            L = 0.05
            a1 = a0 + axis * L
        else:
            axis = d / L
        ref = np.array([1.0, 0.0, 0.0]) if abs(axis[0]) < 0.9 else np.array([0.0, 0.0, 1.0])
        u = np.cross(axis, ref)
        u /= np.linalg.norm(u)
        v = np.cross(axis, u)
        r = min(radius, 0.45 * L + 0.02)
        t = np.linspace(0.0, 1.0, rings)
        ang = np.linspace(0.0, 2 * np.pi, segments, endpoint=False)
        tt, aa = np.meshgrid(t, ang, indexing="ij")
        bulge = r * (0.55 + 0.45 * np.sin(np.pi * tt))
        radial = np.cos(aa)[..., None] * u + np.sin(aa)[..., None] * v
        pts = a0 + tt[..., None] * (a1 - a0) + bulge[..., None] * radial
        nrm = radial + 0.15 * (0.5 - tt)[..., None] * axis
        nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
        P.append(pts.reshape(-1, 3))
        N.append(nrm.reshape(-1, 3))
        UV.append(np.stack([aa / (2 * np.pi), tt], -1).reshape(-1, 2))
        gp = parent[p]
        ch = children[b][0] if children[b] else -1
        tflat = tt.reshape(-1)
        w = np.zeros((tflat.size, 4))
        idx = np.zeros((tflat.size, 4), np.int64)
        idx[:, 0], idx[:, 1] = p, b
        w[:, 0], w[:, 1] = 1.0 - tflat, tflat
        if gp >= 0:
            idx[:, 2] = gp
            w[:, 2] = np.clip(0.25 - tflat, 0.0, None)
        if ch >= 0:
            idx[:, 3] = ch
            w[:, 3] = np.clip(tflat - 0.75, 0.0, None)
        w *= 1.0 + 0.05 * rng.standard_normal(w.shape) * (w > 0)
        w = np.clip(w, 0.0, None)
        w /= w.sum(1, keepdims=True)
        IDX.append(idx)
        W.append(w)
        for i in range(rings - 1):
            for j in range(segments):
                j2 = (j + 1) % segments
                q0, q1 = base + i * segments + j, base + i * segments + j2
                q2, q3 = q0 + segments, q1 + segments
                TRI += [q0, q2, q1, q1, q2, q3]
        base += rings * segments
    return {
        "positions": np.ascontiguousarray(np.concatenate(P), np.float32),
        "normals": np.ascontiguousarray(np.concatenate(N), np.float32),
        "uvs": np.ascontiguousarray(np.concatenate(UV), np.float32),
        "boneIndices": np.ascontiguousarray(np.concatenate(IDX), np.uint16),
        "boneWeights": np.ascontiguousarray(np.concatenate(W), np.float32),
        "indices": np.asarray(TRI, np.uint32),
    }


# --------------------------------------------------------------------------- #
# synthetic static mesh                                                        #
# --------------------------------------------------------------------------- #

def terrain_height(x, z, half_x, half_z):
    h = 1.6 * np.sin(0.11 * x) * np.cos(0.09 * z) + 0.9 * np.sin(0.31 * x + 0.7) * np.sin(0.27 * z - 0.4)
    # "cheese holes": smooth pits with steep (non-walkable) walls
    for cx, cz, rad, depth in ((-40.0, -20.0, 9.0, 5.0), (25.0, 30.0, 7.0, 4.0), (60.0, -35.0, 11.0, 6.0),
                               (-70.0, 40.0, 8.0, 4.5), (0.0, 0.0, 6.0, 3.0)):
        d = np.sqrt((x - cx) ** 2 + (z - cz) ** 2)
        h = h - depth / (1.0 + np.exp((d - rad) * 1.8))
    # rim: a wall too steep to stand on keeps the crowd on the mesh
    edge = np.minimum(half_x - np.abs(x), half_z - np.abs(z))
    h = h + 14.0 * np.clip((7.0 - edge) / 7.0, 0.0, 1.0) ** 2 * 2.0
    return h


def make_synthetic_static_mesh(cells_x=224, cells_z=160, cell=1.0):
    """Height-field terrain with cells_x*cells_z*2 triangles (71,680 = 17-Cheese.fbx's count by default).
    Winding gives +Y normals. Returns (positions [V][3] f32, indices [T*3] u32)."""
    half_x, half_z = cells_x * cell * 0.5, cells_z * cell * 0.5
    xs = np.linspace(-half_x, half_x, cells_x + 1)
    zs = np.linspace(-half_z, half_z, cells_z + 1)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = terrain_height(X, Z, half_x, half_z)
    pos = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(cells_x), np.arange(cells_z), indexing="ij")
    v00 = (i * (cells_z + 1) + j).reshape(-1)
    v10, v01, v11 = v00 + (cells_z + 1), v00 + 1, v00 + (cells_z + 1) + 1
    tris = np.stack([v00, v01, v10, v10, v01, v11], -1).reshape(-1).astype(np.uint32)
    return np.ascontiguousarray(pos), np.ascontiguousarray(tris)


def ground_plane(size=80.0, y=-3.0):
    """ProceduralMeshes.plane(size 80) at y=-3 (ProceduralMeshes.swift:169-181, DemoScene.swift:87,103,122):
    returns local positions, indices and the translation model matrix."""
    s = size * 0.5
    pos = np.array([[-s, 0, s], [s, 0, s], [s, 0, -s], [-s, 0, -s]], np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    m = np.eye(4, dtype=np.float32)
    m[3, :3] = (0, y, 0)  # column-major: row index = column
    return pos, idx, m.reshape(16)


# --------------------------------------------------------------------------- #
# component defaults                                                           #
# --------------------------------------------------------------------------- #

def default_controller_params(n):
    """CharacterControllerComponent defaults (Components.swift:380-404) with the player's
    radius/halfHeight/skin values (CharacterFactory.swift:88-91)."""
    p = np.zeros(n, abi.params_dtype)
    p["radius"], p["halfHeight"], p["skinWidth"], p["groundSnapSkin"] = 1.5, 1.0, 0.3, 0.05
    p["snapDistance"], p["fallProbeDistance"] = 0.8, 200.0
    p["groundSnapMaxSpeed"], p["groundSnapMaxToi"] = 5.0, 0.1
    p["groundSnapMaxStep"], p["groundSweepMaxStep"] = 0.1, 0.1
    p["maxSlideIterations"], p["minGroundDot"] = 4, 0.5
    p["collisionMask"] = 0xFFFFFFFF
    p["agentFlags"], p["agentRadiusOverride"], p["agentMassWeight"] = 0, 0.0, 1.0
    return p


def default_controller_state(n):
    c = np.zeros(n, abi.controller_dtype)
    c["groundNormal"] = (0, 1, 0)
    c["groundTriangleIndex"] = -1
    c["groundDistance"] = np.finfo(np.float32).max
    return c


def default_bodies(n, positions):
    b = np.zeros(n, abi.body_dtype)
    b["position"] = positions
    b["rotation"] = (0, 0, 0, 1)           # simd_quatf(angle: 0, axis: (0,1,0))
    b["transformRotation"] = (0, 0, 0, 1)
    b["bodyType"] = abi.BODY_DYNAMIC
    return b


def default_intents(n, desired_velocity=None):
    """MoveIntentComponent + the player's MovementComponent(maxAcceleration 20, maxDeceleration 36)
    (CharacterFactory.swift:86-87)."""
    it = np.zeros(n, abi.intent_dtype)
    it["flags"] = abi.INTENT_PRESENT
    it["maxAcceleration"], it["maxDeceleration"] = 20.0, 36.0
    if desired_velocity is not None:
        it["desiredVelocity"] = desired_velocity
    return it


def default_locomotion(n, assets, state=abi.LOCO_IDLE):
    """LocomotionProfileComponent as CharacterFactory.swift:97-107 builds it."""
    L = np.zeros(n, abi.locomotion_dtype)
    L["profile"] = (assets.profile_index("Idle"), assets.profile_index("Walking"),
                    assets.profile_index("Running"), assets.profile_index("FallingIdle"))
    L["idleEnterSpeed"], L["idleExitSpeed"], L["runEnterSpeed"], L["runExitSpeed"] = 0.15, 0.3, 6.0, 5.0
    L["fallMinDropHeight"], L["blendTime"], L["blendT"] = 50.0, 0.2, 1.0
    L["idleInertiaHalfLife"], L["idleInertia"] = 0.18, 0.0
    L["fromState"], L["state"] = abi.LOCO_IDLE, state
    L["flags"] = abi.LOCO_PRESENT | abi.MOTION_PRESENT | abi.MOTION_LOOP | abi.MOTION_IN_PLACE
    L["playbackRate"] = 1.0
    L["motionProfile"] = L["profile"][:, state]
    return L


def default_actions(n, assets=None, present=False):
    """ActionAnimationComponent + DodgeActionComponent of CharacterFactory.swift:109-123 (inactive)."""
    a = np.zeros(n, abi.action_dtype)
    a["playbackRate"], a["blendInTime"], a["blendOutHalfLife"] = 1.0, 0.08, 0.18
    if present and assets is not None:
        k = assets.profile_index("StandingDodgeBackward")
        a["profile"] = k
        fps = max(assets.profiles[k]["sampleFps"], 1)
        a["dodgeEnd"] = np.float32(34.0) / np.float32(fps)
        a["flags"] = abi.ACTION_PRESENT | abi.ACTION_IN_PLACE | abi.ACTION_HAS_DODGE
    return a


def crowd_phase_offsets(n, cycle):
    """time_i = cycle * ((i * 2654435761 mod 2^32) / 2^32)."""
    i = np.arange(n, dtype=np.uint64)
    frac = ((i * np.uint64(2654435761)) % np.uint64(1 << 32)).astype(np.float64) / float(1 << 32)
    return (np.float32(cycle) * frac.astype(np.float32)).astype(np.float32)
