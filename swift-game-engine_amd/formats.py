"""On-disk asset formats of the character-update path -> the dense float32 tables the C ABI takes.

Host-side mirrors of the reference's four loaders (same field names, the same validation and the same
"nil / empty asset" outcomes; numpy float32 arithmetic in the reference's operation order):

  load_skeleton          SkeletonLoader.loadSkeleton / buildSkeleton      Game/SkeletonLoader.swift:12-87, 119-167
  load_motion_profile    MotionProfileLoader.load                         Game/Animation.swift:11-63
  load_skinned_mesh      SkinnedMeshLoader.loadSkinnedMeshAsset/buildAsset Game/SkinnedMeshLoader.swift:17-188
  load_static_mesh       StaticMeshLoader.loadStaticMeshAsset/buildAsset   Game/StaticMeshLoader.swift:30-161
  transform_from_matrix  DemoScene.transformFromMatrix                     Game/DemoScene.swift:718-735
  model_matrix           TransformComponent.modelMatrix                    Game/Components.swift:26-44

JSON numbers are decoded the way Swift's JSONDecoder yields `Float`: parse as double, round once to float32.
Every loader accepts a path, a parsed JSON object, or an exporter payload (exporters.py) holding numpy arrays.
"""
import json
import os

import numpy as np

from . import abi, assets

MAX_COEFFS = abi.SGE_MAX_COEFFS
AXIS_ABSENT = 255


def _obj(src):
    """path | dict -> dict, or None when the file is missing / not JSON (the loaders' `return nil`)."""
    if isinstance(src, dict):
        return src
    try:
        with open(os.fspath(src), "r", encoding="utf-8") as f:
            return json.load(f)
    except (OSError, ValueError) as e:
        print("formats: failed to load json:", src, e)
        return None


def _f32(values):
    return np.asarray(values, np.float64).astype(np.float32)


def _vec3(values, fallback=(0.0, 0.0, 0.0)):
    """SkeletonLoader.swift:163-166: fewer than three components -> fallback."""
    if len(values) < 3:
        return np.asarray(fallback, np.float32)
    return _f32(values[:3])


# --------------------------------------------------------------------------- #
# skeleton                                                                     #
# --------------------------------------------------------------------------- #

class SkeletonAsset:
    """The fields of `Skeleton` the GPU path consumes, before sge_skeleton_build derives bindLocal/invBindModel."""

    def __init__(self, names, parent, translations, pre_rotation_degrees, unit_scale, root_fix_degrees, zero_root,
                 aliases=None):
        self.names = list(names)
        self.parent = np.ascontiguousarray(parent, np.int32)
        self.translations = np.ascontiguousarray(translations, np.float32)
        self.pre_rotation_degrees = np.ascontiguousarray(pre_rotation_degrees, np.float32)
        self.unit_scale = float(np.float32(unit_scale))
        self.root_fix_degrees = np.ascontiguousarray(root_fix_degrees, np.float32)
        self.zero_root = bool(zero_root)
        sem = assets.resolve_semantic(self.names, aliases if aliases is not None else assets.MIXAMO_ALIASES)
        self.semantic = sem
        self.pelvis_index = sem.get("pelvis", -1)
        # chest ?? spine3 ?? spine2 ?? spine1 (ProceduralPoseSystem.swift:371-374)
        self.lean_index = sem.get("chest", sem.get("spine3", sem.get("spine2", sem.get("spine1", -1))))
        self.profile_names = []
        self.profiles = []

    @property
    def bone_count(self):
        return len(self.names)

    def profile_index(self, name):
        return self.profile_names.index(name)

    def add_profile(self, name, profile):
        self.profile_names.append(name)
        self.profiles.append(dict(profile, name=name))
        return len(self.profiles) - 1


def load_skeleton(src, rig_profile_aliases=None):
    """-> SkeletonAsset, or None (missing file, undecodable JSON, mismatched arrays: SkeletonLoader.swift:31-36, 44-46)."""
    js = _obj(src)
    if js is None:
        return None
    try:
        names = js["names"]
        parent = js["parent"]
        translations = js["translations"]
        pre = js["preRotationDegrees"]
        rig = js["rigProfile"]
        root = js["root"]
        scale = js["unitScale"]
        _ = js["version"], js["name"]
    except KeyError as e:  # JSONDecoder throws on a missing non-optional key -> nil
        print("formats: skeleton json lacks", e)
        return None
    count = len(names)
    if count == 0 or len(parent) != count or len(translations) != count:
        print("formats: skeleton arrays do not match.")
        return None
    raw = np.stack([_vec3(t) for t in translations])
    if len(pre) == 0:
        pre_rot = np.zeros((count, 3), np.float32)
    elif len(pre) == count:
        pre_rot = np.stack([_vec3(t) for t in pre])
    else:
        print("formats: preRotationDegrees count mismatch.")
        return None
    rig_name = str(rig["name"])
    if rig_profile_aliases is not None:
        aliases = rig_profile_aliases
        override_is_mixamo = bool(assets.resolve_semantic(["mixamorig:Hips"], aliases))
    else:
        # rigProfileFrom (:119-139): mixamo base or empty, overridden per semantic key by rigProfile.aliases
        aliases = dict(assets.MIXAMO_ALIASES) if rig_name.lower() == "mixamo" else {}
        for key, lst in (rig.get("aliases") or {}).items():
            aliases[key] = list(lst)
        override_is_mixamo = False
    rule = str(root["rule"]).lower()
    if rule in ("zero", "zero_root", "zero-root"):
        zero_root = True
    elif rule == "auto":
        zero_root = rig_name.lower() == "mixamo" or override_is_mixamo
    else:  # keep / preserve / anything else
        zero_root = False
    return SkeletonAsset(names, parent, raw, pre_rot, np.float32(scale), _vec3(root["rotationFixDegrees"]), zero_root,
                         aliases)


# --------------------------------------------------------------------------- #
# motion profiles                                                              #
# --------------------------------------------------------------------------- #

def load_motion_profile(src, names):
    """MotionProfile JSON -> the dense per-skeleton-bone table of sge_motion_profile_desc, or None.

    Flattening replaces the `profile.bones[boneName]` dictionary lookups of ProceduralPoseSystem.swift:153-154:
    bonePresent[i] = the dictionary has skeleton.names[i]; coeffCount = the array length per axis
    (SGE_AXIS_ABSENT for a nil axis); cycleDuration = phase?.cycle_duration ?? duration (:39-44)."""
    js = _obj(src)
    if js is None:
        return None
    try:
        bones = js["bones"]
        order = int(js["order"])
        duration = js["duration"]
        sample_fps = int(js["sample_fps"])
        _ = js["version"], js["name"]
    except (KeyError, TypeError, ValueError) as e:
        print("formats: motion profile json lacks", e)
        return None
    B = len(names)
    present = np.zeros(B, np.uint8)
    count = np.full((B, 6), AXIS_ABSENT, np.uint8)
    coeffs = np.zeros((B, 6, MAX_COEFFS), np.float32)
    for i, name in enumerate(names):
        bone = bones.get(name)
        if bone is None:
            continue
        if "translation" not in bone or "rotation" not in bone:  # both channels are non-optional (:18-21)
            print("formats: motion profile bone lacks a channel:", name)
            return None
        present[i] = 1
        for c, chan in enumerate(("translation", "rotation")):
            for a, axis in enumerate("xyz"):
                v = bone[chan].get(axis)
                if v is None:
                    continue
                if len(v) > MAX_COEFFS:
                    raise ValueError("motion profile axis longer than SGE_MAX_COEFFS: %s.%s.%s" % (name, chan, axis))
                count[i, c * 3 + a] = len(v)
                coeffs[i, c * 3 + a, :len(v)] = _f32(v)
    cycle = (js.get("phase") or {}).get("cycle_duration")
    if cycle is None:
        cycle = duration
    return {"name": js["name"], "order": order, "cycleDuration": float(np.float32(cycle)),
            "duration": float(np.float32(duration)), "sampleFps": sample_fps,
            "bonePresent": present, "coeffCount": count, "coeffs": coeffs}


# --------------------------------------------------------------------------- #
# skinned mesh                                                                 #
# --------------------------------------------------------------------------- #

def matrix_from_array_row_major(values):
    """16 row-major floats -> column-major [16] (matrixFromArrayRowMajor, SkinnedMeshLoader.swift:181-188)."""
    return np.ascontiguousarray(_f32(values).reshape(4, 4).T).reshape(16)


def _last_component(name):
    """`name.split(separator: ":").last`: Swift's split omits empty subsequences ("hips:" -> "hips", ":" -> nil)."""
    parts = [p for p in name.split(":") if p]
    return parts[-1] if parts else None


def make_bone_remap(skin_bone_names, skeleton_names):
    """makeBoneRemap (SkinnedMeshLoader.swift:139-163): skin bone -> skeleton index by lower-cased name, with the
    part after the last ':' registered too; -1 when missing."""
    lookup = {}
    for i, name in enumerate(skeleton_names):
        lookup[name.lower()] = i
        short = _last_component(name)
        if short is not None:
            lookup[short.lower()] = i
    out = np.full(len(skin_bone_names), -1, np.int32)
    for i, name in enumerate(skin_bone_names):
        key = name.lower()
        idx = lookup.get(key)
        if idx is None and ":" in key:
            short = _last_component(key)  # (the reference force-unwraps: a key made of colons only would trap there)
            idx = lookup.get(short) if short is not None else None
        if idx is not None:
            out[i] = idx
    missing = int((out < 0).sum())
    if missing:
        print("formats: missing bones:", missing, "of", len(skin_bone_names))
    return out


def load_skinned_mesh(src, skeleton, skeleton_inv_bind_model, merge_submeshes=True):
    """-> dict(streams..., meshes=[descriptors], materialNames=[...]) or None.

    streams: positions [V,3] (x unitScale), normals [V,3], uvs [V,2], boneIndices [V,4] u16 (remapped to skeleton
    indices, unmapped influences zeroed), boneWeights [V,4] (renormalised in float32: ((x+y)+z)+w, then /=).
    invBindModel [B,16]: the skeleton's, with every mapped skin bone's inverseBindMatrix (translation x unitScale)
    substituted — the matrices the palette is re-bound with (Systems.swift:2519-2527).
    meshes: one descriptor per non-empty submesh, sharing the streams (SkinnedMeshLoader.swift:118-134); with
    merge_submeshes a single "indices" over all submeshes is also returned for the crowd path, which skins the
    shared streams once instead of once per submesh."""
    js = _obj(src)
    if js is None:
        return None
    try:
        mesh, skin = js["mesh"], js["skin"]
        positions, normals, uvs = mesh["positions"], mesh["normals"], mesh["uvs"]
        joints, weights, indices = mesh["joints"], mesh["weights"], mesh["indices"]
        bones = skin["bones"]
        _ = js["version"]
    except KeyError as e:
        print("formats: skinned json lacks", e)
        return None
    v_count = len(positions) // 3
    if not (v_count > 0 and len(positions) == v_count * 3 and len(normals) == v_count * 3 and len(uvs) == v_count * 2
            and len(joints) == v_count * 4 and len(weights) == v_count * 4):
        print("formats: attribute counts do not match.")
        return {"vertexCount": 0, "meshes": [], "materialNames": []}
    bone_names = [b["name"] for b in bones]
    bone_map = make_bone_remap(bone_names, skeleton.names)

    # buildInvBindModel (:165-179)
    inv_bind = np.array(skeleton_inv_bind_model, np.float32).reshape(-1, 16).copy()
    scale = np.float32(skeleton.unit_scale)
    for i, b in enumerate(bones):
        m = b["inverseBindMatrix"]
        if bone_map[i] < 0 or len(m) != 16:
            continue
        col = matrix_from_array_row_major(m)
        col[12:15] = col[12:15] * scale
        inv_bind[bone_map[i]] = col

    pos = _f32(positions).reshape(v_count, 3) * scale
    nrm = _f32(normals).reshape(v_count, 3)
    uv = _f32(uvs).reshape(v_count, 2)
    src_joint = np.asarray(joints, np.int64).reshape(v_count, 4)
    w = _f32(weights).reshape(v_count, 4).copy()
    in_range = src_joint < len(bone_map)
    mapped = np.where(in_range, bone_map[np.minimum(src_joint, max(len(bone_map) - 1, 0))], -1)
    w[mapped < 0] = 0
    bone_indices = np.where(mapped < 0, 0, mapped).astype(np.uint16)
    total = ((w[:, 0] + w[:, 1]) + w[:, 2]) + w[:, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where((total > 0)[:, None], w / total[:, None], w).astype(np.float32)

    idx = np.asarray(indices, np.uint32)
    subs = mesh.get("submeshes") or [{"start": 0, "count": len(idx), "material": "Default"}]
    descriptors, material_names = [], []
    for sub in subs:
        start = max(int(sub["start"]), 0)
        end = min(start + int(sub["count"]), len(idx))
        if start >= end:
            continue
        descriptors.append({"name": "SkinnedMesh:%s" % sub["material"], "indices": idx[start:end].copy()})
        material_names.append(sub["material"])
    out = {"vertexCount": v_count, "positions": pos, "normals": nrm, "uvs": uv, "boneIndices": bone_indices,
           "boneWeights": w, "invBindModel": inv_bind, "meshes": descriptors, "materialNames": material_names,
           "boneMap": bone_map}
    if merge_submeshes and descriptors:
        out["indices"] = np.concatenate([d["indices"] for d in descriptors])
    return out


# --------------------------------------------------------------------------- #
# static mesh                                                                  #
# --------------------------------------------------------------------------- #

def load_static_mesh(src):
    """-> list of parts dict(name, transform [16] column-major, positions [V,3], normals|None, uvs|None, indices u32,
    submeshes, collisionHulls=[dict(positions, indices)]) or None; invalid entries are skipped as in the reference."""
    js = _obj(src)
    if js is None:
        return None
    try:
        entries = js["meshes"]
        _ = js["version"]
    except KeyError as e:
        print("formats: static json lacks", e)
        return None
    parts = []
    for entry in entries:
        mesh = entry["mesh"]
        positions, indices = mesh["positions"], mesh["indices"]
        v_count = len(positions) // 3
        if not (v_count > 0 and len(positions) == v_count * 3):
            print("formats: invalid positions for mesh:", entry["name"])
            continue
        if len(indices) == 0:
            print("formats: missing indices for mesh:", entry["name"])
            continue
        normals, uvs = mesh.get("normals", []), mesh.get("uvs", [])
        has_n, has_uv = len(normals) == v_count * 3, len(uvs) == v_count * 2
        subs = mesh.get("submeshes") or [{"start": 0, "count": len(indices), "material": "Default"}]
        hulls = []
        for hull in entry.get("collisionHulls") or []:
            hv = len(hull["positions"]) // 3
            if not (hv > 0 and len(hull["positions"]) == hv * 3) or len(hull["indices"]) == 0:
                continue
            hulls.append({"positions": _f32(hull["positions"]).reshape(hv, 3), "indices": np.asarray(hull["indices"], np.uint32)})
        tr = entry.get("transform", [])
        transform = matrix_from_array_row_major(tr) if len(tr) == 16 else np.eye(4, dtype=np.float32).reshape(16)
        parts.append({
            "name": entry["name"], "transform": transform,
            "positions": _f32(positions).reshape(v_count, 3),
            "normals": _f32(normals).reshape(v_count, 3) if has_n else None,
            "uvs": _f32(uvs).reshape(v_count, 2) if has_uv else None,
            "indices": np.asarray(indices, np.uint32),
            "submeshes": [{"start": int(s["start"]), "count": int(s["count"]), "material": s["material"]} for s in subs],
            "collisionHulls": hulls,
        })
    return parts


# --------------------------------------------------------------------------- #
# TransformComponent (float32, Apple simd operation order as restated in oracle/sge_oracle_math.h)
# --------------------------------------------------------------------------- #

_F = np.float32


def quat_from_rotation(cols):
    """simd_quatf(float3x3): trace / largest-diagonal branches. cols = (x, y, z) column vectors -> (x, y, z, w)."""
    (m00, m01, m02), (m10, m11, m12), (m20, m21, m22) = [[_F(v) for v in c] for c in cols]
    trace = _F(_F(m00 + m11) + m22)
    one, two, four = _F(1), _F(2), _F(4)
    if trace >= 0:
        r = _F(two * np.sqrt(_F(one + trace)))
        ri = _F(one / r)
        return np.array([ri * _F(m12 - m21), ri * _F(m20 - m02), ri * _F(m01 - m10), r / four], np.float32)
    if m00 >= m11 and m00 >= m22:
        r = _F(two * np.sqrt(_F(_F(_F(one - m11) - m22) + m00)))
        ri = _F(one / r)
        return np.array([r / four, ri * _F(m01 + m10), ri * _F(m02 + m20), ri * _F(m12 - m21)], np.float32)
    if m11 >= m22:
        r = _F(two * np.sqrt(_F(_F(_F(one - m00) - m22) + m11)))
        ri = _F(one / r)
        return np.array([ri * _F(m01 + m10), r / four, ri * _F(m12 + m21), ri * _F(m20 - m02)], np.float32)
    r = _F(two * np.sqrt(_F(_F(_F(one - m00) - m11) + m22)))
    ri = _F(one / r)
    return np.array([ri * _F(m02 + m20), ri * _F(m12 + m21), r / four, ri * _F(m01 - m10)], np.float32)


def quat_mul(p, q):
    """simd_mul(simd_quatf, simd_quatf) in the SDK's shuffle order."""
    p, q = np.asarray(p, np.float32), np.asarray(q, np.float32)
    a = np.array([q[3], -q[2], q[1], -q[0]], np.float32) * p[0] + np.array([q[2], q[3], -q[0], -q[1]], np.float32) * p[1]
    b = np.array([-q[1], q[0], q[3], -q[2]], np.float32) * p[2] + np.array([q[0], q[1], q[2], q[3]], np.float32) * p[3]
    return (a + b).astype(np.float32)


def quat_angle_axis(angle, axis):
    h = _F(_F(angle) / _F(2))
    s, c = _F(np.sin(h)), _F(np.cos(h))
    ax = np.asarray(axis, np.float32)
    return np.array([s * ax[0], s * ax[1], s * ax[2], c], np.float32)


def matrix_from_quat(v):
    """matrix_float4x4(simd_quatf) -> [4 columns][4]."""
    x, y, z, w = [_F(c) for c in v]
    one, two = _F(1), _F(2)
    return np.array([
        [one - two * _F(y * y + z * z), two * _F(x * y + z * w), two * _F(x * z - y * w), 0],
        [two * _F(x * y - z * w), one - two * _F(z * z + x * x), two * _F(y * z + x * w), 0],
        [two * _F(z * x + y * w), two * _F(y * z - x * w), one - two * _F(y * y + x * x), 0],
        [0, 0, 0, 1]], np.float32)


def _mat_mul_cols(a, b):
    """simd_mul(float4x4, float4x4) on [column][row] arrays: ((c0*x + c1*y) + c2*z) + c3*w per result column."""
    out = np.zeros((4, 4), np.float32)
    for j in range(4):
        v = b[j]
        out[j] = ((a[0] * v[0] + a[1] * v[1]) + a[2] * v[2]) + a[3] * v[3]
    return out


def transform_from_matrix(m):
    """DemoScene.transformFromMatrix: column-major [16] -> dict(translation, rotation (x,y,z,w), scale)."""
    c = np.asarray(m, np.float32).reshape(4, 4)  # c[j] = column j
    t = c[3, :3].copy()
    axes, scale = [], []
    for j, fallback in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))):
        v = c[j, :3]
        length = _F(np.sqrt(_F(_F(_F(v[0] * v[0]) + _F(v[1] * v[1])) + _F(v[2] * v[2]))))
        scale.append(length)
        axes.append(v / length if length > 0 else np.asarray(fallback, np.float32))
    return {"translation": t, "rotation": quat_from_rotation(axes), "scale": np.asarray(scale, np.float32)}


def model_matrix(transform):
    """TransformComponent.modelMatrix = T * (R * S) -> column-major [16]."""
    t, s = np.asarray(transform["translation"], np.float32), np.asarray(transform["scale"], np.float32)
    T = np.eye(4, dtype=np.float32)
    T[3, :3] = t
    R = matrix_from_quat(transform["rotation"])
    S = np.diag(np.array([s[0], s[1], s[2], 1], np.float32)).astype(np.float32)
    return _mat_mul_cols(T, _mat_mul_cols(R, S)).reshape(16)


# --------------------------------------------------------------------------- #
# compact binary form of the two mesh payloads (what tests/golden/ keeps instead of multi-MB JSON text)
# --------------------------------------------------------------------------- #

def _small_index(a):
    a = np.asarray(a)
    return a.astype(np.uint16) if a.size == 0 or int(a.max()) <= 0xFFFF else a.astype(np.uint32)


def save_skinned_payload(path, payload):
    m, bones = payload["mesh"], payload["skin"]["bones"]
    subs = m.get("submeshes") or []
    np.savez_compressed(
        path, kind=np.array("skinned"), version=np.int32(payload["version"]),
        positions=np.asarray(m["positions"], np.float32), normals=np.asarray(m["normals"], np.float32),
        uvs=np.asarray(m["uvs"], np.float32), joints=np.asarray(m["joints"], np.uint8 if len(bones) <= 256 else np.uint16),
        weights=np.asarray(m["weights"], np.float32), indices=_small_index(m["indices"]),
        submeshStart=np.asarray([s["start"] for s in subs], np.int64),
        submeshCount=np.asarray([s["count"] for s in subs], np.int64),
        submeshMaterial=np.array([s["material"] for s in subs]),
        boneNames=np.array([b["name"] for b in bones]),
        inverseBindMatrix=np.asarray([b["inverseBindMatrix"] for b in bones], np.float32))


def save_static_payload(path, payload, keep_shading=False):
    out = {"kind": np.array("static"), "version": np.int32(payload["version"]), "meshCount": np.int32(len(payload["meshes"]))}
    for k, e in enumerate(payload["meshes"]):
        m = e["mesh"]
        subs = m.get("submeshes") or []
        out[f"m{k}.name"] = np.array(e["name"])
        out[f"m{k}.transform"] = np.asarray(e["transform"], np.float32)
        out[f"m{k}.positions"] = np.asarray(m["positions"], np.float32)
        out[f"m{k}.indices"] = _small_index(m["indices"])
        if keep_shading:
            out[f"m{k}.normals"] = np.asarray(m["normals"], np.float32)
            out[f"m{k}.uvs"] = np.asarray(m["uvs"], np.float32)
        out[f"m{k}.submeshStart"] = np.asarray([s["start"] for s in subs], np.int64)
        out[f"m{k}.submeshCount"] = np.asarray([s["count"] for s in subs], np.int64)
        out[f"m{k}.submeshMaterial"] = np.array([s["material"] for s in subs])
        hulls = e.get("collisionHulls") or []
        out[f"m{k}.hullCount"] = np.int32(len(hulls))
        for h, hull in enumerate(hulls):
            out[f"m{k}.hull{h}.positions"] = np.asarray(hull["positions"], np.float32)
            out[f"m{k}.hull{h}.indices"] = _small_index(hull["indices"])
    np.savez_compressed(path, **out)


def load_payload(path):
    """.npz written by save_*_payload -> the payload dict the JSON file would decode to (arrays stay numpy)."""
    z = np.load(path)
    kind = str(z["kind"])
    if kind == "skinned":
        subs = [{"start": int(s), "count": int(c), "material": str(mname)}
                for s, c, mname in zip(z["submeshStart"], z["submeshCount"], z["submeshMaterial"])]
        return {"version": int(z["version"]),
                "mesh": {"positions": z["positions"], "normals": z["normals"], "uvs": z["uvs"], "joints": z["joints"],
                         "weights": z["weights"], "indices": z["indices"].astype(np.uint32), "submeshes": subs},
                "skin": {"bones": [{"name": str(n), "inverseBindMatrix": m} for n, m in zip(z["boneNames"], z["inverseBindMatrix"])]}}
    meshes = []
    for k in range(int(z["meshCount"])):
        subs = [{"start": int(s), "count": int(c), "material": str(mname)}
                for s, c, mname in zip(z[f"m{k}.submeshStart"], z[f"m{k}.submeshCount"], z[f"m{k}.submeshMaterial"])]
        hulls = [{"positions": z[f"m{k}.hull{h}.positions"], "indices": z[f"m{k}.hull{h}.indices"].astype(np.uint32)}
                 for h in range(int(z[f"m{k}.hullCount"]))]
        meshes.append({"name": str(z[f"m{k}.name"]), "transform": z[f"m{k}.transform"],
                       "mesh": {"positions": z[f"m{k}.positions"],
                                "normals": z[f"m{k}.normals"] if f"m{k}.normals" in z else np.zeros(0, np.float32),
                                "uvs": z[f"m{k}.uvs"] if f"m{k}.uvs" in z else np.zeros(0, np.float32),
                                "indices": z[f"m{k}.indices"].astype(np.uint32), "submeshes": subs},
                       "collisionHulls": hulls})
    return {"version": int(z["version"]), "meshes": meshes}
