#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ from the reference checkout.

Run in the build container only (it reads /root/reference, which does not exist on
the GPU box):   python tests/golden/make_golden.py

Outputs (data only — inputs and expected outputs, no reference source text):
  ybot_assets.npz           Game/YBot.skeleton.json + the five Game/*.motionProfile.json,
                            flattened to the dense float32 tables sge_skeleton_upload /
                            sge_motion_profiles_upload take (JSON doubles -> Float exactly
                            as Swift's JSONDecoder rounds them).
  ornate_mirror_static.npz  Game/ornate_mirror.static.json (the one static asset present in
                            the checkout): part mesh + its two collision hulls + transform.
  ybot_skinned.npz          ExternalResources/Y Bot.fbx in the *.skinned.json schema (positions, normals, uvs,
  cheese_static.npz         joints, weights, indices, submeshes, skin bones + inverseBindMatrix) and
  semla_static.npz          17-Cheese.fbx / Semla.fbx in the *.static.json schema (positions, indices, transform),
                            produced by swift-game-engine_amd/exporters.py (see make_fbx_assets).
  pose_chain_f64.npz        float64 golden vectors for the pose chain, computed by functions
                            IMPORTED from the reference's own Tools/FitMotion/fit_motion.py
                            (rotation_xyz_degrees, mat_mul, translation_matrix,
                            build_model_transforms, parse_skeleton_json :137-244, and
                            compute_foot_contacts :247-351 for the foot positions).
"""
import importlib.util
import json
import math
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
MAX_COEFFS = 17
AXIS_ABSENT = 255
PROFILES = ["Idle", "Walking", "Running", "FallingIdle", "StandingDodgeBackward"]


def load_fit_motion():
    path = os.path.join(REF, "Tools/FitMotion/fit_motion.py")
    spec = importlib.util.spec_from_file_location("ref_fit_motion", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def flatten_profile(profile, names):
    B = len(names)
    present = np.zeros(B, np.uint8)
    count = np.full((B, 6), AXIS_ABSENT, np.uint8)
    coeffs = np.zeros((B, 6, MAX_COEFFS), np.float32)
    for i, name in enumerate(names):
        bone = profile["bones"].get(name)
        if bone is None:
            continue
        present[i] = 1
        for c, chan in enumerate(("translation", "rotation")):
            for a, axis in enumerate("xyz"):
                v = bone[chan].get(axis)
                if v is None:
                    continue
                assert len(v) <= MAX_COEFFS
                count[i, c * 3 + a] = len(v)
                coeffs[i, c * 3 + a, : len(v)] = np.asarray(v, np.float64).astype(np.float32)
    phase = profile.get("phase") or {}
    cycle = phase.get("cycle_duration")
    if cycle is None:
        cycle = profile["duration"]
    return present, count, coeffs, np.float32(cycle)


def make_assets():
    sk = json.load(open(os.path.join(REF, "Game/YBot.skeleton.json")))
    names = sk["names"]
    out = {
        "names": np.array(names),
        "parent": np.asarray(sk["parent"], np.int32),
        "translations": np.asarray(sk["translations"], np.float64).astype(np.float32),
        "preRotationDegrees": np.asarray(sk["preRotationDegrees"], np.float64).astype(np.float32),
        "unitScale": np.float32(sk["unitScale"]),
        "rootRotationFixDegrees": np.asarray(sk["root"]["rotationFixDegrees"], np.float32),
        "rootRule": np.array(sk["root"]["rule"]),
        "rigProfile": np.array(sk["rigProfile"]["name"]),
        "profileNames": np.array(PROFILES),
    }
    for p in PROFILES:
        prof = json.load(open(os.path.join(REF, f"Game/{p}.motionProfile.json")))
        present, count, coeffs, cycle = flatten_profile(prof, names)
        out[f"{p}.bonePresent"] = present
        out[f"{p}.coeffCount"] = count
        out[f"{p}.coeffs"] = coeffs
        out[f"{p}.cycleDuration"] = cycle
        out[f"{p}.duration"] = np.float32(prof["duration"])
        out[f"{p}.order"] = np.int32(prof["order"])
        out[f"{p}.sampleFps"] = np.int32(prof["sample_fps"])
    np.savez_compressed(os.path.join(OUT, "ybot_assets.npz"), **out)
    return sk


def make_static():
    m = json.load(open(os.path.join(REF, "Game/ornate_mirror.static.json")))
    e = m["meshes"][0]
    out = {
        "name": np.array(e["name"]),
        "transformRowMajor": np.asarray(e["transform"], np.float64).astype(np.float32),
        "positions": np.asarray(e["mesh"]["positions"], np.float64).astype(np.float32).reshape(-1, 3),
        "indices": np.asarray(e["mesh"]["indices"], np.uint32),
    }
    for k, hull in enumerate(e.get("collisionHulls") or []):
        out[f"hull{k}.positions"] = np.asarray(hull["positions"], np.float64).astype(np.float32).reshape(-1, 3)
        out[f"hull{k}.indices"] = np.asarray(hull["indices"], np.uint32)
    np.savez_compressed(os.path.join(OUT, "ornate_mirror_static.npz"), **out)


def fourier_f64(coeffs, phase, order):
    """Animation.swift:66-78 evaluated in float64 on the float32-rounded coefficients."""
    if len(coeffs) == 0:
        return 0.0
    p = max(0.0, min(phase, 1.0))
    r = float(coeffs[0])
    idx = 1
    for k in range(1, order + 1):
        if idx + 1 >= len(coeffs):
            break
        ang = 2.0 * math.pi * k * p
        r += float(coeffs[idx]) * math.cos(ang) + float(coeffs[idx + 1]) * math.sin(ang)
        idx += 2
    return r


class FourierCurve:
    """Duck-typed stand-in for fit_motion.AnimationCurve: .sample(t) evaluates the fitted series at phase t."""

    def __init__(self, coeffs, order):
        self.coeffs = coeffs
        self.order = order

    def sample(self, t):
        return fourier_f64(self.coeffs, t, self.order)


def make_pose_chain(fm):
    sk = fm.parse_skeleton_json(__import__("pathlib").Path(os.path.join(REF, "Game/YBot.skeleton.json")))
    names, parent = sk["names"], sk["parent"]
    translations, pre, scale = sk["translations"], sk["pre_rotations"], sk["scale"]
    # float32-rounded inputs, as the Swift loader sees them
    translations = [[float(np.float32(v)) for v in t] for t in translations]
    pre = [[float(np.float32(v)) for v in t] for t in pre]
    scale = float(np.float32(scale))
    sk = dict(sk, translations=translations, pre_rotations=pre, scale=scale)
    root_fix_deg = sk.get("root_rotation_fix", [0.0, 0.0, 0.0])
    root_fix = fm.rotation_xyz_degrees(*root_fix_deg)
    phases = [0.0, 0.125, 0.37, 0.5, 0.73, 0.999]
    out = {"phases": np.asarray(phases, np.float64)}

    # bind pose (SkeletonLoader.buildSkeleton restated with fit_motion's primitives)
    bind_local = []
    for i in range(len(names)):
        rest = [0.0, 0.0, 0.0] if i == 0 else [v * scale for v in translations[i]]
        rot = fm.mat_mul(fm.rotation_xyz_degrees(*pre[i]), fm.rotation_xyz_degrees(0.0, 0.0, 0.0))
        if i == 0:
            rot = fm.mat_mul(root_fix, rot)
        bind_local.append(fm.mat_mul(fm.translation_matrix(*rest), rot))
    out["bind.local"] = np.asarray(bind_local, np.float64).reshape(len(names), 16)
    out["bind.model"] = np.asarray(fm.build_model_transforms(parent, bind_local), np.float64).reshape(len(names), 16)

    for p in PROFILES:
        prof = json.load(open(os.path.join(REF, f"Game/{p}.motionProfile.json")))
        order = prof["order"]
        locals_, models = [], []
        bone_anims = {}
        for name in names:
            bone = prof["bones"].get(name)
            if bone is None:
                continue
            entry = {"translation": {}, "rotation": {}}
            for chan in ("translation", "rotation"):
                for axis in "xyz":
                    v = bone[chan].get(axis)
                    if v is not None:
                        entry[chan][axis] = FourierCurve([float(np.float32(x)) for x in v], order)
            bone_anims[name] = entry
        for ph in phases:
            local = []
            for i, name in enumerate(names):
                anim = bone_anims.get(name)
                rest_raw = translations[i]
                rest_scaled = [0.0, 0.0, 0.0] if i == 0 else [v * scale for v in rest_raw]
                if anim is None:
                    local.append(bind_local[i])  # ProceduralPoseSystem.swift:245 `continue`
                    continue
                tc, rc = anim["translation"], anim["rotation"]
                anim_raw = [tc[a].sample(ph) if a in tc else rest_raw[k] for k, a in enumerate("xyz")]
                delta = [anim_raw[k] - rest_raw[k] for k in range(3)]
                trans = [rest_scaled[k] + delta[k] * scale for k in range(3)]
                if i == 0:  # in_place
                    trans[0] = rest_scaled[0]
                    trans[2] = rest_scaled[2]
                anim_rot = [rc[a].sample(ph) if a in rc else 0.0 for a in "xyz"]
                rot = fm.mat_mul(fm.rotation_xyz_degrees(*pre[i]), fm.rotation_xyz_degrees(*anim_rot))
                if i == 0:
                    rot = fm.mat_mul(root_fix, rot)
                local.append(fm.mat_mul(fm.translation_matrix(*trans), rot))
            locals_.append(np.asarray(local, np.float64).reshape(len(names), 16))
            models.append(np.asarray(fm.build_model_transforms(parent, local), np.float64).reshape(len(names), 16))
        out[f"{p}.local"] = np.stack(locals_)
        out[f"{p}.model"] = np.stack(models)
        # Entirely reference-computed: compute_foot_contacts builds the same chain internally
        # (fit_motion.py:271-313) and thresholds the foot heights; we keep its weights.
        lw, rw, lh, rh = fm.compute_foot_contacts(bone_anims, sk, phases, True)
        out[f"{p}.footContactLeft"] = np.asarray(lw, np.float64)
        out[f"{p}.footContactRight"] = np.asarray(rw, np.float64)
        if lh and rh:
            out[f"{p}.footAuxLeft"] = np.asarray(lh, np.float64)
            out[f"{p}.footAuxRight"] = np.asarray(rh, np.float64)
    np.savez_compressed(os.path.join(OUT, "pose_chain_f64.npz"), **out)


def make_fbx_assets():
    """The three assets the checkout lacks as exporter output (.MISSING_LARGE_BLOBS: YBot.skinned.json,
    17-Cheese.static.json, Semla.static.json), regenerated from the binary FBX data files under
    ExternalResources/ by the package's Blender-free restatement of the two exporter scripts.  The static
    exporter is first pinned against the one exporter output the checkout does hold."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    pkg = importlib.import_module("swift-game-engine_amd")
    fbx, ex, formats = pkg.fbx, pkg.exporters, pkg.formats

    pin = ex.export_static_mesh(fbx.FbxScene(os.path.join(REF, "ExternalResources/ornate-mirror/source/ornate_mirror.fbx")))
    ref = json.load(open(os.path.join(REF, "Game/ornate_mirror.static.json")))["meshes"][0]
    got = pin["meshes"][0]
    assert np.array_equal(np.asarray(ref["mesh"]["indices"], np.uint32), got["mesh"]["indices"])
    assert np.array_equal(np.asarray(ref["mesh"]["positions"], np.float64).astype(np.float32), got["mesh"]["positions"])
    assert np.abs(np.asarray(ref["transform"], np.float64) - got["transform"]).max() < 1e-7
    assert np.abs(np.asarray(ref["mesh"]["uvs"], np.float64) - got["mesh"]["uvs"]).max() < 1e-6
    print("static exporter pinned against Game/ornate_mirror.static.json: indices/positions exact")

    skinned = ex.export_skinned_mesh(fbx.FbxScene(os.path.join(REF, "ExternalResources/Y Bot.fbx")))
    formats.save_skinned_payload(os.path.join(OUT, "ybot_skinned.npz"), skinned)
    for src, dst in (("ExternalResources/17-Cheese.fbx", "cheese_static.npz"),
                     ("ExternalResources/semla/source/Semla.fbx", "semla_static.npz")):
        formats.save_static_payload(os.path.join(OUT, dst), ex.export_static_mesh(fbx.FbxScene(os.path.join(REF, src))))


def main():
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present; fixtures are committed, nothing to do")
    make_assets()
    make_static()
    make_pose_chain(load_fit_motion())
    make_fbx_assets()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
