"""Benchmark / test world construction on top of CharacterEngine (host side, numpy only).

Implements the concrete configs of SURVEY.md §8(d):
  config 1  one Y-Bot over the 80x80 ground quad at y=-3 (CharacterFactory.swift:77-78, DemoScene.swift:87-130)
  config 2  N clones, Running profile, steady state, per-clone phase offsets, pose + LBS only
  config 3  config 2's crowd + CCD against the static triangle mesh, seeded headings, walk/run intents
"""
import os

import numpy as np

from . import abi, assets as A, formats as F


def upload_character_assets(engine, ybot, rings=22, segments=10, mesh_inv_bind=False):
    """Skeleton, the five motion profiles and the (synthetic) skinned source mesh."""
    built = engine.upload_skeleton(ybot)
    engine.upload_profiles(ybot.profiles)
    B = ybot.bone_count
    # bind model = inverse of invBindModel is not needed: recompute from bindLocal on the host
    local = built["bindLocal"].reshape(B, 4, 4).astype(np.float64)
    model = np.zeros_like(local)
    for i in range(B):
        p = ybot.parent[i]
        # stored column-major: arr[c][r]; (A*B)[c] = sum_k A[k] * B[c][k]
        model[i] = local[i] if p < 0 else np.einsum("kr,ck->cr", model[p], local[i])
    mesh = A.make_synthetic_skinned_mesh(ybot.parent, model.reshape(B, 16).astype(np.float32), rings=rings, segments=segments)
    inv = built["invBindModel"] if mesh_inv_bind else None
    engine.upload_skinned_mesh(mesh, inv_bind_model=inv)
    return built, mesh


def upload_ybot_mesh(engine, ybot, path=None):
    """Skeleton, profiles and the real Y-Bot skinned mesh (tests/golden/ybot_skinned.npz = ExternalResources/Y Bot.fbx
    through exporters.export_skinned_mesh) loaded the way SkinnedMeshLoader.buildAsset does; the palette is re-bound
    with the mesh's own inverse bind matrices (Systems.swift:2519-2527).  The submeshes share one vertex stream and
    are skinned once (the reference re-skins the whole stream per submesh item, RTGeometryCache.swift:266-315)."""
    built = engine.upload_skeleton(ybot)
    engine.upload_profiles(ybot.profiles)
    payload = F.load_payload(path or os.path.join(A.GOLDEN_DIR, "ybot_skinned.npz"))
    asset = F.load_skinned_mesh(payload, ybot, built["invBindModel"])
    if asset is None or not asset["meshes"]:
        raise ValueError("skinned mesh asset is empty")
    mesh = {k: asset[k] for k in ("positions", "normals", "uvs", "boneIndices", "boneWeights", "indices")}
    engine.upload_skinned_mesh(mesh, inv_bind_model=asset["invBindModel"])
    return built, asset


# DemoScene.swift:331-333: upright (pi/2 about X) then flip (pi about X) — takes the exporters' Blender Z-up part
# transforms to the engine's Y-up
def _upright_flip():
    pi = np.float32(float.fromhex("0x1.921fb4p+1"))  # Swift's Float.pi (rounded toward zero)
    return F.quat_mul(F.quat_angle_axis(pi * np.float32(0.5), (1, 0, 0)), F.quat_angle_axis(pi, (1, 0, 0)))


STATIC_ASSETS = {"cheese": "cheese_static.npz", "semla": "semla_static.npz", "mirror": "ornate_mirror_static.npz"}


def _load_static_parts(name):
    path = os.path.join(A.GOLDEN_DIR, STATIC_ASSETS[name])
    z = np.load(path)
    if "kind" in z:  # exporter payload
        return F.load_static_mesh(F.load_payload(path))
    # ornate_mirror_static.npz predates the payload form: the reference's own JSON, flattened
    return [{"name": str(z["name"]), "transform": F.matrix_from_array_row_major(z["transformRowMajor"]),
             "positions": z["positions"], "indices": z["indices"], "collisionHulls": []}]


def asset_scene_entities(which=("cheese",), footprint=200.0, gap=30.0, ground=True):
    """Static entities for a crowd-sized scene out of the engine's own static assets.

    Each part keeps the transform its *.static.json carries (transformFromMatrix, DemoScene.swift:718-735), turned
    from the exporter's Blender Z-up world into the engine's Y-up by the demo's upright * flip quaternion (:331-333)
    applied in world space, so a part with its own rotation (the cheese) ends up resting on its Blender-world base.
    Because the demo places these props for a single character (a 3-unit cheese, an 8x mirror), each asset is scaled
    uniformly so its XZ footprint spans `footprint` units, rested on y = 0 and laid out along +X `gap` apart.  `ground` adds the demo's ground quad (plane at
    y = -3, DemoScene.swift:103-130) stretched under the whole layout.  Returns (entities, bounds list)."""
    entities, bounds = [], []
    cursor = 0.0
    uf = _upright_flip()
    for layer_bit, name in enumerate(which):
        parts = _load_static_parts(name)
        prepared = []
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        for part in parts:
            t = F.transform_from_matrix(part["transform"])
            t["rotation"] = F.quat_mul(uf, t["rotation"])
            t["translation"] = (F.matrix_from_quat(uf)[:3, :3].T @ t["translation"]).astype(np.float32)
            m = F.model_matrix(t).reshape(4, 4)  # [col][row]
            w = part["positions"].astype(np.float64) @ m[:3, :3].astype(np.float64) + m[3, :3]
            lo, hi = np.minimum(lo, w.min(0)), np.maximum(hi, w.max(0))
            prepared.append((part, t))
        scale = np.float32(footprint / max(hi[0] - lo[0], hi[2] - lo[2]))
        centre = 0.5 * (lo + hi)
        width = (hi[0] - lo[0]) * scale
        shift = np.array([cursor + 0.5 * width - centre[0] * scale, -lo[1] * scale, -centre[2] * scale])
        for part, t in prepared:
            t = dict(t, scale=(t["scale"] * scale).astype(np.float32),
                     translation=(t["translation"] * scale + shift).astype(np.float32))
            entities.append({"positions": part["positions"], "indices": part["indices"], "modelMatrix": F.model_matrix(t),
                             "material": (0.6, 0.5, 0), "layer": 1 << layer_bit, "name": "%s:%s" % (name, part["name"])})
        bounds.append({"name": name, "lo": lo * scale + shift, "hi": hi * scale + shift, "scale": float(scale)})
        cursor += width + gap
    if ground:
        total = cursor - gap
        hx, hz = 0.5 * total + 150.0, 0.5 * footprint + 150.0
        pos = np.array([[-hx, 0, -hz], [hx, 0, -hz], [hx, 0, hz], [-hx, 0, hz]], np.float32)
        idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)
        m = np.eye(4, dtype=np.float32)
        m[3, :3] = (0.5 * total, -3.0, 0.0)
        entities.append({"positions": pos, "indices": idx, "modelMatrix": m.reshape(16), "material": (0.9, 0.8, 0),
                         "layer": 1 << 30, "name": "ground"})
    return entities, bounds


def upload_asset_scene(engine, which=("cheese",), footprint=200.0, gap=30.0, ground=True):
    entities, bounds = asset_scene_entities(which, footprint, gap, ground)
    engine.rebuild_static(entities)
    return {"entities": entities, "bounds": bounds, "kind": "assets"}


def spawn_positions_on_scene(engine, n, scene, seed=1234, drop=5.0, radius=1.5, half_height=1.0, margin=8.0):
    """Jittered grid over the assets' XZ bounds; the local height comes from the collision world itself: a downward
    capsuleCast from above the scene (CollisionQuery.capsuleCast), then `drop` units above the contact."""
    rng = np.random.default_rng(seed)
    per = [n // len(scene["bounds"])] * len(scene["bounds"])
    per[0] += n - sum(per)
    xs, zs, tops = [], [], []
    for cnt, b in zip(per, scene["bounds"]):
        hx, hz = 0.5 * (b["hi"][0] - b["lo"][0]), 0.5 * (b["hi"][2] - b["lo"][2])
        m = min(margin, 0.25 * min(hx, hz))
        hx, hz = hx - m, hz - m
        cx, cz = 0.5 * (b["hi"][0] + b["lo"][0]), 0.5 * (b["hi"][2] + b["lo"][2])
        gx = max(int(np.ceil(np.sqrt(cnt * hx / hz))), 1)
        gz = max(int(np.ceil(cnt / gx)), 1)
        ix, iz = np.meshgrid(np.arange(gx), np.arange(gz), indexing="ij")
        ix, iz = ix.reshape(-1)[:cnt], iz.reshape(-1)[:cnt]
        xs.append(cx - hx + (ix + 0.5 + rng.uniform(-0.35, 0.35, cnt)) * (2 * hx / gx))
        zs.append(cz - hz + (iz + 0.5 + rng.uniform(-0.35, 0.35, cnt)) * (2 * hz / gz))
        tops.append(np.full(cnt, b["hi"][1]))
    x, z, top = np.concatenate(xs), np.concatenate(zs), np.concatenate(tops)
    start = top + radius + half_height + 10.0
    origins = np.stack([x, start, z], -1)
    reach = start + 3.0 + 10.0  # down to below the ground quad
    deltas = np.stack([np.zeros(n), -reach, np.zeros(n)], -1)
    from .engine import make_queries
    hits = engine.capsule_cast(make_queries(origins, deltas, radius=radius, half_height=half_height))
    y = np.where(hits["hit"] != 0, start - hits["toi"], -3.0 + radius + half_height) + drop
    return np.stack([x, y, z], -1), rng


def upload_terrain(engine, cells=(224, 160), cell=1.0):
    pos, idx = A.make_synthetic_static_mesh(cells[0], cells[1], cell)
    engine.rebuild_static([{"positions": pos, "indices": idx, "material": (0.8, 0.6, 0), "layer": 1}])
    return {"positions": pos, "indices": idx, "cells": cells, "cell": cell,
            "half": (cells[0] * cell * 0.5, cells[1] * cell * 0.5)}


def upload_ground_plane(engine):
    pos, idx, m = A.ground_plane()
    engine.rebuild_static([{"positions": pos, "indices": idx, "modelMatrix": m, "material": (0.9, 0.8, 0), "layer": 1}])


def spawn_positions(n, terrain, seed=1234, margin=12.0, drop=5.0, radius=1.5, half_height=1.0):
    """Jittered grid over the terrain's XZ bounds, `drop` units above the local height (capsule bottom)."""
    rng = np.random.default_rng(seed)
    margin = min(margin, 0.45 * min(terrain["half"]))
    hx, hz = terrain["half"][0] - margin, terrain["half"][1] - margin
    gx = int(np.ceil(np.sqrt(n * hx / hz)))
    gz = int(np.ceil(n / gx))
    ix, iz = np.meshgrid(np.arange(gx), np.arange(gz), indexing="ij")
    ix, iz = ix.reshape(-1)[:n], iz.reshape(-1)[:n]
    x = -hx + (ix + 0.5 + rng.uniform(-0.35, 0.35, n)) * (2 * hx / gx)
    z = -hz + (iz + 0.5 + rng.uniform(-0.35, 0.35, n)) * (2 * hz / gz)
    full_hx, full_hz = terrain["half"]
    y = A.terrain_height(x, z, full_hx, full_hz) + radius + half_height + drop
    return np.stack([x, y, z], -1), rng


def spawn_crowd(engine, ybot, n, terrain=None, seed=1234, mode="ccd", agents=False, mixed=False):
    """Uploads n characters. mode 'lbs': config 2 (steady run, no collision use);
    mode 'ccd': config 3 (intents walk 4.5 / run 12.5 units/s in a seeded random heading)."""
    engine.resize(n)
    params = A.default_controller_params(n)
    if agents:
        params["agentFlags"] = abi.AGENT_PRESENT | abi.AGENT_SOLID
        params["agentMassWeight"] = 1.0
    ctrl = A.default_controller_state(n)
    actions = A.default_actions(n, ybot, present=True)
    run_cycle = ybot.profiles[ybot.profile_index("Running")]["cycleDuration"]
    if mode == "lbs":
        pos = np.zeros((n, 3))
        pos[:, 0] = (np.arange(n) % 100) * 4.0
        pos[:, 2] = (np.arange(n) // 100) * 4.0
        bodies = A.default_bodies(n, pos)
        loco = A.default_locomotion(n, ybot, state=abi.LOCO_RUN)
        loco["time"][:, 2] = A.crowd_phase_offsets(n, run_cycle)
        loco["motionTime"] = loco["time"][:, 2]
        intents = A.default_intents(n)
        intents["flags"] = 0
    else:
        if terrain.get("kind") == "assets":
            pos, rng = spawn_positions_on_scene(engine, n, terrain, seed)
        else:
            pos, rng = spawn_positions(n, terrain, seed)
        bodies = A.default_bodies(n, pos)
        heading = rng.uniform(0, 2 * np.pi, n)
        speed = np.where(rng.uniform(size=n) < 0.5, 4.5, 12.5)  # MovementComponent walk/run speeds, Components.swift:691-692
        vel = np.stack([np.cos(heading) * speed, np.zeros(n), np.sin(heading) * speed], -1)
        intents = A.default_intents(n, vel.astype(np.float32))
        loco = A.default_locomotion(n, ybot, state=abi.LOCO_IDLE)
        loco["time"][:, 0] = A.crowd_phase_offsets(n, ybot.profiles[ybot.profile_index("Idle")]["cycleDuration"])
        if mixed:  # config 4: states drawn from {idle, walk, run, falling}, 10 % mid-blend
            st = rng.integers(0, 4, n)
            loco["state"] = st
            loco["fromState"] = rng.integers(0, 4, n)
            blending = rng.uniform(size=n) < 0.1
            loco["flags"] |= np.where(blending, abi.LOCO_IS_BLENDING, 0).astype(np.uint32)
            loco["blendT"] = np.where(blending, rng.uniform(0, 1, n), 1.0)
            loco["idleInertia"] = np.where(blending, rng.uniform(0.05, 1, n), 0.0)
            for s in range(4):
                cyc = ybot.profiles[int(loco["profile"][0, s])]["cycleDuration"]
                loco["time"][:, s] = rng.uniform(0, cyc, n)
    engine.upload(bodies=bodies, params=params, controllers=ctrl, intents=intents, locomotion=loco, actions=actions)
    return {"bodies": bodies, "params": params, "controllers": ctrl, "intents": intents, "locomotion": loco, "actions": actions}
