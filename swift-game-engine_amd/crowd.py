"""Benchmark / test world construction on top of CharacterEngine (host side, numpy only).

Implements the concrete configs of SURVEY.md §8(d):
  config 1  one Y-Bot over the 80x80 ground quad at y=-3 (CharacterFactory.swift:77-78, DemoScene.swift:87-130)
  config 2  N clones, Running profile, steady state, per-clone phase offsets, pose + LBS only
  config 3  config 2's crowd + CCD against the static triangle mesh, seeded headings, walk/run intents
"""
import numpy as np

from . import abi, assets as A


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
