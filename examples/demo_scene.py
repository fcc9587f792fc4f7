#!/usr/bin/env python3
"""The physical content of the reference's DemoScene (Game/DemoScene.swift) on the MI355X path, system by system:

  ground plane 80 x 80 at y = -3                                  DemoScene.swift:103-130
  ornate mirror: its two collision hulls, x8, offset (-10, 1, 4)  :296-376 (17-Cheese / Semla hulls need Blender)
  two kinematic platforms (box 4.0 scaled (1.5, 0.2, 1.5))        :379-452: elevator at (16, -1, 0), mover at (-16, -2, 12)
  the player: a Y-Bot with the CharacterFactory defaults          CharacterFactory.swift:60-110, dropped at (0, 7.5, 0)

and the fixed-step order of DemoScene.swift:56-75:
  KinematicPlatformMotionSystem -> CollisionQueryRefreshSystem -> PhysicsIntentSystem -> GravitySystem ->
  KinematicMoveStopSystem -> LocomotionProfileSystem -> ActionAnimationSystem -> PoseStackSystem -> PhysicsWritebackSystem
  (+ the skinning encode of RayTracingScene.buildGeometryBuffers).

and what the renderer does next with the skinned vertices (RayTracingScene.buildGeometryBuffers ->
RTAccelerationBuilder.build, RTAccelerationBuilder.swift:75-185; raytraceKernel primary rays, RayTracing.metalinc:225-300):
the player's acceleration structure is refitted behind the skinning of every step, its instance matrix is the entity's
TransformComponent, and `--render` shoots a small image of primary rays at it and prints the depth as ASCII.

Usage:  python examples/demo_scene.py [--steps 600] [--oracle] [--render]   (--oracle runs the CPU checker instead of the GPU)
The player walks to the ground mover, rides it, then heads for the elevator.
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sge = importlib.import_module("swift-game-engine_amd")
A, F = sge.abi, sge.formats


def box_mesh(size=4.0):
    """ProceduralMeshes.box (ProceduralMeshes.swift:183-230): 24 vertices, 12 triangles."""
    s = size * 0.5
    faces = [((0, 0, 1), [(-s, -s, s), (s, -s, s), (s, s, s), (-s, s, s)]), ((0, 0, -1), [(s, -s, -s), (-s, -s, -s), (-s, s, -s), (s, s, -s)]),
             ((1, 0, 0), [(s, -s, s), (s, -s, -s), (s, s, -s), (s, s, s)]), ((-1, 0, 0), [(-s, -s, -s), (-s, -s, s), (-s, s, s), (-s, s, -s)]),
             ((0, 1, 0), [(-s, s, s), (s, s, s), (s, s, -s), (-s, s, -s)]), ((0, -1, 0), [(-s, -s, -s), (s, -s, -s), (s, -s, s), (-s, -s, s)])]
    pos, idx = [], []
    for _, quad in faces:
        b = len(pos)
        pos += quad
        idx += [b, b + 1, b + 2, b, b + 2, b + 3]
    return np.asarray(pos, np.float32), np.asarray(idx, np.uint32)


class Platform:
    """KinematicPlatformComponent + KinematicPlatformMotionSystem (Components.swift:484-505, Systems.swift:122-155)."""

    def __init__(self, entity, origin, axis, amplitude, speed, phase):
        self.e, self.origin, self.amplitude, self.speed, self.phase, self.time = entity, np.asarray(origin, np.float32), amplitude, speed, phase, np.float32(0)
        ax = np.asarray(axis, np.float32)
        n = np.float32(np.sqrt((ax * ax).sum()))
        self.axis = ax / n if n > 1e-4 else np.array([0, 1, 0], np.float32)

    def step(self, dt):
        self.time = np.float32(self.time + np.float32(dt))
        offset = np.float32(np.sin(np.float32(self.time * np.float32(self.speed) + np.float32(self.phase)))) * np.float32(self.amplitude)
        new = self.origin + self.axis * offset
        self.e["prevPosition"] = self.e["position"]     # PhysicsBeginStepSystem (Systems.swift:183-202)
        self.e["translation"] = tuple(new)
        self.e["position"] = tuple(float(v) for v in new)


def build(engine):
    ybot = sge.assets.YBotAssets()
    sge.crowd.upload_ybot_mesh(engine, ybot)
    gp, gi, _ = sge.assets.ground_plane()
    ident = (0, 0, 0, 1)
    world = [{"id": 1, "translation": (0, -3, 0), "rotation": ident, "scale": (1, 1, 1), "positions": gp, "indices": gi,
              "bodyType": A.BODY_STATIC, "material": (0.9, 0.8, 0)}]
    z = np.load(os.path.join(sge.assets.GOLDEN_DIR, "ornate_mirror_static.npz"))
    t = F.transform_from_matrix(F.matrix_from_array_row_major(z["transformRowMajor"]))
    upright = F.quat_angle_axis(np.float32(float.fromhex("0x1.921fb4p+1")) * np.float32(0.5), (1, 0, 0))
    flip = F.quat_angle_axis(np.float32(float.fromhex("0x1.921fb4p+1")), (1, 0, 0))
    t["rotation"] = F.quat_mul(t["rotation"], F.quat_mul(upright, flip))       # DemoScene.swift:331-335
    t["scale"] = t["scale"] * np.float32(8.0)
    t["translation"] = t["translation"] + np.array([-10, 1, 4], np.float32)
    for k in range(2):
        world.append({"id": 10 + k, "translation": tuple(t["translation"]), "rotation": tuple(t["rotation"]), "scale": tuple(t["scale"]),
                      "positions": z[f"hull{k}.positions"], "indices": z[f"hull{k}.indices"], "bodyType": A.BODY_STATIC,
                      "material": (0.6, 0.5, 0), "layer": 1 << 4})
    bp, bi = box_mesh(4.0)
    platforms = []
    for eid, origin, axis, amp, speed, phase in ((20, (16, -1.0, 0), (0, 1, 0), 2.0, 1.1, 0.0), (21, (-16, -2.0, 12), (1, 0, 0), 4.0, 0.9, 0.7)):
        e = {"id": eid, "translation": origin, "rotation": ident, "scale": (1.5, 0.2, 1.5), "positions": bp, "indices": bi,
             "bodyType": A.BODY_KINEMATIC, "platform": True, "position": origin, "prevPosition": origin, "material": (0.9, 0.7, 0)}
        world.append(e)
        platforms.append(Platform(e, origin, axis, amp, speed, phase))
    service = sge.services.CollisionQueryService(engine)
    service.rebuild(world)
    engine.resize(1)
    state = {"bodies": sge.assets.default_bodies(1, np.array([[0.0, 7.5, 0.0]])), "params": sge.assets.default_controller_params(1),
             "controllers": sge.assets.default_controller_state(1), "intents": sge.assets.default_intents(1),
             "locomotion": sge.assets.default_locomotion(1, ybot), "actions": sge.assets.default_actions(1, ybot, present=True)}
    engine.upload(**state)
    engine.blas_build(engine.mesh["indices"])           # encoder.build of the skinned item (RTAccelerationBuilder.swift:75-112)
    return ybot, world, platforms, service


def render_probe(engine, width=56, height=28, fov_y=0.6):
    """Primary rays of the raytraceKernel (RayTracing.metalinc:225-235: origin = camera, direction through the pixel centre,
    min_distance 0.001, max_distance 1e6) against the skinned items (here: the player) -> (hit mask, distances, hit records)."""
    b = engine.download(what=("bodies",))["bodies"]
    pos = b["position"][0].astype(np.float32)
    m = F.model_matrix({"translation": pos, "rotation": b["transformRotation"][0], "scale": np.ones(3, np.float32)})
    engine.blas_instances(np.asarray(m, np.float32).reshape(1, 16))   # instance descriptor = item.modelMatrix (:168-185)
    eye = pos + np.array([0.0, 0.3, 9.0], np.float32)
    target = pos + np.array([0.0, -0.2, 0.0], np.float32)
    fwd = (target - eye) / np.linalg.norm(target - eye)
    right = np.cross(fwd, (0, 1, 0)); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    ys, xs = np.meshgrid((np.arange(height) + 0.5) / height, (np.arange(width) + 0.5) / width, indexing="ij")
    th = np.tan(fov_y / 2)
    d = fwd + ((xs * 2 - 1) * th * width / height * 0.5)[..., None] * right + ((1 - ys * 2) * th)[..., None] * up
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).reshape(-1, 3).astype(np.float32)
    h = engine.blas_intersect(np.tile(eye, (len(d), 1)), d, np.full(len(d), -1, np.int32))   # instance < 0: against every skinned item
    return h["hit"].reshape(height, width) == 1, h["distance"].reshape(height, width), h


def ascii_depth(mask, dist):
    ramp = "@%#*+=-:."
    lo, hi = (dist[mask].min(), dist[mask].max()) if mask.any() else (0.0, 1.0)
    rows = []
    for r in range(mask.shape[0]):
        rows.append("".join(ramp[min(len(ramp) - 1, int((dist[r, c] - lo) / max(hi - lo, 1e-6) * len(ramp)))] if mask[r, c] else " "
                            for c in range(mask.shape[1])))
    return "\n".join(rows)


def run(engine, steps=600, dt=1.0 / 60.0, log=None):
    ybot, world, platforms, service = build(engine)
    trace = []
    for s in range(steps):
        for p in platforms:
            p.step(dt)                                  # KinematicPlatformMotionSystem
        service.update(world)                           # CollisionQueryRefreshSystem -> updateDynamicTransforms + refit
        service.upload_platforms(world)                 # the platform list KinematicMoveStopSystem reads
        pos = engine.download(what=("bodies",))["bodies"]["position"][0]
        mover = np.asarray(world[-1]["translation"], np.float64)
        target = mover + (0, 0, 0) if s < 420 else np.array([16.0, 0, 0])   # walk to the mover, later to the elevator
        d = np.array([target[0] - pos[0], 0.0, target[2] - pos[2]])
        dist = np.linalg.norm(d)
        v = d / dist * min(4.5, dist * 4) if dist > 0.05 else np.zeros(3)
        engine.upload(intents=sge.assets.default_intents(1, v.astype(np.float32)[None]))
        engine.tick(dt=dt, stages=A.STAGE_ALL | A.STAGE_BLAS_REFIT)   # ... skinning encode, then the acceleration-structure refit
        if s % 30 == 29 or s == steps - 1:
            b = engine.download(what=("bodies", "controllers", "locomotion"))
            trace.append((s + 1, b["bodies"]["position"][0].copy(), int(b["controllers"]["flags"][0]), int(b["controllers"]["groundTriangleIndex"][0]),
                          int(b["locomotion"]["state"][0])))
            if log:
                log("step %4d  player (%7.3f %7.3f %7.3f)  flags %x  ground triangle %4d  locomotion %d  mover x %.3f  elevator y %.3f" % (
                    s + 1, *trace[-1][1], trace[-1][2], trace[-1][3], trace[-1][4], world[-1]["translation"][0], world[-2]["translation"][1]))
    return trace, engine.skinned()[0]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--oracle", action="store_true", help="run the CPU checker (tests/oracle_binding.py) instead of the GPU library")
    ap.add_argument("--render", action="store_true", help="shoot primary rays at the player after the last step and print the depth image")
    args = ap.parse_args()
    if args.oracle:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding
        eng = oracle_binding.oracle_engine()
    else:
        eng = sge.CharacterEngine(0)
    trace, skinned = run(eng, args.steps, log=print)
    print("skinned vertices: %d, bounds %s .. %s" % (len(skinned), skinned.min(0).round(3), skinned.max(0).round(3)))
    if args.render:
        mask, dist, hits = render_probe(eng)
        print("primary rays: %d of %d hit the player, distances %.3f .. %.3f" % (mask.sum(), mask.size, dist[mask].min(), dist[mask].max()))
        print(ascii_depth(mask, dist))
