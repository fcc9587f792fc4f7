"""Host-side mirror of CollisionQueryService (Game/SceneServices.swift:33-207) over the batched collision world.

The reference keeps one `CollisionQuery` alive and, every fixed step (CollisionQueryRefreshSystem, Systems.swift:157-180),
decides between a full rebuild and a transform update + BVH refit from a per-entity snapshot of
(translation, rotation, scale, vertex count, index count, body type, collides).  This class makes the same decisions and
turns them into `rebuild_static` / `rebuild_dynamic` / `update_transforms` calls on a CharacterEngine; it also derives the
kinematic-platform list PlatformCarry needs (Systems.swift:1832-1835, 644-732) from the same entities.

A scene entity is a plain dict (the reference's TransformComponent + StaticMeshComponent + optional PhysicsBodyComponent
/ KinematicPlatformComponent on one Entity):
    id            Entity.id
    translation   [3], rotation (x, y, z, w), scale [3]      TransformComponent
    positions     [V, 3] float32, indices uint32             StaticMeshComponent.collisionMesh ?? mesh
    collides      bool (default True), dirty bool (default False), material (muS, muK, flatten), layer
    bodyType      None | abi.BODY_STATIC | BODY_KINEMATIC | BODY_DYNAMIC
    platform      bool: carries a KinematicPlatformComponent
    position / prevPosition   [3] float64 body positions (platforms: pDelta = positionF - prevPositionF)
"""
import numpy as np

from . import abi, formats as F


def _quat(e):
    return np.asarray(e.get("rotation", (0, 0, 0, 1)), np.float32)


def _vec(e, key, default):
    return np.asarray(e.get(key, default), np.float32)


def _length_squared(v):
    v = np.asarray(v, np.float32)
    acc = np.float32(0)
    for c in v:  # simd_length_squared: left-to-right for 3 lanes; 4 lanes reduce as (x0+x2)+(x1+x3)
        acc = np.float32(acc + np.float32(c * c))
    if v.shape[0] == 4:
        acc = np.float32(np.float32(v[0] * v[0] + v[2] * v[2]) + np.float32(v[1] * v[1] + v[3] * v[3]))
    return acc


class CollisionQueryService:
    def __init__(self, engine):
        self.engine = engine
        self.has_query = False          # `query != nil`
        self.dirty = True
        self.cache = {}                 # Entity id -> snapshot
        self.last_active = None
        self.slot = {}                  # Entity id -> (set, index in that set's rebuild array)
        self.log = []                   # what the last update() did: "rebuild" | ("static", ids) | ("dynamic", ids)

    def mark_dirty(self):
        self.dirty = True

    # ---- SceneServices.swift:196-206
    @staticmethod
    def _filter(world, active_ids):
        return [e for e in world if (active_ids is None or e["id"] in active_ids) and e.get("collides", True)]

    @staticmethod
    def _model_matrix(e):
        return F.model_matrix({"translation": _vec(e, "translation", (0, 0, 0)), "rotation": _quat(e), "scale": _vec(e, "scale", (1, 1, 1))})

    @staticmethod
    def _is_dynamic(e):
        return e.get("bodyType") is not None and e["bodyType"] != abi.BODY_STATIC   # partitionEntities, CollisionQuery.swift:886-900

    def _entity_desc(self, e):
        return {"positions": e["positions"], "indices": e["indices"], "modelMatrix": self._model_matrix(e),
                "material": e.get("material", (0.8, 0.6, 0)), "layer": e.get("layer", 1)}

    # ---- rebuild (:45-50)
    def rebuild(self, world, active_ids=None):
        entities = self._filter(world, active_ids)
        statics = [e for e in entities if not self._is_dynamic(e)]
        dynamics = [e for e in entities if self._is_dynamic(e)]
        self.engine.rebuild_static([self._entity_desc(e) for e in statics])
        self.engine.rebuild_dynamic([self._entity_desc(e) for e in dynamics])
        self.slot = {e["id"]: (abi.SET_STATIC, k) for k, e in enumerate(statics)}
        self.slot.update({e["id"]: (abi.SET_DYNAMIC, k) for k, e in enumerate(dynamics)})
        self.has_query = True
        self.dirty = False
        self.last_active = None if active_ids is None else set(active_ids)
        self._refresh_cache(world, active_ids)
        self.log = ["rebuild"]

    # ---- update (:52-77)
    def update(self, world, active_ids=None):
        active = None if active_ids is None else set(active_ids)
        if active != self.last_active or self.dirty or not self.has_query:
            self.rebuild(world, active_ids)
            return
        structural, static_moved, dynamic_moved = self._changes(world, active_ids)
        if structural:
            self.rebuild(world, active_ids)
            return
        self.log = []
        by_id = {e["id"]: e for e in world}
        for which, ids in ((abi.SET_STATIC, static_moved), (abi.SET_DYNAMIC, dynamic_moved)):
            if not ids:
                continue
            idx = [self.slot[i][1] for i in ids]
            mats = np.stack([self._model_matrix(by_id[i]) for i in ids])
            self.engine.update_transforms(which, idx, mats)
            self.log.append(("static" if which == abi.SET_STATIC else "dynamic", sorted(ids)))
        self._refresh_cache(world, active_ids)

    # ---- staticMeshChanges (:94-167)
    def _changes(self, world, active_ids):
        entities = self._filter(world, active_ids)
        if len(entities) != len(self.cache):
            return True, [], []
        eps = np.float32(1e-6)
        static_moved, dynamic_moved = [], []
        for e in entities:
            if e.get("dirty", False):
                return True, [], []
            snap = self.cache.get(e["id"])
            if snap is None or snap["bodyType"] != e.get("bodyType") or snap["collides"] != e.get("collides", True):
                return True, [], []
            target = dynamic_moved if self._is_dynamic(e) else static_moved
            moved = (_length_squared(_vec(e, "translation", (0, 0, 0)) - snap["translation"]) > eps
                     or _length_squared(_quat(e) - snap["rotation"]) > eps
                     or _length_squared(_vec(e, "scale", (1, 1, 1)) - snap["scale"]) > eps)
            if moved and e["id"] not in target:
                target.append(e["id"])
            if len(e["positions"]) != snap["vertexCount"] or len(e["indices"]) != snap["indexCount"]:
                return True, [], []
        return False, static_moved, dynamic_moved

    # ---- refreshStaticMeshCache (:169-194)
    def _refresh_cache(self, world, active_ids):
        self.cache = {}
        for e in self._filter(world, active_ids):
            self.cache[e["id"]] = {"translation": _vec(e, "translation", (0, 0, 0)).copy(), "rotation": _quat(e).copy(),
                                   "scale": _vec(e, "scale", (1, 1, 1)).copy(), "vertexCount": len(e["positions"]),
                                   "indexCount": len(e["indices"]), "bodyType": e.get("bodyType"), "collides": e.get("collides", True)}
            if e.get("dirty", False):
                e["dirty"] = False

    # ---- the platform list of KinematicMoveStopSystem.fixedUpdate (Systems.swift:1832-1835) ----
    def upload_platforms(self, world):
        """world.query(PhysicsBody, Transform, StaticMesh, KinematicPlatform) -> sge_platform_state array, uploaded."""
        rows = [e for e in world if e.get("platform") and e.get("bodyType") is not None]
        pf = np.zeros(len(rows), abi.platform_dtype)
        for k, e in enumerate(rows):
            pf[k]["kinematic"] = 1 if e["bodyType"] == abi.BODY_KINEMATIC else 0
            pos = np.asarray(e.get("position", e.get("translation", (0, 0, 0))), np.float64).astype(np.float32)
            prev = np.asarray(e.get("prevPosition", e.get("position", e.get("translation", (0, 0, 0)))), np.float64).astype(np.float32)
            pf[k]["delta"] = pos - prev
            if len(e["positions"]):
                mn, mx = self.engine.mesh_world_aabb(e["positions"], self._model_matrix(e))
                pf[k]["aabbMin"], pf[k]["aabbMax"], pf[k]["hasAABB"] = mn, mx, 1
        self.engine.upload_platforms(pf)
        return pf
