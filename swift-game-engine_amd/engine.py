"""Host driver above the C ABI (include/sge_amd.h): numpy arrays in, numpy arrays out.

`CharacterEngine` is written against a tiny function table so that the parity
tests can run the very same host code over the CPU oracle (tests/oracle_binding.py
supplies that table); the product table below binds libsge_amd.so only and there
is no fallback between the two.
"""
import ctypes as C

import numpy as np

from . import abi
from .abi import ptr


class SgeError(RuntimeError):
    pass


class _ProductTable:
    """sge_* entry points of the HIP library."""

    def __init__(self, device_index=0, lib_path=None):
        self.lib = abi.load_library(lib_path)
        if self.lib.sge_abi_version() != abi.SGE_ABI_VERSION:
            raise SgeError("libsge_amd.so ABI version mismatch")
        self.handle = self.lib.sge_context_create(int(device_index))
        if not self.handle:
            raise SgeError("sge_context_create failed: " + self.last_error())
        self.is_product = True

    def fn(self, name):
        return getattr(self.lib, "sge_" + name)

    def last_error(self):
        return (self.lib.sge_last_error() or b"").decode()

    def skinning_encode(self, handle, out_positions, out_normals, out_tangents, out_layout, jobs, count):
        return self.lib.sge_skinning_encode(handle, out_positions, out_normals, out_tangents, out_layout, jobs, count)

    def close(self):
        if self.handle:
            self.lib.sge_context_destroy(self.handle)
            self.handle = None


class CharacterEngine:
    """One GPU's share of the crowd: skeleton, profiles, source mesh, collision world, characters."""

    def __init__(self, device_index=0, table=None):
        self.t = table if table is not None else _ProductTable(device_index)
        self.h = self.t.handle
        self.bone_count = 0
        self.vertex_count = 0
        self.count = 0
        self._keep = []  # arrays referenced by descriptors during a call

    # -- plumbing ---------------------------------------------------------- #
    def _call(self, name, *args):
        rc = self.t.fn(name)(self.h, *args)
        if rc != abi.SGE_OK:
            raise SgeError(f"{name} failed with code {rc}: {self.t.last_error()}")

    def close(self):
        self.t.close()

    def synchronize(self):
        if self.t.is_product:
            self._call("synchronize")

    def set_option(self, option, value):
        if self.t.is_product:
            self._call("context_set_option", int(option), int(value))

    # -- skeleton ---------------------------------------------------------- #
    def build_skeleton(self, assets):
        """SkeletonLoader.buildSkeleton + Skeleton.init (host helper of the same library)."""
        B = assets.bone_count
        rest = np.zeros((B, 3), np.float32)
        bind = np.zeros((B, 16), np.float32)
        inv = np.zeros((B, 16), np.float32)
        fix = np.zeros(16, np.float32)
        rc = self.t.fn("skeleton_build")(B, ptr(assets.parent), ptr(assets.translations),
                                        ptr(assets.pre_rotation_degrees), ptr(assets.root_fix_degrees),
                                        C.c_float(assets.unit_scale), int(assets.zero_root),
                                        ptr(rest), ptr(bind), ptr(inv), ptr(fix))
        if rc != abi.SGE_OK:
            raise SgeError("skeleton_build failed")
        return {"restTranslation": rest, "bindLocal": bind, "invBindModel": inv, "rootRotationFix": fix}

    def upload_skeleton(self, assets, built=None):
        built = built or self.build_skeleton(assets)
        d = abi.SkeletonDesc()
        d.boneCount = assets.bone_count
        arrays = {
            "parent": np.ascontiguousarray(assets.parent, np.int32),
            "bindLocal": np.ascontiguousarray(built["bindLocal"], np.float32),
            "invBindModel": np.ascontiguousarray(built["invBindModel"], np.float32),
            "restTranslation": np.ascontiguousarray(built["restTranslation"], np.float32),
            "rawRestTranslation": np.ascontiguousarray(assets.translations, np.float32),
            "preRotationDegrees": np.ascontiguousarray(assets.pre_rotation_degrees, np.float32),
        }
        d.parent = arrays["parent"].ctypes.data_as(C.POINTER(C.c_int32))
        for k in ("bindLocal", "invBindModel", "restTranslation", "rawRestTranslation", "preRotationDegrees"):
            setattr(d, k, arrays[k].ctypes.data_as(C.POINTER(C.c_float)))
        d.rootRotationFix = (C.c_float * 16)(*[float(x) for x in built["rootRotationFix"]])
        d.unitScale = assets.unit_scale
        d.pelvisIndex = assets.pelvis_index
        d.leanIndex = assets.lean_index
        self._call("skeleton_upload", C.byref(d))
        self.bone_count = assets.bone_count
        self.skeleton = dict(built, parent=arrays["parent"])
        return built

    def upload_profiles(self, profiles):
        descs = (abi.MotionProfileDesc * len(profiles))()
        keep = []
        for k, p in enumerate(profiles):
            descs[k].order = p["order"]
            descs[k].cycleDuration = p["cycleDuration"]
            a = [np.ascontiguousarray(p["bonePresent"], np.uint8), np.ascontiguousarray(p["coeffCount"], np.uint8),
                 np.ascontiguousarray(p["coeffs"], np.float32)]
            keep.append(a)
            descs[k].bonePresent = a[0].ctypes.data_as(C.POINTER(C.c_uint8))
            descs[k].coeffCount = a[1].ctypes.data_as(C.POINTER(C.c_uint8))
            descs[k].coeffs = a[2].ctypes.data_as(C.POINTER(C.c_float))
        self._call("motion_profiles_upload", descs, len(profiles))

    # -- skinned mesh ------------------------------------------------------ #
    def compute_tangents(self, positions, normals, uvs, indices):
        """MeshTangents.compute (host helper of the same library)."""
        V = positions.shape[0]
        out = np.zeros((V, 4), np.float32)
        idx = np.ascontiguousarray(indices)
        i16 = idx if idx.dtype == np.uint16 else None
        i32 = idx if idx.dtype == np.uint32 else None
        rc = self.t.fn("mesh_tangents_compute")(V, ptr(positions), ptr(normals), ptr(uvs), ptr(i16), ptr(i32),
                                               int(idx.size), ptr(out))
        if rc != abi.SGE_OK:
            raise SgeError("mesh_tangents_compute failed")
        return out

    def upload_skinned_mesh(self, mesh, inv_bind_model=None):
        if "tangents" not in mesh:
            mesh = dict(mesh, tangents=self.compute_tangents(mesh["positions"], mesh["normals"], mesh["uvs"],
                                                             mesh["indices"]))
        d = abi.SkinnedMeshDesc()
        V = mesh["positions"].shape[0]
        d.vertexCount = V
        keep = {k: np.ascontiguousarray(mesh[k], np.float32) for k in ("positions", "normals", "tangents", "boneWeights")}
        keep["boneIndices"] = np.ascontiguousarray(mesh["boneIndices"], np.uint16)
        for k in ("positions", "normals", "tangents", "boneWeights"):
            setattr(d, k, keep[k].ctypes.data_as(C.POINTER(C.c_float)))
        d.boneIndices = keep["boneIndices"].ctypes.data_as(C.POINTER(C.c_uint16))
        if inv_bind_model is not None:
            ib = np.ascontiguousarray(inv_bind_model, np.float32)
            d.invBindModel = ib.ctypes.data_as(C.POINTER(C.c_float))
            d.invBindCount = ib.shape[0]
        self._call("skinned_mesh_upload", C.byref(d))
        self.vertex_count = V
        self.mesh = dict(mesh, **keep)
        return self.mesh

    def skinning_encode(self, out_positions, out_normals, out_tangents, out_layout, jobs):
        """RTSkinningEncoder.encode: out_* are device pointers (ints) on the product, host arrays' pointers on the oracle;
        jobs: list of dict(sourcePositions, sourceNormals, sourceTangents, sourceBoneIndices, sourceBoneWeights, palette (pointers),
        paletteCount, vertexCount, dstBaseVertex, sourceLayout). Asynchronous on the context's stream."""
        arr = (abi.SkinningJob * max(len(jobs), 1))()
        for k, j in enumerate(jobs):
            arr[k] = abi.SkinningJob(j["sourcePositions"], j["sourceNormals"], j["sourceTangents"], j["sourceBoneIndices"],
                                     j["sourceBoneWeights"], j["palette"], j["paletteCount"], j["vertexCount"], j["dstBaseVertex"],
                                     j.get("sourceLayout", abi.LAYOUT_PACKED))
        rc = self.t.skinning_encode(self.h, C.c_void_p(out_positions), C.c_void_p(out_normals), C.c_void_p(out_tangents), int(out_layout), arr, len(jobs))
        if rc != abi.SGE_OK:
            raise SgeError(f"skinning_encode failed with code {rc}: {self.t.last_error()}")

    # -- collision world --------------------------------------------------- #
    def rebuild_static(self, entities):
        """entities: list of dict(positions [V][3], indices u32, modelMatrix [16] (default identity),
        material (muS, muK, flatten) (default SurfaceMaterial.default), layer (default 1))."""
        self._rebuild("collision_rebuild_static", entities)

    def rebuild_dynamic(self, entities):
        """The dynamic triangle set (collidable meshes on non-static bodies): same entity dicts as rebuild_static."""
        self._rebuild("collision_rebuild_dynamic", entities)

    def update_transforms(self, which, entity_indices, model_matrices):
        """CollisionQuery.updateStaticTransforms / updateDynamicTransforms: new model matrices [n][16] for the listed
        entities of the last rebuild of set `which` (abi.SET_STATIC / abi.SET_DYNAMIC); refits the BVH."""
        idx = np.ascontiguousarray(entity_indices, np.int32)
        mats = np.ascontiguousarray(model_matrices, np.float32).reshape(-1, 16)
        assert mats.shape[0] == idx.shape[0]
        self._call("collision_update_transforms", int(which), ptr(idx), ptr(mats), idx.shape[0])

    def raycast(self, origins, directions, max_distance, mask=0xFFFFFFFF):
        """CollisionQuery.raycast, batched -> raycast_hit_dtype array."""
        o = np.asarray(origins, np.float32).reshape(-1, 3)
        q = np.zeros(o.shape[0], abi.ray_query_dtype)
        q["origin"] = o
        q["direction"] = np.asarray(directions, np.float32).reshape(-1, 3)
        q["maxDistance"] = max_distance
        q["mask"] = mask
        out = np.zeros(q.shape[0], abi.raycast_hit_dtype)
        self._call("raycast_batch", ptr(q), q.shape[0], ptr(out))
        return out

    def mesh_world_aabb(self, positions, model_matrix):
        """meshWorldAABB (Systems.swift:627-642), host helper -> (min[3], max[3])."""
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        m = np.ascontiguousarray(model_matrix, np.float32).reshape(16)
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        rc = self.t.fn("mesh_world_aabb")(ptr(pos), pos.shape[0], ptr(m), ptr(mn), ptr(mx))
        if rc != abi.SGE_OK:
            raise SgeError("mesh_world_aabb failed")
        return mn, mx

    def upload_platforms(self, platforms):
        """platforms: platform_dtype array (PlatformCarry inputs per kinematic platform entity), or None / empty."""
        p = np.zeros(0, abi.platform_dtype) if platforms is None else np.ascontiguousarray(platforms, abi.platform_dtype)
        self._call("platforms_upload", ptr(p) if p.shape[0] else None, p.shape[0])

    def _rebuild(self, fn, entities):
        descs = (abi.StaticMeshEntity * max(len(entities), 1))()
        keep = []
        for k, e in enumerate(entities):
            pos = np.ascontiguousarray(e["positions"], np.float32)
            idx = np.ascontiguousarray(e["indices"], np.uint32)
            keep += [pos, idx]
            descs[k].positions = pos.ctypes.data_as(C.POINTER(C.c_float))
            descs[k].vertexCount = pos.shape[0]
            descs[k].indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
            descs[k].indexCount = idx.size
            m = np.asarray(e.get("modelMatrix", np.eye(4, dtype=np.float32).reshape(16)), np.float32).reshape(16)
            descs[k].modelMatrix = (C.c_float * 16)(*[float(x) for x in m])
            mu = e.get("material", (0.8, 0.6, 0))
            descs[k].material = abi.SurfaceMaterial(mu[0], mu[1], int(mu[2]))
            tm = e.get("triangleMaterials")
            if tm is not None:
                tm = np.ascontiguousarray(tm, abi.material_dtype)
                keep.append(tm)
                descs[k].triangleMaterials = tm.ctypes.data_as(C.POINTER(abi.SurfaceMaterial))
                descs[k].triangleMaterialCount = tm.shape[0]
            descs[k].collisionLayer = int(e.get("layer", 1))
        self._call(fn, descs, len(entities))

    def collision_counts(self, which=abi.SET_STATIC):
        v, t, n = C.c_int32(), C.c_int32(), C.c_int32()
        self._call("collision_counts_set", int(which), C.byref(v), C.byref(t), C.byref(n))
        return v.value, t.value, n.value

    def collision_copy(self, which=abi.SET_STATIC):
        v, t, n = self.collision_counts(which)
        out = {"positions": np.zeros((v, 3), np.float32), "indices": np.zeros(t * 3, np.uint32),
               "aabbs": np.zeros((t, 2, 3), np.float32), "nodes": np.zeros(n, abi.bvh_node_dtype),
               "triOrder": np.zeros(t, np.int32), "triLeaf": np.zeros(t, np.int32)}
        self._call("collision_copy_set", int(which), ptr(out["positions"]), ptr(out["indices"]), ptr(out["aabbs"]),
                   ptr(out["nodes"]), ptr(out["triOrder"]), ptr(out["triLeaf"]))
        return out

    def capsule_cast(self, queries):
        q = np.ascontiguousarray(queries, abi.query_dtype)
        out = np.zeros(q.shape[0], abi.cast_hit_dtype)
        self._call("capsule_cast_batch", ptr(q), q.shape[0], ptr(out))
        return out

    def capsule_overlap_all(self, queries, max_hits=8):
        q = np.ascontiguousarray(queries, abi.query_dtype)
        out = np.zeros((q.shape[0], max_hits), abi.overlap_hit_dtype)
        counts = np.zeros(q.shape[0], np.int32)
        self._call("capsule_overlap_all_batch", ptr(q), q.shape[0], int(max_hits), ptr(out), ptr(counts))
        return out, counts

    def capsule_overlap(self, queries):
        """CollisionQuery.capsuleOverlap: deepest hit per query (found flag 0 = nil)."""
        q = np.ascontiguousarray(queries, abi.query_dtype)
        out = np.zeros(q.shape[0], abi.overlap_hit_dtype)
        found = np.zeros(q.shape[0], np.int32)
        self._call("capsule_overlap_batch", ptr(q), q.shape[0], ptr(out), ptr(found))
        return out, found

    # -- characters -------------------------------------------------------- #
    def resize(self, n):
        self._call("characters_resize", int(n))
        self.count = int(n)

    def upload(self, first=0, bodies=None, params=None, controllers=None, intents=None, locomotion=None, actions=None):
        arrs = []
        n = None
        for a, dt in ((bodies, abi.body_dtype), (params, abi.params_dtype), (controllers, abi.controller_dtype),
                      (intents, abi.intent_dtype), (locomotion, abi.locomotion_dtype), (actions, abi.action_dtype)):
            if a is None:
                arrs.append(None)
                continue
            a = np.ascontiguousarray(a, dt)
            n = a.shape[0] if n is None else n
            assert a.shape[0] == n
            arrs.append(a)
        if n is None:
            return
        self._call("characters_upload", int(first), int(n), *[ptr(a) for a in arrs])

    def download(self, first=0, count=None, what=("bodies", "params", "controllers", "intents", "locomotion", "actions")):
        count = self.count - first if count is None else count
        dts = {"bodies": abi.body_dtype, "params": abi.params_dtype, "controllers": abi.controller_dtype,
               "intents": abi.intent_dtype, "locomotion": abi.locomotion_dtype, "actions": abi.action_dtype}
        out = {k: (np.zeros(count, dts[k]) if k in what else None) for k in dts}
        self._call("characters_download", int(first), int(count), *[ptr(out[k]) for k in dts])
        return {k: v for k, v in out.items() if v is not None}

    # -- asynchronous World synchronisation (pinned, event-ordered; product only) -- #
    _STATE = (("bodies", abi.STATE_BODIES, abi.body_dtype), ("controllers", abi.STATE_CONTROLLERS, abi.controller_dtype),
              ("locomotion", abi.STATE_LOCOMOTION, abi.locomotion_dtype), ("actions", abi.STATE_ACTIONS, abi.action_dtype),
              ("intents", abi.STATE_INTENTS, abi.intent_dtype))

    @classmethod
    def _view_arrays(cls, view):
        """numpy views (no copy) of the pinned arrays a sge_state_view names."""
        out = {}
        for name, bit, dt in cls._STATE:
            p = getattr(view, name)
            if p and view.count > 0:
                buf = (C.c_char * (view.count * dt.itemsize)).from_address(p)
                out[name] = np.frombuffer(buf, dtype=dt, count=view.count)
        return out

    def state_pull_async(self, which=abi.STATE_WORLD, first=0, count=0):
        """Snapshot of the selected arrays behind everything enqueued so far, on its way to pinned host memory -> ticket."""
        t = C.c_int32(-1)
        self._call("state_pull_async", int(which), int(first), int(count), C.byref(t))
        return t.value

    def state_wait(self, ticket):
        """Blocks until the pull has landed -> dict of numpy views of the context's pinned memory (valid until the pull after next)."""
        v = abi.StateView()
        self._call("state_wait", int(ticket), C.byref(v))
        return self._view_arrays(v)

    def state_ready(self, ticket):
        rc = self.t.fn("state_poll")(self.h, int(ticket))
        if rc not in (abi.SGE_OK, abi.SGE_ERR_NOT_READY):
            raise SgeError(f"state_poll failed with code {rc}: {self.t.last_error()}")
        return rc == abi.SGE_OK

    def state_push_begin(self, which=abi.STATE_INTENTS, first=0, count=0):
        """Pinned staging arrays (numpy views) for the host to fill; state_push_commit() enqueues the copies in front of the next tick."""
        v = abi.StateView()
        self._call("state_push_begin", int(which), int(first), int(count), C.byref(v))
        return self._view_arrays(v)

    def state_push_commit(self):
        self._call("state_push_commit")

    def palettes(self, first=0, count=None, model=False, local=False):
        count = self.count - first if count is None else count
        pal = np.zeros((count, self.bone_count, 16), np.float32)
        mod = np.zeros_like(pal) if model else None
        loc = np.zeros_like(pal) if local else None
        self._call("palettes_download", int(first), int(count), ptr(pal), ptr(mod), ptr(loc))
        return pal, mod, loc

    def skinned(self, first_vertex=0, vertex_count=None, positions=True, normals=True, tangents=True):
        vertex_count = self.count * self.vertex_count - first_vertex if vertex_count is None else vertex_count
        p = np.zeros((vertex_count, 3), np.float32) if positions else None
        n = np.zeros((vertex_count, 3), np.float32) if normals else None
        t = np.zeros((vertex_count, 4), np.float32) if tangents else None
        self._call("skinned_download", int(first_vertex), int(vertex_count), ptr(p), ptr(n), ptr(t))
        return p, n, t

    def tick(self, dt=1.0 / 60.0, stages=abi.STAGE_ALL, gravity=(0.0, -98.0, 0.0), first=0, count=0):
        d = abi.TickDesc()
        d.dt = dt
        d.gravity = (C.c_float * 3)(*gravity)
        d.stages = stages
        d.first, d.count = first, count
        self._call("tick", C.byref(d))

    # -- skinned-geometry acceleration structure (RTAccelerationBuilder.swift:75-145) -- #
    @staticmethod
    def blas_topology(positions, indices, lib=None):
        """Host helper (no GPU): the wide-BVH topology sge_blas_build derives from a mesh -> dict of arrays + 'info'."""
        lib = lib or abi.load_library()
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        info = abi.BlasInfo()
        args = (ptr(pos), pos.shape[0], ptr(idx), idx.shape[0], C.byref(info))
        if lib.sge_blas_topology(*args, None, None, None, None, None, None) != abi.SGE_OK:
            raise SgeError("blas_topology failed: " + lib.sge_last_error().decode())
        out = {"info": info,
               "entryLink": np.zeros((info.entryCount, 2), np.int32), "wideFirst": np.zeros(info.wideCount + 1, np.int32),
               "wideParentEntry": np.zeros(info.wideCount, np.int32), "slotTriangle": np.zeros(info.triangleCount, np.uint32),
               "vertexEntryStart": np.zeros(pos.shape[0] + 1, np.int32), "vertexEntries": np.zeros(info.incidenceCount, np.int32)}
        rc = lib.sge_blas_topology(*args, ptr(out["entryLink"]), ptr(out["wideFirst"]), ptr(out["wideParentEntry"]),
                                   ptr(out["slotTriangle"]), ptr(out["vertexEntryStart"]), ptr(out["vertexEntries"]))
        if rc != abi.SGE_OK:
            raise SgeError("blas_topology failed")
        return out

    def blas_build(self, indices):
        """encoder.build for the crowd's shared mesh; indices = the item's slice of the dynamic index buffer."""
        idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        self._call("blas_build", ptr(idx), idx.shape[0])
        self.blas_indices = idx
        if self.t.is_product:
            info = abi.BlasInfo()
            self._call("blas_info_get", C.byref(info))
            self.blas_info = info
            return info
        return None

    def blas_set_uvs(self, uvs):
        """The item's slice of the dynamic UV buffer: hits then carry interp_uv."""
        u = np.ascontiguousarray(uvs, np.float32).reshape(-1, 2)
        self._call("blas_set_uvs", ptr(u), u.shape[0])

    def blas_refit(self, first=0, count=None):
        """encoder.refit(options: .vertexData) over the context's skinned positions. Asynchronous."""
        self._call("blas_refit", first, self.count - first if count is None else count)

    def blas_bounds(self, first=0, count=None):
        count = self.count - first if count is None else count
        out = np.zeros((count, self.blas_info.entryCount + 1, 6), np.float32)
        self._call("blas_bounds_download", first, count, ptr(out))
        return out

    def blas_instances(self, model_matrices, first=0):
        m = np.ascontiguousarray(model_matrices, np.float32).reshape(-1, 16)
        self._call("blas_instances_upload", first, m.shape[0], ptr(m))

    def blas_intersect(self, origins, directions, instances, min_distance=0.001, max_distance=1e6):
        """`isect.intersect(ray, accel)` restricted to one instance per ray + the kernel's reads at the hit -> blas_hit_dtype."""
        o = np.asarray(origins, np.float32).reshape(-1, 3)
        r = np.zeros(o.shape[0], abi.blas_ray_dtype)
        r["origin"] = o
        r["direction"] = np.asarray(directions, np.float32).reshape(-1, 3)
        r["minDistance"] = min_distance
        r["maxDistance"] = max_distance
        r["instance"] = instances
        out = np.zeros(r.shape[0], abi.blas_hit_dtype)
        self._call("blas_intersect_batch", ptr(r), r.shape[0], ptr(out))
        return out

    def blas_intersect_device(self, d_rays, count, d_hits, any_instance=True):
        """Rays / hits already in device memory (blas_ray_dtype / blas_hit_dtype records); asynchronous on the context's stream."""
        self._call("blas_intersect_device", C.c_void_p(d_rays), int(count), C.c_void_p(d_hits), int(bool(any_instance)))

    def blas_profile(self, reset=True):
        ms, n = C.c_double(0), C.c_int64(0)
        self._call("blas_profile_read", C.byref(ms), C.byref(n), int(reset))
        return ms.value, n.value

    # -- AgentSeparationSystem (Systems.swift:1906-2210) ---------------------- #
    def separation_params(self, iterations=2, separation_margin=0.2, height_margin=0.1):
        """AgentSeparationSystem.init(iterations:separationMargin:heightMargin:); the stage itself is abi.STAGE_SEPARATION of tick()."""
        self._call("separation_params", int(iterations), C.c_float(separation_margin), C.c_float(height_margin))

    # -- agents (config 5) -------------------------------------------------- #
    def agents_export(self, out_ptr):
        """Packs this engine's characters as AgentSweepState[count] at `out_ptr`
        (device pointer for the product, host pointer for the test oracle)."""
        self._call("agents_export", C.c_void_p(out_ptr))

    def agents_import(self, all_ptr, total, self_offset):
        self._call("agents_import", C.c_void_p(all_ptr), int(total), int(self_offset))

    def agents_allgather(self, nccl_comm=None, rank=0, world_size=1, slot=None):
        """export -> ncclAllGather on the context's stream -> import, in one stream-ordered call (product only);
        nccl_comm: the caller's ncclComm_t as an integer / c_void_p (None with world_size 1)."""
        self._call("agents_allgather", C.c_void_p(nccl_comm), int(rank), int(world_size), int(self.count if slot is None else slot))

    def stream_handle(self):
        """The hipStream_t the context enqueues on, as an integer (product only)."""
        p = C.c_void_p()
        self._call("context_get_stream", C.byref(p))
        return p.value or 0

    # -- diagnostics ------------------------------------------------------- #
    def profile_read(self, reset=True):
        st = abi.StageTimes()
        self._call("profile_read", C.byref(st), int(reset))
        return st

    def skin_launch_times(self):
        """HIP-event duration (ms) of every skin launch since the last profile reset (product only, OPT_PROFILE on)."""
        n = C.c_int32(0)
        self._call("debug_skin_launch_times", None, 0, C.byref(n))
        out = np.zeros(max(n.value, 1), np.float32)
        self._call("debug_skin_launch_times", ptr(out), int(out.shape[0]), C.byref(n))
        return out[: n.value]

    def placement(self):
        """(ms of one three-stream store pass over the kept placement of the skinned output streams, placements timed)."""
        ms, tried = C.c_float(0), C.c_int32(0)
        self._call("debug_placement", C.byref(ms), C.byref(tried))
        return ms.value, tried.value

    def move_cost(self, first=0, count=None):
        """Distance evaluations each character spent in the casts of its last fixed step (product only)."""
        count = self.count - first if count is None else count
        out = np.zeros(count, np.int32)
        self._call("move_cost_read", int(first), int(count), ptr(out))
        return out

    def move_stats(self, reset=True):
        st = abi.MoveStats()
        self._call("move_stats_read", C.byref(st), int(reset))
        return st


def make_queries(origins, deltas=None, radius=1.5, half_height=1.0, mode=abi.CAST, min_normal_y=0.5, mask=0xFFFFFFFF):
    origins = np.asarray(origins, np.float32).reshape(-1, 3)
    q = np.zeros(origins.shape[0], abi.query_dtype)
    q["origin"] = origins
    if deltas is not None:
        q["delta"] = np.asarray(deltas, np.float32).reshape(-1, 3)
    q["radius"], q["halfHeight"], q["minNormalY"], q["mask"], q["mode"] = radius, half_height, min_normal_y, mask, mode
    return q
