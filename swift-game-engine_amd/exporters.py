"""FBX -> the engine's asset JSON payloads, without Blender.

Follows the reference's two exporter scripts step for step on the data a binary FBX holds:
  export_static_mesh(scene)   Tools/FbxToStaticMeshJson/export_static_mesh_json.py:135-229
  export_skinned_mesh(scene)  Tools/FbxToSkinnedJson/export_skinned_json.py:110-230
The scripts run inside Blender after `bpy.ops.import_scene.fbx`; what that import contributes is restated
here from the file itself: mesh-local vertex coordinates are the FBX control points, loop normals are
the file's normals, `matrix_world` is axis-conversion * unit-scale * the FBX node's global transform,
bone rest matrices in armature space are the skin clusters' TransformLink, vertex groups are the clusters'
Indexes/Weights.  Pinned against the one exporter output present in the reference checkout
(Game/ornate_mirror.static.json, see tests/test_formats.py).  Collision hulls (:84-132) follow the script's steps — loose
parts, the two largest, one convex hull each, simplified when it has more than 24 faces — with qhull for the hull and a
convex vertex subset for the simplification: they are valid hulls of the same parts, not Blender's decimate output vertex
for vertex (`build_collision_hulls`).
"""
import numpy as np

from . import fbx


def _q(v):
    """q() of the exporters: int(round(v * 1e6)) with Python's round-half-even."""
    return np.rint(np.asarray(v, np.float64) * 1000000.0).astype(np.int64)


def _f32(a):
    """Blender stores coordinates, normals, uvs and weights as float32; the scripts read them back from there."""
    return np.asarray(a, np.float64).astype(np.float32).astype(np.float64)


def blender_global_matrix(scene):
    """The object-space correction Blender's FBX import applies: axis conversion to Z-up/-Y-forward times
    UnitScaleFactor/100 (FBX unit = cm * factor, Blender unit = m)."""
    up = int(scene.settings.get("UpAxis", [1])[0])
    unit = float(scene.settings.get("UnitScaleFactor", [1.0])[0])
    M = np.eye(4)
    if up == 1:  # Y-up file: (x, y, z) -> (x, -z, y)
        M[:3, :3] = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], np.float64)
    elif up == 0:  # X-up: (x, y, z) -> (y?, ...) not present in the assets; treat like Blender's X-up -> Z-up
        M[:3, :3] = np.array([[0, 0, -1], [0, 1, 0], [1, 0, 0]], np.float64)
    S = np.diag([unit / 100.0] * 3 + [1.0])
    return M @ S


def _loop_triangles(geom):
    """(tri loops [T,3], face index [T]) — Blender's loop triangles."""
    return fbx.triangulate_blender(geom["points"], geom["loops"], geom["poly_start"])


def _unit(n):
    length = np.linalg.norm(n, axis=-1, keepdims=True)
    return np.where(length > 0, n / np.where(length > 0, length, 1.0), n)


def _face_normals_for_loops(geom):
    """Fallback when a geometry carries no normals (Semla.fbx): Blender then shades from face/vertex normals;
    use the area-weighted vertex normal of the control point."""
    pts, loops, ps = geom["points"], geom["loops"], geom["poly_start"]
    tris, _ = fbx.triangulate_fan(ps)
    cp = loops[tris]
    fn = np.cross(pts[cp[:, 1]] - pts[cp[:, 0]], pts[cp[:, 2]] - pts[cp[:, 0]])
    vn = np.zeros_like(pts)
    for k in range(3):
        np.add.at(vn, cp[:, k], fn)
    return _unit(vn)[loops]


def _material_names(scene, model_id):
    names = [scene.name(m) for m in scene.children(model_id, "Material")]
    return names


def _weld(keys):
    """First-appearance weld: returns (unique row index per input row, first input row of every unique)."""
    index = {}
    remap = np.empty(len(keys), np.int64)
    firsts = []
    for i, k in enumerate(map(bytes, np.ascontiguousarray(keys))):
        j = index.get(k)
        if j is None:
            j = len(firsts)
            index[k] = j
            firsts.append(i)
        remap[i] = j
    return remap, np.asarray(firsts, np.int64)


def _submesh_order(tri_material, material_names):
    order, names = [], []
    for m in tri_material:
        name = material_names[m] if m < len(material_names) else "Default"
        if name not in names:
            names.append(name)
        order.append(names.index(name))
    return np.asarray(order, np.int64), names


MAX_HULLS_PER_PART = 2       # export_static_mesh_json.py:7
TARGET_FACES_PER_HULL = 24   # export_static_mesh_json.py:8


def _loose_parts(vertex_count, polygons):
    """Connected components over shared vertices (bpy.ops.mesh.separate(type="LOOSE")), as lists of vertex indices in
    order of their smallest vertex; vertices no polygon uses form no part."""
    parent = np.arange(vertex_count)

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    used = np.zeros(vertex_count, bool)
    for poly in polygons:
        used[poly] = True
        r = find(poly[0])
        for v in poly[1:]:
            q = find(v)
            if q != r:
                parent[q] = r
    roots = np.array([find(i) for i in range(vertex_count)])
    parts = {}
    for i in np.flatnonzero(used):
        parts.setdefault(roots[i], []).append(i)
    return sorted(parts.values(), key=lambda p: p[0])


def _hull(points):
    """(vertices, outward triangles, merged face count) of the convex hull, or None for a degenerate point set."""
    from scipy.spatial import ConvexHull, QhullError
    if len(points) < 4:
        return None
    try:
        h = ConvexHull(points)
    except (QhullError, ValueError):
        return None
    tris = h.simplices.copy()
    a, b, c = points[tris[:, 0]], points[tris[:, 1]], points[tris[:, 2]]
    flip = np.einsum("ij,ij->i", np.cross(b - a, c - a), h.equations[:, :3]) < 0
    tris[flip] = tris[flip][:, ::-1]
    # coplanar triangles form one face (Blender's convex_hull joins them; the script counts polygons, :118)
    planes = np.round(h.equations / max(np.abs(h.equations[:, 3]).max(), 1.0), 6)
    # ... and, with its default join_triangles, also pairs of nearly coplanar triangles into quads (thresholds of 40 degrees): on a
    # finely tessellated hull about 0.6 polygons per triangle remain (the reference's own mirror hulls, 40 and 36 triangles for a
    # target of 24 faces, say 0.60 and 0.67)
    faces = min(len({tuple(p) for p in planes}), int(np.ceil(0.6 * len(tris))))
    return h.vertices, tris, faces


def convex_hull_part(points, target_faces=TARGET_FACES_PER_HULL):
    """One part's hull (:106-121): the convex hull of its vertices; when it has more than `target_faces` faces it is reduced by
    the script's ratio target / faces — here by keeping the subset of hull vertices that preserves most of the volume (greedy:
    the six axis extremes, then always the vertex farthest outside the current polytope), so the result is convex and
    inscribed. -> (positions [H,3] float32, indices [T*3] uint32) or None."""
    pts = np.asarray(points, np.float64).reshape(-1, 3)
    full = _hull(pts)
    if full is None:
        return None
    verts, tris, faces = full
    if faces > target_faces:
        ratio = max(min(target_faces / max(faces, 1), 1.0), 0.01)
        want = max(int(round(ratio * len(tris))), 4)
        cand = pts[verts]
        chosen = list(dict.fromkeys(int(i) for ax in range(3) for i in (cand[:, ax].argmin(), cand[:, ax].argmax())))
        cur = None
        while True:
            cur = _hull(cand[chosen]) if len(chosen) >= 4 else None
            if cur is not None and len(cur[1]) >= want:
                break
            rest = [i for i in range(len(cand)) if i not in chosen]
            if not rest:
                break
            if cur is None:
                far = max(rest, key=lambda i: np.linalg.norm(cand[i] - cand[chosen].mean(0)))
            else:
                from scipy.spatial import ConvexHull
                eq = ConvexHull(cand[chosen]).equations
                out = (cand[rest] @ eq[:, :3].T + eq[:, 3]).max(axis=1)
                far = rest[int(out.argmax())]
            chosen.append(far)
        if cur is not None:
            sub = cand[chosen]
            verts, tris = cur[0], cur[1]
            pts = sub
    order = np.sort(verts)
    remap = -np.ones(len(pts), np.int64)
    remap[order] = np.arange(len(order))
    return pts[order].astype(np.float32), remap[tris].reshape(-1).astype(np.uint32)


def build_collision_hulls(points, polygons):
    """_build_collision_hulls (export_static_mesh_json.py:84-132) on mesh-local vertices and polygon vertex lists."""
    pts = np.asarray(points, np.float64).reshape(-1, 3)
    parts = _loose_parts(len(pts), polygons) or [list(range(len(pts)))]
    if len(parts) > MAX_HULLS_PER_PART:
        parts = sorted(parts, key=len, reverse=True)[:MAX_HULLS_PER_PART]
    out = []
    for part in parts:
        h = convex_hull_part(pts[part])
        if h is not None and len(h[1]):
            out.append({"positions": h[0].reshape(-1), "indices": h[1]})
    return out


def export_static_mesh(scene, flip_v=True, hulls=True):
    """-> payload dict in the *.static.json schema (StaticMeshLoader.swift:163-197), arrays as numpy."""
    G = blender_global_matrix(scene)
    models = sorted(scene.of_kind("Model", "Mesh"), key=lambda m: scene.name(m).lower())
    meshes = []
    for mid in models:
        geoms = scene.children(mid, "Geometry")
        if not geoms:
            continue
        geom = fbx.geometry_arrays(scene.objects[geoms[0]])
        tris, face_of = _loop_triangles(geom)
        loops = tris.reshape(-1)
        cp = geom["loops"][loops]
        p = _f32(geom["points"])[cp]
        n = geom["normals"]
        n = _f32(n)[loops] if n is not None else _face_normals_for_loops(geom)[loops]
        n = _f32(_unit(n))
        if geom["uvs"] is not None:
            uv = _f32(geom["uvs"])[loops]
        else:
            uv = np.zeros((len(loops), 2))
        if flip_v:
            uv = np.stack([uv[:, 0], 1.0 - uv[:, 1]], axis=1)
        keys = np.concatenate([_q(p), _q(n), _q(uv)], axis=1)
        remap, firsts = _weld(keys)
        mats = geom["materials"] if geom["materials"] is not None else np.zeros(len(geom["poly_start"]) - 1, np.int64)
        tri_sub, sub_names = _submesh_order(mats[face_of], _material_names(scene, mid))
        idx = remap.reshape(-1, 3)
        indices, submeshes, cursor = [], [], 0
        for s, name in enumerate(sub_names):
            bucket = idx[tri_sub == s].reshape(-1)
            if len(bucket) == 0:
                continue
            submeshes.append({"start": cursor, "count": int(len(bucket)), "material": name})
            indices.append(bucket)
            cursor += len(bucket)
        world = G @ scene.global_matrix(mid) @ scene.geometric_matrix(mid)
        meshes.append({
            "name": scene.name(mid),
            "transform": world.astype(np.float32).reshape(-1),  # row-major, as _matrix_to_row_major
            "mesh": {
                "positions": p[firsts].astype(np.float32).reshape(-1),
                "normals": n[firsts].astype(np.float32).reshape(-1),
                "uvs": uv[firsts].astype(np.float32).reshape(-1),
                "indices": np.concatenate(indices).astype(np.uint32) if indices else np.zeros(0, np.uint32),
                "submeshes": submeshes,
            },
            "collisionHulls": build_collision_hulls(_f32(geom["points"]), [geom["loops"][geom["poly_start"][k]:geom["poly_start"][k + 1]]
                                                                              for k in range(len(geom["poly_start"]) - 1)]) if hulls else [],
        })
    return {"version": 1, "meshes": meshes}


def _cluster_table(scene, geom_id):
    """[(bone name, control-point indexes, weights, TransformLink 4x4)] in connection order for one geometry."""
    out = []
    for skin in scene.children(geom_id, "Deformer", "Skin"):
        for cl in scene.children(skin, "Deformer", "Cluster"):
            node = scene.objects[cl]
            bones = scene.children(cl, "Model")
            if not bones:
                continue
            idx = node.value("Indexes")
            wts = node.value("Weights")
            link = np.asarray(node.value("TransformLink"), np.float64).reshape(4, 4).T  # stored column-major
            out.append((scene.name(bones[0]),
                        np.asarray(idx if idx is not None else [], np.int64),
                        np.asarray(wts if wts is not None else [], np.float64), link))
    return out


def _limb_order(scene):
    """Armature bone order = depth-first limb-node hierarchy in connection order (Blender creates edit bones
    while walking the FBX node tree); for the Y-Bot this is the order of YBot.skeleton.json."""
    limbs = set(scene.of_kind("Model", "LimbNode"))
    roots = [m for m in limbs if not [p for p in scene.parents(m, "Model") if p in limbs]]
    roots.sort(key=lambda m: list(scene.objects).index(m))
    order = []

    def walk(m):
        order.append(m)
        for c in scene.children(m, "Model"):
            if c in limbs:
                walk(c)

    for r in roots:
        walk(r)
    return order


def export_skinned_mesh(scene):
    """-> payload dict in the *.skinned.json schema (SkinnedMeshLoader.swift:190-220), arrays as numpy."""
    limb_ids = _limb_order(scene)
    if not limb_ids:
        raise ValueError("No armature found in FBX.")
    bone_names = [scene.name(m) for m in limb_ids]
    name_to_index = {}
    for i, name in enumerate(bone_names):
        name_to_index[name.lower()] = i
        if ":" in name:
            name_to_index[name.split(":")[-1].lower()] = i

    models = [m for m in scene.of_kind("Model", "Mesh")
              if any(scene.children(g, "Deformer", "Skin") for g in scene.children(m, "Geometry"))]
    models.sort(key=lambda m: scene.name(m))  # bpy scene objects enumerate in name order
    if not models:
        raise ValueError("No mesh found in FBX.")

    # armature space: Blender parents the rig to an armature object carrying the global correction, and every
    # mesh object gets the same correction, so arm_inv @ mesh.matrix_world reduces to the FBX-space mesh global
    bind_global = {}
    P, N, UV, J, W, KEYS, SUB = [], [], [], [], [], [], []
    sub_names = []
    for mid in models:
        gid = scene.children(mid, "Geometry")[0]
        geom = fbx.geometry_arrays(scene.objects[gid])
        clusters = _cluster_table(scene, gid)
        npts = len(geom["points"])
        # vertex groups: per control point, (bone index, weight) in cluster order
        groups_bone = [[] for _ in range(npts)]
        groups_w = [[] for _ in range(npts)]
        for name, idx, wts, link in clusters:
            key = name.lower()
            b = name_to_index.get(key)
            if b is None and ":" in key:
                b = name_to_index.get(key.split(":")[-1])
            bind_global.setdefault(name, link)
            if b is None:
                continue
            w32 = _f32(wts)
            for i, w in zip(idx.tolist(), w32.tolist()):
                groups_bone[i].append(b)
                groups_w[i].append(w)
        jidx = np.zeros((npts, 4), np.int64)
        jw = np.zeros((npts, 4), np.float64)
        for i in range(npts):
            if not groups_bone[i]:
                jw[i, 0] = 1.0  # _vertex_weights: no groups -> joint 0, weight 1
                continue
            order = sorted(range(len(groups_w[i])), key=lambda k: groups_w[i][k], reverse=True)[:4]  # stable
            ws = [groups_w[i][k] for k in order]
            total = sum(ws)
            if total > 0:
                ws = [w / total for w in ws]
            for k, o in enumerate(order):
                jidx[i, k] = groups_bone[i][o]
                jw[i, k] = ws[k]

        mesh_to_arm = scene.global_matrix(mid) @ scene.geometric_matrix(mid)
        normal_mat = mesh_to_arm[:3, :3]
        tris, face_of = _loop_triangles(geom)
        loops = tris.reshape(-1)
        cp = geom["loops"][loops]
        p = _f32(geom["points"])[cp] @ mesh_to_arm[:3, :3].T + mesh_to_arm[:3, 3]
        n = geom["normals"]
        n = _f32(_unit(n))[loops] if n is not None else _face_normals_for_loops(geom)[loops]
        n = n @ normal_mat.T
        uv = _f32(geom["uvs"])[loops] if geom["uvs"] is not None else np.zeros((len(loops), 2))
        j = jidx[cp]
        w = jw[cp]
        mats = geom["materials"] if geom["materials"] is not None else np.zeros(len(geom["poly_start"]) - 1, np.int64)
        names = _material_names(scene, mid)
        for m in mats[face_of]:
            name = names[m] if m < len(names) else "Default"
            if name not in sub_names:
                sub_names.append(name)
            SUB.append(sub_names.index(name))
        P.append(p), N.append(n), UV.append(uv), J.append(j), W.append(w)
        KEYS.append(np.concatenate([_q(p), _q(n), _q(uv), j, _q(w)], axis=1))

    p, n, uv, j, w = (np.concatenate(a) for a in (P, N, UV, J, W))
    remap, firsts = _weld(np.concatenate(KEYS))
    idx = remap.reshape(-1, 3)
    tri_sub = np.asarray(SUB, np.int64)
    indices, submeshes, cursor = [], [], 0
    for s, name in enumerate(sub_names):
        bucket = idx[tri_sub == s].reshape(-1)
        if len(bucket) == 0:
            continue
        submeshes.append({"start": cursor, "count": int(len(bucket)), "material": name})
        indices.append(bucket)
        cursor += len(bucket)

    bones = []
    for name, mid in zip(bone_names, limb_ids):
        link = bind_global.get(name)
        if link is None:
            link = scene.global_matrix(mid)  # bones without a cluster: rest pose from the node tree
        bones.append({"name": name, "inverseBindMatrix": np.linalg.inv(link).astype(np.float32).reshape(-1)})  # row-major
    return {
        "version": 1,
        "mesh": {
            "positions": p[firsts].astype(np.float32).reshape(-1),
            "normals": n[firsts].astype(np.float32).reshape(-1),
            "uvs": uv[firsts].astype(np.float32).reshape(-1),
            "joints": j[firsts].astype(np.uint16).reshape(-1),
            "weights": w[firsts].astype(np.float32).reshape(-1),
            "indices": np.concatenate(indices).astype(np.uint32),
            "submeshes": submeshes,
        },
        "skin": {"bones": bones},
    }


def to_jsonable(payload):
    """numpy arrays -> lists, so json.dump writes the exporters' file format."""
    if isinstance(payload, dict):
        return {k: to_jsonable(v) for k, v in payload.items()}
    if isinstance(payload, (list, tuple)):
        return [to_jsonable(v) for v in payload]
    if isinstance(payload, np.ndarray):
        return payload.tolist()
    if isinstance(payload, (np.floating, np.integer)):
        return payload.item()
    return payload
