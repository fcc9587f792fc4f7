"""Minimal binary-FBX reader (stdlib + numpy): enough of the node tree to pull geometry, skin clusters,
limb nodes and connections out of the engine's source assets.

It stands in for the Blender import the reference's exporters start from
(Tools/FbxToSkinnedJson/export_skinned_json.py:118 and Tools/FbxToStaticMeshJson/export_static_mesh_json.py:
`bpy.ops.import_scene.fbx`); `bpy` does not exist outside Blender.  The record layout is the public
binary FBX 7.x layout: 27-byte header, then nested node records
{endOffset, numProperties, propertyListLen, nameLen, name, properties..., children..., null record}
with 32-bit offsets below version 7500 and 64-bit from 7500 on; array properties may be zlib-deflated.
"""
import struct
import zlib

import numpy as np

_MAGIC = b"Kaydara FBX Binary  \x00\x1a\x00"
_ARRAY_DTYPES = {b"f": "<f4", b"d": "<f8", b"l": "<i8", b"i": "<i4", b"b": "u1"}
_SCALARS = {b"Y": "<h", b"C": "<?", b"I": "<i", b"F": "<f", b"D": "<d", b"L": "<q"}


class FbxNode:
    __slots__ = ("name", "props", "children")

    def __init__(self, name, props, children):
        self.name = name
        self.props = props
        self.children = children

    def find(self, name):
        for c in self.children:
            if c.name == name:
                return c
        return None

    def find_all(self, name):
        return [c for c in self.children if c.name == name]

    def value(self, name, default=None):
        c = self.find(name)
        if c is None or not c.props:
            return default
        return c.props[0]

    def __repr__(self):
        return "FbxNode(%s, %d props, %d children)" % (self.name, len(self.props), len(self.children))


def _read_props(buf, pos, count):
    props = []
    for _ in range(count):
        t = buf[pos:pos + 1]
        pos += 1
        if t in _SCALARS:
            fmt = _SCALARS[t]
            size = struct.calcsize(fmt)
            props.append(struct.unpack_from(fmt, buf, pos)[0])
            pos += size
        elif t in _ARRAY_DTYPES:
            length, encoding, clen = struct.unpack_from("<III", buf, pos)
            pos += 12
            raw = buf[pos:pos + clen]
            pos += clen
            if encoding == 1:
                raw = zlib.decompress(raw)
            arr = np.frombuffer(raw, dtype=_ARRAY_DTYPES[t], count=length)
            props.append(arr)
        elif t in (b"S", b"R"):
            (n,) = struct.unpack_from("<I", buf, pos)
            pos += 4
            data = bytes(buf[pos:pos + n])
            pos += n
            props.append(data.decode("utf-8", "replace") if t == b"S" else data)
        else:
            raise ValueError("fbx: unknown property type %r at %d" % (t, pos - 1))
    return props, pos


def _read_node(buf, pos, wide):
    if wide:
        end, nprops, plen = struct.unpack_from("<QQQ", buf, pos)
        pos += 24
    else:
        end, nprops, plen = struct.unpack_from("<III", buf, pos)
        pos += 12
    nlen = buf[pos]
    pos += 1
    if end == 0:
        return None, pos  # null record
    name = bytes(buf[pos:pos + nlen]).decode("ascii", "replace")
    pos += nlen
    props, p2 = _read_props(buf, pos, nprops)
    if p2 != pos + plen:
        raise ValueError("fbx: property list length mismatch in node %s" % name)
    pos = p2
    children = []
    while pos < end:
        child, pos = _read_node(buf, pos, wide)
        if child is None:
            break
        children.append(child)
    return FbxNode(name, props, children), end


def parse(path):
    """Returns (version, root FbxNode whose children are the top-level records)."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    if bytes(buf[:len(_MAGIC)]) != _MAGIC:
        raise ValueError("fbx: %s is not a binary FBX file" % path)
    (version,) = struct.unpack_from("<I", buf, 23)
    wide = version >= 7500
    pos = 27
    top = []
    while pos < len(buf):
        node, pos = _read_node(buf, pos, wide)
        if node is None:
            break
        top.append(node)
    return version, FbxNode("", [], top)


def _clean(name):
    """'Model::mixamorig:Hips' is stored as 'mixamorig:Hips\\x00\\x01Model'."""
    return name.split("\x00\x01")[0]


class FbxScene:
    """Objects by id + the OO connection graph, and the handful of typed views the exporters need."""

    def __init__(self, path):
        self.version, self.root = parse(path)
        self.objects = {}
        objs = self.root.find("Objects")
        for o in (objs.children if objs else []):
            if o.props and isinstance(o.props[0], int):
                self.objects[o.props[0]] = o
        self.children_of = {}
        self.parents_of = {}
        conns = self.root.find("Connections")
        for c in (conns.children if conns else []):
            if c.name != "C" or c.props[0] not in ("OO", "OP"):
                continue
            child, parent = c.props[1], c.props[2]
            self.children_of.setdefault(parent, []).append(child)
            self.parents_of.setdefault(child, []).append(parent)
        gs = self.root.find("GlobalSettings")
        self.settings = self._properties(gs) if gs else {}

    @staticmethod
    def _properties(node):
        out = {}
        p70 = node.find("Properties70")
        for p in (p70.children if p70 else []):
            if p.name == "P" and p.props:
                out[p.props[0]] = p.props[4:]
        return out

    def properties(self, obj_id):
        return self._properties(self.objects[obj_id])

    def kind(self, obj_id):
        o = self.objects.get(obj_id)
        return (o.name, o.props[2] if len(o.props) > 2 else "") if o else (None, None)

    def name(self, obj_id):
        return _clean(self.objects[obj_id].props[1])

    def of_kind(self, node_name, subtype=None):
        return [i for i, o in self.objects.items()
                if o.name == node_name and (subtype is None or (len(o.props) > 2 and o.props[2] == subtype))]

    def children(self, obj_id, node_name=None, subtype=None):
        out = []
        for c in self.children_of.get(obj_id, []):
            n, s = self.kind(c)
            if n is None:
                continue
            if (node_name is None or n == node_name) and (subtype is None or s == subtype):
                out.append(c)
        return out

    def parents(self, obj_id, node_name=None):
        return [p for p in self.parents_of.get(obj_id, []) if node_name is None or self.kind(p)[0] == node_name]

    # ---- transforms -------------------------------------------------------------------------------
    def local_trs(self, model_id):
        """Lcl Translation / Rotation / Scaling, PreRotation, and the pivots Mixamo/Max files never use."""
        p = self.properties(model_id)

        def v(key, default):
            return np.asarray(p.get(key, default), np.float64)[:3]

        return {
            "T": v("Lcl Translation", (0, 0, 0)), "R": v("Lcl Rotation", (0, 0, 0)), "S": v("Lcl Scaling", (1, 1, 1)),
            "PreRotation": v("PreRotation", (0, 0, 0)), "PostRotation": v("PostRotation", (0, 0, 0)),
            "RotationOffset": v("RotationOffset", (0, 0, 0)), "RotationPivot": v("RotationPivot", (0, 0, 0)),
            "ScalingOffset": v("ScalingOffset", (0, 0, 0)), "ScalingPivot": v("ScalingPivot", (0, 0, 0)),
            "GeometricTranslation": v("GeometricTranslation", (0, 0, 0)),
            "GeometricRotation": v("GeometricRotation", (0, 0, 0)),
            "GeometricScaling": v("GeometricScaling", (1, 1, 1)),
        }

    def local_matrix(self, model_id):
        """FBX node local transform  T * Roff * Rp * Rpre * R * Rpost^-1 * Rp^-1 * Soff * Sp * S * Sp^-1 (XYZ euler)."""
        t = self.local_trs(model_id)
        M = translation(t["T"]) @ translation(t["RotationOffset"]) @ translation(t["RotationPivot"])
        M = M @ euler_xyz(t["PreRotation"]) @ euler_xyz(t["R"]) @ np.linalg.inv(euler_xyz(t["PostRotation"]))
        M = M @ translation(-t["RotationPivot"]) @ translation(t["ScalingOffset"]) @ translation(t["ScalingPivot"])
        M = M @ scaling(t["S"]) @ translation(-t["ScalingPivot"])
        return M

    def geometric_matrix(self, model_id):
        t = self.local_trs(model_id)
        return translation(t["GeometricTranslation"]) @ euler_xyz(t["GeometricRotation"]) @ scaling(t["GeometricScaling"])

    def global_matrix(self, model_id):
        M = self.local_matrix(model_id)
        cur = model_id
        while True:
            ps = self.parents(cur, "Model")
            if not ps:
                return M
            cur = ps[0]
            M = self.local_matrix(cur) @ M


def translation(t):
    M = np.eye(4)
    M[:3, 3] = t
    return M


def scaling(s):
    return np.diag([s[0], s[1], s[2], 1.0])


def euler_xyz(deg):
    """FBX eEulerXYZ: R = Rz * Ry * Rx (the same composition as Skeleton.rotationXYZDegrees, Skeleton.swift:212-217)."""
    x, y, z = np.radians(np.asarray(deg, np.float64))
    cx, sx, cy, sy, cz, sz = np.cos(x), np.sin(x), np.cos(y), np.sin(y), np.cos(z), np.sin(z)
    Rx = np.array([[1, 0, 0, 0], [0, cx, -sx, 0], [0, sx, cx, 0], [0, 0, 0, 1.0]])
    Ry = np.array([[cy, 0, sy, 0], [0, 1, 0, 0], [-sy, 0, cy, 0], [0, 0, 0, 1.0]])
    Rz = np.array([[cz, -sz, 0, 0], [sz, cz, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    return Rz @ Ry @ Rx


# ---- geometry -----------------------------------------------------------------------------------

def _layer_values(geom, layer_name, data_name, index_name, width, polygon_vertex_index):
    """One LayerElement resolved to a per-polygon-vertex (loop) array, or None."""
    le = geom.find(layer_name)
    if le is None:
        return None
    data = le.value(data_name)
    if data is None or len(data) == 0:
        return None
    data = np.asarray(data, np.float64).reshape(-1, width)
    mapping = le.value("MappingInformationType", "ByPolygonVertex")
    ref = le.value("ReferenceInformationType", "Direct")
    nloops = len(polygon_vertex_index)
    if ref == "Direct":
        index = None
    else:
        index = np.asarray(le.value(index_name), np.int64)
    if mapping == "ByPolygonVertex":
        sel = index if index is not None else np.arange(nloops)
    elif mapping in ("ByVertice", "ByVertex"):
        cp = np.where(polygon_vertex_index < 0, -polygon_vertex_index - 1, polygon_vertex_index)
        sel = index[cp] if index is not None else cp
    elif mapping == "AllSame":
        sel = np.zeros(nloops, np.int64) if index is None else np.full(nloops, index[0])
    else:
        return None
    return data[sel]


def geometry_arrays(geom):
    """Geometry node -> dict(points [P,3], loops [L] control-point index, poly_start [F+1], normals [L,3]|None,
    uvs [L,2]|None, materials [F]|None)."""
    points = np.asarray(geom.value("Vertices"), np.float64).reshape(-1, 3)
    pvi = np.asarray(geom.value("PolygonVertexIndex"), np.int64)
    ends = np.nonzero(pvi < 0)[0]
    poly_start = np.concatenate([[0], ends + 1])
    loops = np.where(pvi < 0, -pvi - 1, pvi)
    normals = _layer_values(geom, "LayerElementNormal", "Normals", "NormalsIndex", 3, pvi)
    uvs = _layer_values(geom, "LayerElementUV", "UV", "UVIndex", 2, pvi)
    materials = None
    lm = geom.find("LayerElementMaterial")
    if lm is not None and lm.value("Materials") is not None:
        m = np.asarray(lm.value("Materials"), np.int64)
        nf = len(poly_start) - 1
        if lm.value("MappingInformationType", "AllSame") == "ByPolygon" and len(m) == nf:
            materials = m
        else:
            materials = np.full(nf, m[0] if len(m) else 0)
    return {"points": points, "loops": loops, "poly_start": poly_start, "normals": normals, "uvs": uvs, "materials": materials}


def triangulate_fan(poly_start):
    """Loop-index triples of a plain triangle fan per polygon."""
    tris = []
    face_of = []
    for f in range(len(poly_start) - 1):
        s, e = int(poly_start[f]), int(poly_start[f + 1])
        for k in range(1, e - s - 1):
            tris.append((s, s + k, s + k + 1))
            face_of.append(f)
    return np.asarray(tris, np.int64).reshape(-1, 3), np.asarray(face_of, np.int64)


def _quad_flip(v0, v1, v2, v3):
    """Blender's quad rule: split along 0-2 unless corners 1 and 3 fall on the same side of that diagonal."""
    d12, d13, d14 = v1 - v0, v2 - v0, v3 - v0
    return float(np.dot(np.cross(d12, d13), np.cross(d14, d13))) > 0.0


def _ear_clip(poly2d):
    """Ear clipping in the visiting order of Blender's polyfill: cut an ear, restart two corners further on,
    reverse the sweep when that corner is reflex; the last triangle starts at the surviving list head."""
    n = len(poly2d)
    area = 0.0
    for i in range(n):
        a, b = poly2d[i], poly2d[(i + 1) % n]
        area += a[0] * b[1] - b[0] * a[1]
    orient = 1.0 if area >= 0 else -1.0

    def sign(a, b, c):
        v = orient * ((b[0] - a[0]) * (c[1] - b[1]) - (b[1] - a[1]) * (c[0] - b[0]))
        return 1 if v > 0 else (-1 if v < 0 else 0)

    nxt = list(range(1, n)) + [0]
    prv = [n - 1] + list(range(0, n - 1))
    alive = n
    head = 0
    out = []

    def corner_sign(i):
        return sign(poly2d[prv[i]], poly2d[i], poly2d[nxt[i]])

    def is_ear(i):
        if corner_sign(i) != 1:
            return False
        a, b, c = poly2d[prv[i]], poly2d[i], poly2d[nxt[i]]
        k = nxt[nxt[i]]
        while k != prv[i]:
            if corner_sign(k) != 1:
                p = poly2d[k]
                if sign(a, b, p) >= 0 and sign(b, c, p) >= 0 and sign(c, a, p) >= 0:
                    return False
            k = nxt[k]
        return True

    init, reverse = head, False
    while alive > 3:
        ear = None
        k = init
        for _ in range(alive):
            if is_ear(k):
                ear = k
                break
            k = prv[k] if reverse else nxt[k]
        if ear is None:  # degenerate polygon: fall back to the first convex (or any) corner
            k = init
            for _ in range(alive):
                if corner_sign(k) >= 0:
                    ear = k
                    break
                k = nxt[k]
            if ear is None:
                ear = init
        p, q = prv[ear], nxt[ear]
        out.append((p, ear, q))
        nxt[p], prv[q] = q, p
        if ear == head:
            head = q
        alive -= 1
        init = prv[p] if reverse else nxt[q]
        if corner_sign(init) != 1:
            init = prv[init] if reverse else nxt[init]
            reverse = not reverse
    out.append((head, nxt[head], nxt[nxt[head]]))
    return out


def triangulate_blender(points, loops, poly_start):
    """Loop-index triples per polygon as Blender's `calc_loop_triangles` produces them for the shapes the
    engine's assets hold: triangles as they are, quads split 0-2 (or 1-3 when 0-2 leaves the quad), n-gons
    ear-clipped on the polygon's plane."""
    pts = np.asarray(points, np.float32).astype(np.float64)
    tris, face_of = [], []
    for f in range(len(poly_start) - 1):
        s, e = int(poly_start[f]), int(poly_start[f + 1])
        n = e - s
        if n < 3:
            continue
        if n == 3:
            tris.append((s, s + 1, s + 2))
            face_of.append(f)
        elif n == 4:
            v = pts[loops[s:e]]
            if _quad_flip(v[0], v[1], v[2], v[3]):
                tris += [(s, s + 1, s + 3), (s + 1, s + 2, s + 3)]
            else:
                tris += [(s, s + 1, s + 2), (s, s + 2, s + 3)]
            face_of += [f, f]
        else:
            v = pts[loops[s:e]]
            normal = np.zeros(3)
            for i in range(n):  # Newell
                a, b = v[i], v[(i + 1) % n]
                normal += np.cross(a, b)
            ln = np.linalg.norm(normal)
            normal = normal / ln if ln > 0 else np.array([0.0, 0.0, 1.0])
            ref = np.eye(3)[int(np.argmin(np.abs(normal)))]
            u = np.cross(normal, ref)
            u /= np.linalg.norm(u)
            w = np.cross(normal, u)
            poly2d = np.stack([v @ u, v @ w], axis=1)
            for a, b, c in _ear_clip(poly2d):
                tris.append((s + a, s + b, s + c))
                face_of.append(f)
    return np.asarray(tris, np.int64).reshape(-1, 3), np.asarray(face_of, np.int64)
