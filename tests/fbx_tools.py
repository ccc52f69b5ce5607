"""Binary FBX for the tests: an independent READER (pure Python + zlib, no code shared with dxrexperiments_amd/csrc/rt_fbx.cpp)
that ingests a file under the rules DESIGN.md section 2 defines, and a WRITER that synthesises small files with known content
(32- and 64-bit record headers, raw and zlib arrays, the normal mappings, a Model transform), so the product reader is
exercised on the GPU box too, where the reference's assets/models/ground.fbx does not exist.

Format (public): 27-byte header "Kaydara FBX Binary  \0\x1a\0" + uint32 version; node record = end offset, property count,
property-list bytes (uint32 each; uint64 from version 7500), uint8 name length, name, properties, nested records, and a null
record (13 / 25 zero bytes) after the children.  Property = type char + payload: Y int16, C bool, I int32, F float, D double,
L int64, S / R uint32 length + bytes, f d l i b = uint32 count, uint32 encoding (1 = zlib), uint32 byte length, data."""
import struct
import zlib

import numpy as np


# ---- independent reader ---------------------------------------------------------------------------------------

def _props(d, p, n):
    out = []
    for _ in range(n):
        t = chr(d[p]); p += 1
        if t in "YCIFDL":
            fmt = {"Y": "<h", "C": "<B", "I": "<i", "F": "<f", "D": "<d", "L": "<q"}[t]
            out.append(struct.unpack_from(fmt, d, p)[0]); p += struct.calcsize(fmt)
        elif t in "fdlib":
            cnt, enc, clen = struct.unpack_from("<III", d, p); p += 12
            raw = d[p:p + clen]; p += clen
            if enc == 1:
                raw = zlib.decompress(raw)
            out.append(np.frombuffer(raw, {"f": "<f4", "d": "<f8", "l": "<i8", "i": "<i4", "b": "u1"}[t], cnt))
        elif t in "SR":
            ln = struct.unpack_from("<I", d, p)[0]; p += 4
            out.append(bytes(d[p:p + ln])); p += ln
        else:
            raise ValueError("property type %r" % t)
    return out


def _node(d, off, wide):
    if wide:
        end, nprops, plen = struct.unpack_from("<QQQ", d, off); off += 24
    else:
        end, nprops, plen = struct.unpack_from("<III", d, off); off += 12
    nlen = d[off]; off += 1
    if end == 0:
        return None, off
    name = bytes(d[off:off + nlen]).decode(); off += nlen
    props = _props(d, off, nprops)
    off += plen
    kids = []
    while off < end:
        k, nxt = _node(d, off, wide)
        if k is None:
            break
        kids.append(k)
        off = k["end"]
    return {"name": name, "props": props, "kids": kids, "end": end}, end


def parse(path):
    d = open(path, "rb").read()
    assert d[:20] == b"Kaydara FBX Binary  ", "not a binary FBX"
    version = struct.unpack_from("<I", d, 23)[0]
    off, top = 27, []
    while off + 13 <= len(d):
        n, _ = _node(d, off, version >= 7500)
        if n is None:
            break
        top.append(n)
        off = n["end"]
    return version, top


def _kid(n, name):
    return next((k for k in n["kids"] if k["name"] == name), None)


def ingest(path):
    """(pos+normal float32[n, 6], uint32[m, 3]) under the ordering rules of DESIGN.md section 2 / rt_fbx.cpp's header:
    Geometry nodes in file order, polygons fanned from their first corner, one vertex per distinct (position index, normal
    value) in first-use order, the owning Model's Lcl transform applied.  Only what the fixtures need: identity or T / R / S."""
    _, top = parse(path)
    objects = next(n for n in top if n["name"] == "Objects")
    conns = next((n for n in top if n["name"] == "Connections"), None)
    models = {m["props"][0]: m for m in objects["kids"] if m["name"] == "Model"}
    parent = {c["props"][1]: c["props"][2] for c in (conns["kids"] if conns else []) if c["name"] == "C" and c["props"][0] == b"OO"}
    verts, idx = [], []
    for g in objects["kids"]:
        if g["name"] != "Geometry" or g["props"][2] != b"Mesh":
            continue
        M = np.eye(4)
        m = models.get(parent.get(g["props"][0]))
        if m is not None and _kid(m, "Properties70"):
            t, r, s = np.zeros(3), np.zeros(3), np.ones(3)
            for p in _kid(m, "Properties70")["kids"]:
                dst = {b"Lcl Translation": t, b"Lcl Rotation": r, b"Lcl Scaling": s}.get(p["props"][0])
                if dst is not None:
                    dst[:] = p["props"][4:7]
            cx, cy, cz = np.cos(np.radians(r)); sx, sy, sz = np.sin(np.radians(r))
            Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
            Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
            M[:3, :3] = (Rz @ Ry @ Rx) * s[None, :]
            M[:3, 3] = t
        P = np.asarray(_kid(g, "Vertices")["props"][0], np.float64).reshape(-1, 3)
        pv = np.asarray(_kid(g, "PolygonVertexIndex")["props"][0], np.int64)
        plain = np.array_equal(M, np.eye(4))
        Pw = P if plain else P @ M[:3, :3].T + M[:3, 3]
        ln = _kid(g, "LayerElementNormal")
        N = np.asarray(_kid(ln, "Normals")["props"][0], np.float64).reshape(-1, 3)
        by_pv = _kid(ln, "MappingInformationType")["props"][0] == b"ByPolygonVertex"
        ni = _kid(ln, "NormalsIndex")
        indexed = _kid(ln, "ReferenceInformationType")["props"][0] == b"IndexToDirect" and ni is not None
        if not plain:
            it = np.linalg.inv(M[:3, :3]).T
        corners, poly = [], []
        for k, v in enumerate(pv):
            last = v < 0
            poly.append((int(~v if last else v), k))
            if last:
                for j in range(1, len(poly) - 1):
                    corners += [poly[0], poly[j], poly[j + 1]]
                poly = []
        base, seen = len(verts), {}
        for p, k in corners:
            e = k if by_pv else p
            if indexed:
                e = int(ni["props"][0][e])
            n = N[e]
            if not plain:
                n = it @ n
                n = n / np.linalg.norm(n)
            n32 = n.astype(np.float32)
            key = (p, n32.tobytes())
            if key not in seen:
                seen[key] = len(verts)
                verts.append(np.concatenate([Pw[p].astype(np.float32), n32]))
            idx.append(seen[key])
        assert base <= len(verts)
    return np.array(verts, np.float32).reshape(-1, 6), np.array(idx, np.uint32).reshape(-1, 3)


# ---- writer -----------------------------------------------------------------------------------------------------

def _prop(v, compress):
    if isinstance(v, bytes):
        return b"S" + struct.pack("<I", len(v)) + v
    if isinstance(v, float):
        return b"D" + struct.pack("<d", v)
    if isinstance(v, (int, np.integer)):
        return (b"L" + struct.pack("<q", int(v))) if abs(int(v)) > 2 ** 31 - 1 else (b"I" + struct.pack("<i", int(v)))
    a = np.ascontiguousarray(v)
    t = {"float64": b"d", "float32": b"f", "int32": b"i", "int64": b"l"}[a.dtype.name]
    raw = a.tobytes()
    if compress:
        z = zlib.compress(raw)
        return t + struct.pack("<III", a.size, 1, len(z)) + z
    return t + struct.pack("<III", a.size, 0, len(raw)) + raw


def _record(name, props, kids, at, wide, compress):
    """bytes of one node record that starts at absolute offset `at`"""
    pb = b"".join(_prop(p, compress) for p in props)
    head = (24 if wide else 12) + 1 + len(name)
    body_at = at + head + len(pb)
    kb = b""
    for k in kids:
        kb += _record(k[0], k[1], k[2], body_at + len(kb), wide, compress)
    if kids:
        kb += b"\0" * (25 if wide else 13)
    end = body_at + len(kb)
    hdr = struct.pack("<QQQ" if wide else "<III", end, len(props), len(pb))
    return hdr + bytes([len(name)]) + name.encode() + pb + kb


def write(path, meshes, version=7500, compress=True):
    """meshes: list of dict(positions float[n,3], polygons list of index lists, normals float[k,3], mapping "ByPolygonVertex" |
    "ByVertice", normals_index int list or None, translation / rotation / scaling 3-tuples or None)."""
    wide = version >= 7500
    objs, conns = [], []
    for i, m in enumerate(meshes):
        gid, mid = 1000 + i, 2000 + i
        pvi = []
        for poly in m["polygons"]:
            pvi += list(poly[:-1]) + [~int(poly[-1])]
        ln = [("Version", [101], []), ("Name", [b""], []), ("MappingInformationType", [m.get("mapping", "ByPolygonVertex").encode()], []),
              ("ReferenceInformationType", [b"IndexToDirect" if m.get("normals_index") is not None else b"Direct"], []),
              ("Normals", [np.asarray(m["normals"], np.float64).reshape(-1)], [])]
        if m.get("normals_index") is not None:
            ln.append(("NormalsIndex", [np.asarray(m["normals_index"], np.int32)], []))
        objs.append(("Geometry", [gid, b"\x00\x01Geometry", b"Mesh"],
                     [("Vertices", [np.asarray(m["positions"], np.float64).reshape(-1)], []),
                      ("PolygonVertexIndex", [np.asarray(pvi, np.int32)], []),
                      ("LayerElementNormal", [0], ln)]))
        p70 = []
        for key, val in (("Lcl Translation", m.get("translation")), ("Lcl Rotation", m.get("rotation")), ("Lcl Scaling", m.get("scaling"))):
            if val is not None:
                p70.append(("P", [key.encode(), key.encode(), b"", b"A"] + [float(x) for x in val], []))
        objs.append(("Model", [mid, b"Mesh%d\x00\x01Model" % i, b"Mesh"], [("Version", [232], []), ("Properties70", [], p70)]))
        conns += [("C", [b"OO", mid, 0], []), ("C", [b"OO", gid, mid], [])]
    top = [("FBXHeaderExtension", [], [("FBXVersion", [version], [])]), ("Objects", [], objs), ("Connections", [], conns)]
    out = b"Kaydara FBX Binary  \x00\x1a\x00" + struct.pack("<I", version)
    for t in top:
        out += _record(t[0], t[1], t[2], len(out), wide, compress)
    out += b"\0" * (25 if wide else 13)
    open(path, "wb").write(out)
