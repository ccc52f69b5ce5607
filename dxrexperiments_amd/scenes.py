"""Synthetic scenes for the configurations BASELINE.json names.

The reference hard-codes one scene (src/DXRExperimentsApp.cpp:86-92, an FBX
that is not in the checkout) and ships no Sponza; there is no network, so the
"Sponza-class" atrium, the instanced multi-mesh scene and the 10 M-triangle
stress mesh are generated here from fixed seeds.  Everything
returns the reference's geometry convention (libs/DXRFramework/RtModel.cpp:
13-17,33-53): interleaved {position, normal} vertices + uint32 triangle list.

Winding: a triangle is front-facing when cross(v1-v0, v2-v0) points at the
viewer (the DXR clockwise/left-handed rule restated algebraically), so closed
shapes are wound with that vector pointing outwards and room surfaces with it
pointing into the room.
"""
import numpy as np

from .rtypes import VERTEX


# ---------------------------------------------------------------------------
# small mesh toolkit
# ---------------------------------------------------------------------------

def _hash01(ix, seed):
    """Deterministic integer hash -> float64 in [0,1).  ix: integer array."""
    x = (np.asarray(ix, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(30)
    x = (x * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(27)
    x = (x * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def _grid_indices(nu, nv, flip=False):
    """Triangles of an (nu+1) x (nv+1) vertex grid, vertex id = j*(nu+1)+i."""
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="xy")
    a = (j * (nu + 1) + i).ravel()
    b = a + 1
    c = a + (nu + 1)
    d = c + 1
    t = np.stack([np.stack([a, b, d], 1), np.stack([a, d, c], 1)], 1).reshape(-1, 3)
    if flip:
        t = t[:, ::-1]
    return t.astype(np.uint32)


def _smooth_normals(pos, tri):
    n = np.zeros_like(pos)
    fn = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]])
    for k in range(3):
        np.add.at(n, tri[:, k], fn)
    l = np.linalg.norm(n, axis=1, keepdims=True)
    l[l == 0] = 1.0
    return n / l


class MeshBuilder:
    def __init__(self):
        self.pos, self.nrm, self.tri, self.nv = [], [], [], 0

    def add(self, pos, tri, normals=None):
        pos = np.asarray(pos, np.float64).reshape(-1, 3)
        tri = np.asarray(tri, np.uint32).reshape(-1, 3)
        if normals is None:
            normals = _smooth_normals(pos, tri)
        self.pos.append(pos)
        self.nrm.append(normals)
        self.tri.append(tri + np.uint32(self.nv))
        self.nv += pos.shape[0]

    def surface(self, fn, nu, nv, flip=False):
        """fn(u, v) -> (x, y, z) arrays for u, v in [0,1] on an (nu+1)x(nv+1) grid."""
        u, v = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="xy")
        x, y, z = fn(u.ravel(), v.ravel())
        self.add(np.stack([x, y, z], 1), _grid_indices(nu, nv, flip))

    def box(self, lo, hi):
        lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
        c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                      [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]])
        quads = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)]
        for q in quads:
            p = c[list(q)]
            n = np.cross(p[1] - p[0], p[2] - p[0])
            n = n / np.linalg.norm(n)
            self.add(p, [[0, 1, 2], [0, 2, 3]], np.tile(n, (4, 1)))

    def finish(self):
        pos = np.concatenate(self.pos).astype(np.float32)
        nrm = np.concatenate(self.nrm).astype(np.float32)
        tri = np.concatenate(self.tri).astype(np.uint32)
        v = np.zeros(pos.shape[0], VERTEX)
        v["position"] = pos
        v["normal"] = nrm
        return v, tri


def icosphere(level):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    p = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(level):
        edge = {}
        pts = list(p)

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in edge:
                m = pts[a] + pts[b]
                pts.append(m / np.linalg.norm(m))
                edge[k] = len(pts) - 1
            return edge[k]
        nf = []
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        p, f = np.array(pts), np.array(nf, np.int64)
    return p, f.astype(np.uint32)


# ---------------------------------------------------------------------------
# C1: Cornell box arrays come from the OBJ fixture (tests/golden/cornell.obj)
# ---------------------------------------------------------------------------

def cornell_camera():
    """eye, at, up, vfov, for the +-1 Cornell box."""
    return dict(eye=(0.0, 0.0, 3.2), at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=float(np.float32(np.pi / 4)))


# ---------------------------------------------------------------------------
# C2: "Sponza-class" atrium, ~262k triangles
# ---------------------------------------------------------------------------

def sponza_class(seed=42, detail=1.0):
    """Two-storey colonnaded atrium, open to the sky.  x in [-16,16] (long axis),
    z in [-7,7], floor y=-4, gallery y=1, parapet y=7.  detail scales tessellation;
    detail=1 gives ~262k triangles."""
    mb = MeshBuilder()
    X0, X1, Z0, Z1, Y0, Y1, Y2 = -16.0, 16.0, -7.0, 7.0, -4.0, 1.0, 7.0
    q = lambda n: max(2, int(round(n * detail)))
    TAU = 2.0 * np.pi

    def bump(u, v, s, amp, fu, fv):
        i = np.floor(u * fu).astype(np.int64)
        j = np.floor(v * fv).astype(np.int64)
        return amp * (_hash01(i * 7919 + j * 104729, seed + s) - 0.5)

    # floor: flagstones, normal +y
    nfu, nfv = q(96), q(44)
    mb.surface(lambda u, v: (X0 + (X1 - X0) * u, Y0 + bump(u, v, 1, 0.03, 32, 14), Z1 - (Z1 - Z0) * v), nfu, nfv)

    # four walls, brick relief, normals into the room
    nwu, nwv = q(160), q(56)
    wall_h = Y2 - Y0

    def wall_long(zc, sign, s):
        def f(u, v):
            d = bump(u, v, s, 0.05, 64, 28)
            return X0 + (X1 - X0) * u, Y0 + wall_h * v, zc + sign * d
        return f
    mb.surface(wall_long(Z0, +1.0, 2), nwu, nwv, flip=False)          # -z wall faces +z
    mb.surface(wall_long(Z1, -1.0, 3), nwu, nwv, flip=True)           # +z wall faces -z
    nsu = q(72)

    def wall_short(xc, sign, s):
        def f(u, v):
            d = bump(u, v, s, 0.05, 28, 28)
            return xc + sign * d, Y0 + wall_h * v, Z0 + (Z1 - Z0) * u
        return f
    mb.surface(wall_short(X0, +1.0, 4), nsu, nwv, flip=True)          # -x wall faces +x
    mb.surface(wall_short(X1, -1.0, 5), nsu, nwv, flip=False)         # +x wall faces -x

    # gallery slabs along both long walls (upper storey floor), 3 m deep
    for zc0, zc1 in ((Z0, Z0 + 3.0), (Z1 - 3.0, Z1)):
        mb.box((X0, Y1 - 0.3, zc0), (X1, Y1, zc1))

    # columns: 2 storeys x 2 rows x 11, fluted shafts with entasis, plus plinth and capital boxes
    ncol = 11
    seg, rings = q(32), q(16)
    col_x = np.linspace(X0 + 2.0, X1 - 2.0, ncol)
    rows = (Z0 + 3.0, Z1 - 3.0)
    for storey, (yb, yt, rad) in enumerate(((Y0, Y1 - 0.3, 0.42), (Y1, Y1 + 4.2, 0.32))):
        for zc in rows:
            for ci, xc in enumerate(col_x):
                ph = TAU * _hash01(np.array([ci + 100 * storey]), seed + 11)[0]

                def shaft(u, v, xc=xc, zc=zc, yb=yb, yt=yt, rad=rad, ph=ph):
                    a = TAU * u
                    r = rad * (1.0 - 0.18 * v * v) * (1.0 + 0.04 * np.cos(12.0 * a + ph))
                    return xc + r * np.cos(a), yb + 0.35 + (yt - yb - 0.7) * v, zc - r * np.sin(a)
                mb.surface(shaft, seg, rings)
                mb.box((xc - rad * 1.4, yb, zc - rad * 1.4), (xc + rad * 1.4, yb + 0.35, zc + rad * 1.4))
                mb.box((xc - rad * 1.5, yt - 0.35, zc - rad * 1.5), (xc + rad * 1.5, yt, zc + rad * 1.5))

    # arches between neighbouring ground-storey columns (half tori), both rows
    au, av = q(28), q(10)
    for zc in rows:
        for k in range(ncol - 1):
            xa, xb = col_x[k], col_x[k + 1]
            cx, R = 0.5 * (xa + xb), 0.5 * (xb - xa) - 0.1

            def arch(u, v, cx=cx, R=R, zc=zc):
                th = np.pi * u
                ph = TAU * v
                rr = R + 0.22 * np.cos(ph)
                return cx - rr * np.cos(th), (Y1 - 0.3 - R - 0.25) + rr * np.sin(th), zc + 0.22 * np.sin(ph)
            mb.surface(arch, au, av)

    # balustrade on both galleries: rail boxes + turned balusters
    nbal = q(90)
    bseg, bring = q(10), q(8)
    for zc in (Z0 + 3.0, Z1 - 3.0):
        mb.box((X0, Y1 + 1.0, zc - 0.08), (X1, Y1 + 1.12, zc + 0.08))
        for bx in np.linspace(X0 + 0.3, X1 - 0.3, nbal):
            def bal(u, v, bx=bx, zc=zc):
                a = TAU * u
                r = 0.05 + 0.035 * np.sin(np.pi * v) ** 2 + 0.02 * np.cos(3.0 * np.pi * v) ** 2
                return bx + r * np.cos(a), Y1 + 1.0 * v, zc - r * np.sin(a)
            mb.surface(bal, bseg, bring)

    # curtains hanging between upper-storey columns: two-sided wavy sheets
    cu, cv = q(46), q(46)
    for k in range(1, ncol - 1, 2):
        for zi, zc in enumerate(rows):
            xa, xb = col_x[k] + 0.35, col_x[k + 1] - 0.35
            ph = TAU * _hash01(np.array([k * 2 + zi]), seed + 21)[0]

            def sheet(off):
                def f(u, v, xa=xa, xb=xb, zc=zc, ph=ph, off=off):
                    w = 0.18 * np.sin(9.0 * np.pi * u + ph) * (0.3 + 0.7 * v) + 0.05 * np.sin(23.0 * u + 5.0 * v + ph)
                    return xa + (xb - xa) * u, (Y1 + 4.0) - 2.8 * v, zc + w + off
                return f
            mb.surface(sheet(+0.012), cu, cv, flip=False)
            mb.surface(sheet(-0.012), cu, cv, flip=True)

    # ornaments: displaced icospheres (stand-ins for vases / lion heads) on the floor axis
    lvl = 4 if detail >= 0.75 else (3 if detail >= 0.4 else 2)
    sp, sf = icosphere(lvl)
    for k, xc in enumerate(np.linspace(X0 + 4.0, X1 - 4.0, 6)):
        d = 1.0 + 0.12 * np.sin(7.0 * sp[:, 0] + k) * np.sin(5.0 * sp[:, 1] + 2 * k) * np.sin(6.0 * sp[:, 2])
        p = sp * d[:, None] * np.array([0.6, 0.9, 0.6]) + np.array([xc, Y0 + 0.9, 0.0 + (1.5 if k % 2 else -1.5)])
        mb.add(p, sf)

    return mb.finish()


def sponza_camera():
    """Camera inside the atrium looking down the long axis and slightly up, so the
    frame holds floor, both colonnades, curtains and a strip of open sky."""
    return dict(eye=(-13.5, -1.2, 0.6), at=(6.0, 1.4, -0.4), up=(0.0, 1.0, 0.0), fov=float(np.float32(np.pi / 4)))


# ---------------------------------------------------------------------------
# C4 stand-in: many instances of one small mesh; C5: displaced grid
# ---------------------------------------------------------------------------

def blob_mesh(seed=3, level=3):
    """~1.3k-triangle lumpy closed mesh (stand-in for susanne.obj, 968 tris)."""
    sp, sf = icosphere(level)
    d = 1.0 + 0.25 * np.sin(3.0 * sp[:, 0] + seed) * np.cos(4.0 * sp[:, 1]) + 0.15 * np.sin(5.0 * sp[:, 2] + 2.0 * seed)
    mb = MeshBuilder()
    mb.add(sp * d[:, None], sf)
    return mb.finish()


def instance_grid(n_side, spacing=3.0, seed=9):
    """n_side^2 rigid transforms (3x4 row-major, float32) on a grid in the xz plane with
    seeded rotations about y and x and a uniform scale in [0.7,1.3]."""
    k = np.arange(n_side * n_side)
    a = 2.0 * np.pi * _hash01(k, seed)
    b = 0.6 * (_hash01(k, seed + 1) - 0.5)
    s = 0.7 + 0.6 * _hash01(k, seed + 2)
    ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
    ry = np.zeros((k.size, 3, 3)); rx = np.zeros((k.size, 3, 3))
    ry[:, 0, 0] = ca; ry[:, 0, 2] = sa; ry[:, 1, 1] = 1; ry[:, 2, 0] = -sa; ry[:, 2, 2] = ca
    rx[:, 0, 0] = 1; rx[:, 1, 1] = cb; rx[:, 1, 2] = -sb; rx[:, 2, 1] = sb; rx[:, 2, 2] = cb
    r = np.einsum("nij,njk->nik", ry, rx) * s[:, None, None]
    m = np.zeros((k.size, 3, 4))
    m[:, :, :3] = r
    m[:, 0, 3] = ((k % n_side) - 0.5 * (n_side - 1)) * spacing
    m[:, 2, 3] = ((k // n_side) - 0.5 * (n_side - 1)) * spacing
    return m.astype(np.float32).reshape(-1, 12)


def displaced_grid(n_side, seed=7, extent=40.0):
    """2*n_side^2 triangles: a terrain with multi-octave relief (C5 uses n_side=2236 -> 10.0 M)."""
    u, v = np.meshgrid(np.linspace(0.0, 1.0, n_side + 1, dtype=np.float64),
                       np.linspace(0.0, 1.0, n_side + 1, dtype=np.float64), indexing="xy")
    u, v = u.ravel(), v.ravel()
    h = np.zeros_like(u)
    amp, f = 3.0, 2.0
    for o in range(6):
        h += amp * np.sin(f * 2.0 * np.pi * u + 1.3 * o + seed) * np.cos(f * 2.0 * np.pi * v + 0.7 * o)
        amp *= 0.5
        f *= 2.1
    pos = np.stack([(u - 0.5) * extent, h - 4.0, (0.5 - v) * extent], 1)
    tri = _grid_indices(n_side, n_side)
    # normals from finite differences of the height field (np.add.at over 30 M corners is far too slow)
    hg = h.reshape(n_side + 1, n_side + 1)
    step = extent / n_side
    dhdx = np.gradient(hg, step, axis=1)
    dhdz = -np.gradient(hg, step, axis=0)            # z runs against v
    nrm = np.stack([-dhdx, np.ones_like(hg), -dhdz], -1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    vv = np.zeros(pos.shape[0], VERTEX)
    vv["position"] = pos.astype(np.float32)
    vv["normal"] = nrm.astype(np.float32)
    return vv, tri


# ---------------------------------------------------------------------------
# Stress scene with real-asset triangle statistics ("teapot in a stadium", round 5, VERDICT r4 task 6)
# ---------------------------------------------------------------------------

def stadium_class(seed=5, scale=1.0, parts=("hall", "pots", "cables", "slats", "debris")):
    """A hall of a FEW huge quads with dense detail standing on them, cables and slats of 20:1 ... 2000:1 slivers through the air, and a
    debris field whose triangle areas are log-normal over four decades: what a scanned or modelled asset looks like to a BVH builder and
    what no uniformly tessellated procedural mesh does (every timed scene of rounds 1 - 4; the reference's default scene is a real
    asset, src/DXRExperimentsApp.cpp:91, Machines.fbx, not in the checkout).  scale = 1 -> about 262 k triangles (the headline
    scene's count), scale = 8 -> about 2 M.  Hall: x in [-30, 30], z in [-20, 20], floor y = -4, roof y = 14.
    Composition at scale 1: 16 hall triangles (floor, four walls, two roof halves with a slot of sky between them: 600 - 2400 m^2 each);
    ~110 k triangles in 7 'teapots' (displaced icospheres of 2 - 80 k triangles, 0.3 - 2 m across, mm- to cm-sized triangles) on the floor
    and on 3 plinth boxes; ~12 k in 1500 cables (strips 15 - 55 m long, 6 - 30 mm wide, four segments each: slivers of ~1000:1) and ~30 k
    in two slatted fences and a grandstand of long steps (slats 20:1 ... 60:1); ~110 k debris triangles, log-normal edge length with
    sigma = 1.15 (areas over four decades, 1 mm^2 ... 10 m^2), 70 % of them in eight clusters, isotropic orientation, each with a
    random 1:1 ... 20:1 aspect.  parts: which of the components to build (experiments: what each one costs a builder)."""
    rng = np.random.default_rng(seed)
    mb = MeshBuilder()
    X0, X1, Z0, Z1, Y0, Y1 = -30.0, 30.0, -20.0, 20.0, -4.0, 14.0

    def quad(a, b, c, d):                  # two triangles, front face = cross(b - a, c - a)
        p = np.array([a, b, c, d], np.float64)
        n = np.cross(p[1] - p[0], p[2] - p[0])
        mb.add(p, [[0, 1, 2], [0, 2, 3]], np.tile(n / np.linalg.norm(n), (4, 1)))
    # the hall: huge quads, normals into the room
    quad((X0, Y0, Z1), (X1, Y0, Z1), (X1, Y0, Z0), (X0, Y0, Z0))                      # floor (+y)
    quad((X0, Y0, Z0), (X1, Y0, Z0), (X1, Y1, Z0), (X0, Y1, Z0))                      # -z wall (+z)
    quad((X1, Y0, Z1), (X0, Y0, Z1), (X0, Y1, Z1), (X1, Y1, Z1))                      # +z wall (-z)
    quad((X0, Y0, Z1), (X0, Y0, Z0), (X0, Y1, Z0), (X0, Y1, Z1))                      # -x wall (+x)
    quad((X1, Y0, Z0), (X1, Y0, Z1), (X1, Y1, Z1), (X1, Y1, Z0))                      # +x wall (-x)
    quad((X0, Y1, Z0), (X1, Y1, Z0), (X1, Y1, -3.0), (X0, Y1, -3.0))                  # roof halves (-y), a 6 m slot of sky between them
    quad((X0, Y1, 3.0), (X1, Y1, 3.0), (X1, Y1, Z1), (X0, Y1, Z1))

    # plinths (large quads under dense detail) and the 'teapots' on them / on the floor
    if "hall" not in parts:
        mb = MeshBuilder()
    plinths = [((-14.0, Y0, -6.0), (-8.0, Y0 + 1.2, 0.0)), ((4.0, Y0, 5.0), (9.0, Y0 + 0.8, 10.0)), ((14.0, Y0, -12.0), (22.0, Y0 + 2.0, -5.0))]
    for lo, hi in plinths:
        mb.box(lo, hi)
    lv_hi = 6 if scale >= 1.0 else 5
    pots = [(-11.0, Y0 + 1.2, -3.0, 1.0, lv_hi), (6.5, Y0 + 0.8, 7.5, 0.8, 5), (18.0, Y0 + 2.0, -8.5, 0.9, 5), (-2.0, Y0, 3.0, 0.45, 4),
            (-20.0, Y0, 12.0, 0.3, 4), (24.0, Y0, 14.0, 0.35, 3), (0.0, Y0, -14.0, 0.15, 3)]
    reps = max(1, int(round(scale))) if "pots" in parts else 0
    for rep in range(reps):
        for k, (cx, cy, cz, r, lv) in enumerate(pots):
            sp, sf = icosphere(lv)
            ph = 1.7 * k + 0.31 * rep
            d = 1.0 + 0.2 * np.sin(5.0 * sp[:, 0] + ph) * np.sin(4.0 * sp[:, 1] + 2.0 * ph) + 0.08 * np.sin(17.0 * sp[:, 2] + ph) * np.cos(13.0 * sp[:, 0])
            off = np.array([0.0, 0.0, 0.0]) if rep == 0 else np.array([rng.uniform(-3.0, 3.0), 0.0, rng.uniform(-3.0, 3.0)])
            mb.add(sp * (d * r)[:, None] * np.array([1.0, 0.8, 1.0]) + np.array([cx, cy + 0.8 * r * 0.82, cz]) + off, sf)

    # cables: long thin strips through the air (four segments each, ~1000:1 slivers), sagging a little
    n_cab = int(1500 * scale) if "cables" in parts else 0
    a = np.stack([rng.uniform(X0 + 1, X1 - 1, n_cab), rng.uniform(Y0 + 3.0, Y1 - 0.5, n_cab), rng.uniform(Z0 + 1, Z1 - 1, n_cab)], 1)
    dirn = rng.normal(size=(n_cab, 3)) * np.array([1.0, 0.12, 0.7])
    dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    length = rng.uniform(15.0, 55.0, n_cab)
    width = rng.uniform(0.006, 0.03, n_cab)
    side = np.cross(dirn, np.array([0.0, 1.0, 0.0]))
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    t = np.linspace(-0.5, 0.5, 5)
    for k in range(n_cab):
        c = a[k] + dirn[k] * (t * length[k])[:, None]
        c[:, 1] -= 0.6 * (1.0 - (2.0 * t) ** 2)                      # sag
        c = np.clip(c, [X0 + 0.05, Y0 + 0.05, Z0 + 0.05], [X1 - 0.05, Y1 - 0.05, Z1 - 0.05])
        p = np.concatenate([c - 0.5 * width[k] * side[k], c + 0.5 * width[k] * side[k]])
        tri = [[i, i + 1, 6 + i] for i in range(4)] + [[i, 6 + i, 5 + i] for i in range(4)]
        mb.add(p, tri)

    # slatted fences (20:1 ... 60:1 slats) and a grandstand of long steps
    for zc, n_slats in ((-16.0, int(700 * scale)), (15.0, int(500 * scale))) if "slats" in parts else ():
        for xs in np.linspace(X0 + 2.0, X1 - 2.0, n_slats):
            h = 1.2 + 1.8 * _hash01(np.array([int(xs * 1000)]), seed + 3)[0]
            mb.box((xs - 0.025, Y0, zc - 0.02), (xs + 0.025, Y0 + h, zc + 0.02))
    for k in range(int(40 * min(scale, 2.0)) if "slats" in parts else 0):
        y = Y0 + 0.25 * k
        z = Z0 + 0.5 + 0.3 * k
        mb.box((X0 + 3.0, y, z), (X1 - 3.0, y + 0.25, z + 0.3))

    # debris: log-normal edge lengths (sigma 1.15 -> areas over ~4 decades), clustered, isotropic, aspect 1:1 ... 20:1
    n_deb = int(110000 * scale) if "debris" in parts else 1
    centres = np.stack([rng.uniform(X0 + 4, X1 - 4, 8), rng.uniform(Y0, Y0 + 6.0, 8), rng.uniform(Z0 + 4, Z1 - 4, 8)], 1)
    which = rng.integers(0, 8, n_deb)
    clustered = rng.random(n_deb) < 0.7
    c = np.where(clustered[:, None], centres[which] + rng.normal(size=(n_deb, 3)) * np.array([2.5, 1.2, 2.5]),
                 np.stack([rng.uniform(X0 + 0.5, X1 - 0.5, n_deb), rng.uniform(Y0 + 0.05, Y1 - 0.5, n_deb), rng.uniform(Z0 + 0.5, Z1 - 0.5, n_deb)], 1))
    edge = np.clip(np.exp(rng.normal(np.log(0.06), 1.15, n_deb)), 0.0015, 4.5)
    aspect = np.exp(rng.uniform(0.0, np.log(20.0), n_deb))
    u = rng.normal(size=(n_deb, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    w = np.cross(u, rng.normal(size=(n_deb, 3))); w /= np.linalg.norm(w, axis=1, keepdims=True)
    p0 = c - 0.5 * edge[:, None] * u
    p1 = c + 0.5 * edge[:, None] * u
    p2 = c + (edge / aspect)[:, None] * w + (rng.uniform(-0.5, 0.5, n_deb) * edge)[:, None] * u
    pts = np.clip(np.stack([p0, p1, p2], 1), [X0 + 0.02, Y0 + 0.02, Z0 + 0.02], [X1 - 0.02, Y1 - 0.02, Z1 - 0.02]).reshape(-1, 3)
    fn = np.cross(pts[1::3] - pts[0::3], pts[2::3] - pts[0::3])
    ln = np.linalg.norm(fn, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    mb.add(pts, np.arange(3 * n_deb, dtype=np.uint32).reshape(-1, 3), np.repeat(fn / ln, 3, axis=0))
    return mb.finish()


def stadium_camera():
    """From a corner of the hall, 2.5 m above the floor, across the plinths and the debris clusters towards the far corner: the frame holds
    the floor and two walls (a few huge triangles), the dense meshes, cables against the roof slot's sky."""
    return dict(eye=(-26.0, -1.5, 16.0), at=(8.0, -1.0, -6.0), up=(0.0, 1.0, 0.0), fov=float(np.float32(np.pi / 4)))


def triangle_statistics(verts, tris):
    """(areas, aspect) per triangle: area, and longest edge^2 / (2 * area) -- 1.15 for an equilateral triangle, ~L / h for a sliver."""
    p = verts["position"].astype(np.float64)
    a, b, c = p[tris[:, 0]], p[tris[:, 1]], p[tris[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    longest = np.maximum(np.maximum(((b - a) ** 2).sum(1), ((c - b) ** 2).sum(1)), ((a - c) ** 2).sum(1))
    return area, longest / np.maximum(2.0 * area, 1e-300)


# ---------------------------------------------------------------------------
# OBJ writer (fixtures) and a procedural sky cube map
# ---------------------------------------------------------------------------

def write_obj(path, verts, tri, header=""):
    """One v/vn per vertex, faces 'f a//a b//b c//c' in primitive order."""
    with open(path, "w") as f:
        if header:
            for line in header.splitlines():
                f.write("# %s\n" % line)
        for p in verts["position"]:
            f.write("v %s %s %s\n" % tuple(repr(float(x)) for x in p))
        for n in verts["normal"]:
            f.write("vn %s %s %s\n" % tuple(repr(float(x)) for x in n))
        for a, b, c in np.asarray(tri) + 1:
            f.write("f %d//%d %d//%d %d//%d\n" % (a, a, b, b, c, c))


def sky_cubemap(size=64):
    """6 x size x size x 4 float32 faces (+X -X +Y -Y +Z -Z): a smooth sky gradient
    with a soft sun, standing in for CathedralRadiance.dds (4 MB, not shipped)."""
    faces = np.zeros((6, size, size, 4), np.float32)
    s = (np.arange(size) + 0.5) / size * 2.0 - 1.0
    sc, tc = np.meshgrid(s, s, indexing="xy")
    one = np.ones_like(sc)
    dirs = [(one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one)]
    sun = np.array([0.3, 0.8, 0.52]); sun /= np.linalg.norm(sun)
    for k, (x, y, z) in enumerate(dirs):
        l = np.sqrt(x * x + y * y + z * z)
        x, y, z = x / l, y / l, z / l
        up = np.clip(y * 0.5 + 0.5, 0, 1)
        col = np.stack([0.35 + 0.25 * up, 0.45 + 0.35 * up, 0.55 + 0.45 * up], -1)
        d = np.clip(x * sun[0] + y * sun[1] + z * sun[2], 0, 1)
        col += (d ** 64)[..., None] * np.array([6.0, 5.0, 3.5])
        faces[k, :, :, :3] = col
        faces[k, :, :, 3] = 1.0
    return faces
