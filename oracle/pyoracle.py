"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (dxrexperiments_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VERTEX = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3)])
BVH_NODE = np.dtype([("bmin", "<f4", 3), ("left", "<u4"), ("bmax", "<f4", 3), ("right", "<u4")])
PFC_BYTES = 188
MATERIAL_BYTES = 64

FN = dict(sin=0, cos=1, exp=2, log=3, pow=4, sqrt=5, div=6, min=7, max=8)
SAMPLE = dict(cos=0, uniform=1, phong=2, perp=3)


class RenderStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits",
                 "nodes", "tris", "shaded_hits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    """Compile oracle/liboracle.so with the committed Makefile (g++)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "dxr_amd_types.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        p = C.c_void_p
        L.orc_init_rand.restype = C.c_uint32
        L.orc_init_rand.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_next_rand.restype = C.c_float
        L.orc_next_rand.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_math_batch.argtypes = [C.c_int, p, p, p, C.c_size_t]
        L.orc_sample_batch.argtypes = [C.c_int, p, p, C.c_float, p, p, p, C.c_size_t]
        L.orc_fresnel.argtypes = [p, p, p, p]
        L.orc_sample_cube.argtypes = [p, C.c_int, p, p, C.c_size_t]
        L.orc_set_cube_seamless.argtypes = [C.c_int]
        L.orc_obj_load.argtypes = [C.c_char_p, C.POINTER(p), C.POINTER(C.c_uint32), C.POINTER(p), C.POINTER(C.c_uint32)]
        L.orc_free.argtypes = [p]
        L.orc_scene_create.restype = p
        L.orc_scene_destroy.argtypes = [p]
        L.orc_scene_add_model.argtypes = [p, p, C.c_uint32, p, C.c_uint32]
        L.orc_scene_add_instance.argtypes = [p, C.c_uint32, p]
        L.orc_scene_build.argtypes = [p]
        L.orc_scene_bvh_info.argtypes = [p, C.c_int] + [C.POINTER(C.c_uint32)] * 3
        L.orc_scene_bvh_read.argtypes = [p, C.c_int, p, p, p]
        L.orc_set_split_refs.argtypes = [C.c_int]
        L.orc_set_split_refs.restype = None
        L.orc_scene_refs_info.argtypes = [p, C.c_uint32, p]
        L.orc_scene_refs_read.argtypes = [p, C.c_uint32, p, p]
        L.orc_scene_instance_info.argtypes = [p, C.c_uint32, p, p]
        L.orc_trace.argtypes = [p, p, p, C.c_size_t, C.c_uint32, C.c_int] + [p] * 7 + [C.c_int]
        L.orc_truth64_trace.argtypes = [p, p, p, C.c_size_t, C.c_uint32] + [p] * 5 + [C.c_int]
        L.orc_render.argtypes = [p, p, C.c_uint32, p, C.c_int, p, p] + [C.c_uint32] * 9 + [C.c_int, p, C.c_int, p]
        L.orc_render_realtime.argtypes = [p, p, C.c_uint32, p, C.c_int, p, p] + [C.c_uint32] * 4 + [p, p, C.c_int, p]
        L.orc_denoise.argtypes = [p, p, C.c_uint32, C.c_uint32, p, p, p, C.c_int]
        L.orc_round_to_half.argtypes = [p, p, C.c_size_t, C.c_int]
        L.orc_camera_look.argtypes = [p] * 5
        L.orc_camera_basis.argtypes = [p, p, C.c_float, C.c_float, p, p, p]
        L.orc_progressive_create.restype = p
        L.orc_progressive_create.argtypes = [C.c_uint32]
        L.orc_progressive_destroy.argtypes = [p]
        L.orc_progressive_options.restype = p
        L.orc_progressive_options.argtypes = [p]
        L.orc_progressive_set_flags.argtypes = [p, C.c_int, C.c_int]
        L.orc_progressive_update.argtypes = [p, p, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, p]
        _LIB = L
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def init_rand(v0, v1):
    return int(lib().orc_init_rand(v0 & 0xFFFFFFFF, v1 & 0xFFFFFFFF))


def next_rand(state):
    s = C.c_uint32(state)
    f = lib().orc_next_rand(C.byref(s))
    return int(s.value), float(f)


def math(fn, x, y=None):
    x = _f32(x)
    y = _f32(y) if y is not None else x
    out = np.empty_like(x)
    lib().orc_math_batch(FN[fn], _ptr(x), _ptr(y), _ptr(out), x.size)
    return out


def sample(kind, seeds, vecs, exponent=0.0):
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    vecs = _f32(vecs).reshape(-1, 3)
    n = seeds.size
    out = np.empty((n, 3), np.float32)
    pb = np.empty((n, 2), np.float32)
    so = np.empty(n, np.uint32)
    lib().orc_sample_batch(SAMPLE[kind], _ptr(seeds), _ptr(vecs), exponent, _ptr(out), _ptr(pb), _ptr(so), n)
    return out, pb, so


def fresnel(I, N, f0):
    I, N, f0 = _f32(I), _f32(N), _f32(f0)
    out = np.empty(3, np.float32)
    lib().orc_fresnel(_ptr(I), _ptr(N), _ptr(f0), _ptr(out))
    return out


def set_cube_seamless(on=True):
    """Cube-map filter of sample_cube / render / render_realtime: cross-face taps (default) or clamped to the face."""
    lib().orc_set_cube_seamless(1 if on else 0)


def sample_cube(faces, dirs):
    faces = _f32(faces)
    dirs = _f32(dirs).reshape(-1, 3)
    out = np.empty_like(dirs)
    lib().orc_sample_cube(_ptr(faces), faces.shape[1], _ptr(dirs), _ptr(out), dirs.shape[0])
    return out


def obj_load(path):
    L = lib()
    v, i = C.c_void_p(), C.c_void_p()
    nv, nt = C.c_uint32(), C.c_uint32()
    rc = L.orc_obj_load(path.encode(), C.byref(v), C.byref(nv), C.byref(i), C.byref(nt))
    if rc != 0:
        raise IOError("orc_obj_load(%s) -> %d" % (path, rc))
    verts = np.frombuffer(C.string_at(v, nv.value * 24), dtype=VERTEX).copy()
    idx = np.frombuffer(C.string_at(i, nt.value * 12), dtype=np.uint32).copy().reshape(-1, 3)
    L.orc_free(v)
    L.orc_free(i)
    return verts, idx


IDENTITY = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)


def set_split_refs(on):
    """scenes built from now on: False = no triangle is split into references (the candidate rule of rounds 1 - 4)"""
    lib().orc_set_split_refs(1 if on else 0)


class Scene:
    def __init__(self):
        self.h = C.c_void_p(lib().orc_scene_create())
        self.n_models = 0
        self.n_instances = 0

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None and lib is not None:
            _LIB.orc_scene_destroy(self.h)
            self.h = None

    def add_model(self, verts, idx):
        verts = np.ascontiguousarray(verts, dtype=VERTEX)
        idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1, 3)
        r = lib().orc_scene_add_model(self.h, _ptr(verts), verts.shape[0], _ptr(idx), idx.shape[0])
        if r < 0:
            raise ValueError("orc_scene_add_model -> %d" % r)
        self.n_models += 1
        return r

    def add_instance(self, model, xform=None):
        x = _f32(IDENTITY if xform is None else xform).reshape(12)
        r = lib().orc_scene_add_instance(self.h, model, _ptr(x))
        if r < 0:
            raise ValueError("orc_scene_add_instance -> %d" % r)
        self.n_instances += 1
        return r

    def build(self):
        lib().orc_scene_build(self.h)

    def bvh(self, which):
        """which = -1 for the TLAS, else model index. -> (nodes, keys, parents, max_depth)"""
        n, nn, md = C.c_uint32(), C.c_uint32(), C.c_uint32()
        if lib().orc_scene_bvh_info(self.h, which, C.byref(n), C.byref(nn), C.byref(md)) != 0:
            raise RuntimeError("scene not built")
        nodes = np.empty(nn.value, BVH_NODE)
        keys = np.empty(n.value, np.uint64)
        parents = np.empty(nn.value, np.uint32)
        lib().orc_scene_bvh_read(self.h, which, _ptr(nodes), _ptr(keys), _ptr(parents))
        return nodes, keys, parents, md.value

    def refs(self, model, n_tris):
        """the validation boxes of a model's triangles: (off uint32[n_tris + 1], boxes float32[n_refs, 6]); (None, None) if none is split"""
        n = C.c_uint32()
        if lib().orc_scene_refs_info(self.h, model, C.byref(n)) != 0:
            raise RuntimeError("scene not built")
        if n.value == 0:
            return None, None
        off = np.empty(n_tris + 1, np.uint32)
        boxes = np.empty((n.value, 6), np.float32)
        lib().orc_scene_refs_read(self.h, model, _ptr(off), _ptr(boxes))
        return off, boxes

    def instance_info(self, i):
        box = np.empty(6, np.float32)
        inv = np.empty(12, np.float32)
        lib().orc_scene_instance_info(self.h, i, _ptr(box), _ptr(inv))
        return box, inv

    def trace(self, origin_tmin, dir_tmax, flags=0, mode=1, nthreads=1):
        o = _f32(origin_tmin).reshape(-1, 4)
        d = _f32(dir_tmax).reshape(-1, 4)
        n = o.shape[0]
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32); inst = np.empty(n, np.uint32)
        cn = np.zeros(n, np.uint32); ct = np.zeros(n, np.uint32)
        rc = lib().orc_trace(self.h, _ptr(o), _ptr(d), n, flags, mode, _ptr(t), _ptr(u), _ptr(v),
                             _ptr(prim), _ptr(inst), _ptr(cn), _ptr(ct), nthreads)
        if rc != 0:
            raise RuntimeError("orc_trace -> %d" % rc)
        return dict(t=t, u=u, v=v, prim=prim, inst=inst, nodes=cn, tris=ct)

    def truth64(self, origin_tmin, dir_tmax, flags=0, nthreads=8):
        """the float64 geometric truth of every ray (oracle/truth64.h: no box of any kind); t = -1 on a miss"""
        o = _f32(origin_tmin).reshape(-1, 4)
        d = _f32(dir_tmax).reshape(-1, 4)
        n = o.shape[0]
        t = np.empty(n, np.float64); u = np.empty(n, np.float64); v = np.empty(n, np.float64)
        prim = np.empty(n, np.uint32); inst = np.empty(n, np.uint32)
        rc = lib().orc_truth64_trace(self.h, _ptr(o), _ptr(d), n, flags, _ptr(t), _ptr(u), _ptr(v), _ptr(prim), _ptr(inst), nthreads)
        if rc != 0:
            raise RuntimeError("orc_truth64_trace -> %d" % rc)
        return dict(t=t, u=u, v=v, prim=prim, inst=inst)

    def render_realtime(self, materials, pfc, width, height, env_faces=None, env_constant=(0.5, 0.5, 0.5),
                        max_radiance_depth=1, max_shadow_depth=2, nthreads=1):
        mats = np.ascontiguousarray(materials)
        pfc = np.ascontiguousarray(pfc)
        assert pfc.nbytes == PFC_BYTES
        ef = _f32(env_faces) if env_faces is not None else None
        ec = _f32(env_constant)
        direct = np.zeros((height, width, 4), np.float32)
        indirect = np.zeros((height, width, 4), np.float32)
        st = RenderStats()
        rc = lib().orc_render_realtime(self.h, _ptr(mats), mats.nbytes // MATERIAL_BYTES, _ptr(ef), 0 if ef is None else ef.shape[1],
                                       _ptr(ec), _ptr(pfc), width, height, max_radiance_depth, max_shadow_depth,
                                       _ptr(direct), _ptr(indirect), nthreads, C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render_realtime -> %d" % rc)
        return direct, indirect, st.as_dict()

    def render(self, materials, pfc, width, height, accum=None, env_faces=None, env_constant=(0.5, 0.5, 0.5),
               tile=None, accum_mode=0, max_radiance_depth=1, max_shadow_depth=2, use_brute=False, nthreads=1, accum_f16=0):
        """accum_f16: 1 / 2 = the running mean is rounded to fp16 every frame (nearest even / toward zero): the reference's RGBA16F storage"""
        mats = np.ascontiguousarray(materials)
        assert mats.nbytes % MATERIAL_BYTES == 0
        pfc = np.ascontiguousarray(pfc)
        assert pfc.nbytes == PFC_BYTES
        if accum is None:
            accum = np.zeros((height, width, 4), np.float32)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.size == width * height * 4
        ef = _f32(env_faces) if env_faces is not None else None
        ec = _f32(env_constant)
        x0, y0, x1, y1 = tile if tile else (0, 0, width, height)
        st = RenderStats()
        rc = lib().orc_render(self.h, _ptr(mats), mats.nbytes // MATERIAL_BYTES,
                              _ptr(ef), 0 if ef is None else ef.shape[1], _ptr(ec), _ptr(pfc),
                              width, height, x0, y0, x1, y1, accum_mode | (accum_f16 << 8), max_radiance_depth, max_shadow_depth,
                              int(use_brute), _ptr(accum), nthreads, C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render -> %d" % rc)
        return accum, st.as_dict()


DENOISE_PARAMS = np.dtype([("exposure", "<f4"), ("gamma", "<f4"), ("tonemap", "<u4"), ("gammaCorrect", "<u4"),
                           ("maxKernelSize", "<i4"), ("debugVisualize", "<u4")])


def denoise(direct, indirect, params, nthreads=8):
    """DenoiseCompositor: returns (H-pass image, final V-pass image)."""
    d = _f32(direct); i = _f32(indirect)
    h, w = d.shape[0], d.shape[1]
    prm = np.ascontiguousarray(params, dtype=DENOISE_PARAMS)
    oh = np.empty((h, w, 4), np.float32); ov = np.empty((h, w, 4), np.float32)
    rc = lib().orc_denoise(_ptr(d), _ptr(i), w, h, _ptr(prm), _ptr(oh), _ptr(ov), nthreads)
    if rc != 0:
        raise RuntimeError("orc_denoise -> %d" % rc)
    return oh, ov


def round_to_half(x, nearest=True):
    """float32 -> float16 -> float32 by the oracle's own conversion (tests pin it against numpy's)"""
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_round_to_half(_ptr(x), _ptr(out), x.size, 1 if nearest else 0)
    return out


def camera_look(eye, at, up):
    eye, at, up = _f32(eye), _f32(at), _f32(up)
    f = np.empty(3, np.float32); u = np.empty(3, np.float32)
    lib().orc_camera_look(_ptr(eye), _ptr(at), _ptr(up), _ptr(f), _ptr(u))
    return f, u


def camera_basis(forward, up, fov, aspect):
    forward, up = _f32(forward), _f32(up)
    U = np.empty(4, np.float32); V = np.empty(4, np.float32); W = np.empty(4, np.float32)
    lib().orc_camera_basis(_ptr(forward), _ptr(up), fov, aspect, _ptr(U), _ptr(V), _ptr(W))
    return U, V, W


class Progressive:
    """Host-side update() state machine of the oracle."""

    def __init__(self, rng_seed=1234):
        self.h = C.c_void_p(lib().orc_progressive_create(rng_seed))

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None and lib is not None:
            _LIB.orc_progressive_destroy(self.h)
            self.h = None

    def options_buffer(self):
        p = lib().orc_progressive_options(self.h)
        return (C.c_uint8 * 44).from_address(p)

    def set_flags(self, accumulation_enabled=True, animation_paused=True):
        lib().orc_progressive_set_flags(self.h, int(accumulation_enabled), int(animation_paused))

    def update(self, camera11, elapsed_time, elapsed_frames, width, height):
        cam = _f32(camera11).reshape(11)
        out = np.zeros(PFC_BYTES, np.uint8)
        lib().orc_progressive_update(self.h, _ptr(cam), elapsed_time, elapsed_frames, width, height, _ptr(out))
        return out
