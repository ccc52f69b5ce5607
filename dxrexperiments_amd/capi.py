"""ctypes binding of the C ABI declared in include/dxr_amd.h.

This is the ONLY compute path of the package: every call goes to the HIP
library dxrexperiments_amd/lib/libdxrexperiments_amd.so.  There is no CPU
fallback; if the library is missing or no GPU is usable the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import rtypes as T

_HERE = os.path.dirname(os.path.abspath(__file__))
# DXR_AMD_LIB selects an alternative build of the same library (kernel experiments)
LIB_PATH = os.environ.get("DXR_AMD_LIB") or os.path.join(_HERE, "lib", "libdxrexperiments_amd.so")
_LIB = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dxr_amd error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits")] + \
               [(n, C.c_float) for n in ("ms_primary", "ms_shade0", "ms_trace_secondary", "ms_trace_shadow0", "ms_shade1",
                                          "ms_trace_shadow1", "ms_resolve", "ms_total")] + [("frames", C.c_uint64), ("rays_shadow_skipped", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class StageWork(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("tris", C.c_uint64)]


class StageWalk(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_global", C.c_uint64), ("nodes_lds", C.c_uint64), ("tris", C.c_uint64),
                ("instance_entries", C.c_uint64), ("lines", C.c_uint64), ("longest_walk", C.c_uint64), ("longest_walk_ray", C.c_uint64),
                ("wave_node_steps", C.c_uint64), ("wave_leaf_phases", C.c_uint64), ("wave_tri_steps", C.c_uint64),
                ("node_lines", C.c_uint64)]


STAGES = ("primary", "secondary", "shadow0", "shadow1")

PIPELINE_PROGRESSIVE, PIPELINE_REALTIME = 0, 1
CUBE_SEAMLESS, CUBE_FACE_CLAMP = 0, 1
DENOISER_PARAMS = np.dtype([("exposure", "<f4"), ("gamma", "<f4"), ("tonemap", "<u4"), ("gammaCorrect", "<u4"),
                            ("maxKernelSize", "<i4"), ("debugVisualize", "<u4")])

# name -> (restype, argtypes); every symbol include/dxr_amd.h declares
_p, _u32, _i, _f, _sz = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_size_t
_pp = C.POINTER(C.c_void_p)
_pu = C.POINTER(C.c_uint32)
SIGNATURES = {
    "rt_version": (C.c_char_p, []),
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (_i, []),
    "rt_context_create": (_i, [_i, _pp]),
    "rt_context_create_on_stream": (_i, [_i, _p, _pp]),
    "rt_context_destroy": (_i, [_p]),
    "rt_context_synchronize": (_i, [_p]),
    "rt_context_get_stack_memory": (_i, [_p, _p]),
    "rt_context_get_stream": (_i, [_p, _pp]),
    "rt_context_get_device": (_i, [_p, C.POINTER(_i)]),
    "rt_device_alloc": (_i, [_p, _sz, _pp]),
    "rt_device_free": (_i, [_p, _p]),
    "rt_device_upload": (_i, [_p, _p, _p, _sz]),
    "rt_device_download": (_i, [_p, _p, _p, _sz]),
    "rt_model_create_from_obj": (_i, [_p, C.c_char_p, _pp]),
    "rt_model_create_from_arrays": (_i, [_p, _p, _u32, _p, _u32, _pp]),
    "rt_model_get_counts": (_i, [_p, _pu, _pu]),
    "rt_model_read_geometry": (_i, [_p, _p, _p]),
    "rt_model_retain": (_i, [_p]),
    "rt_model_destroy": (_i, [_p]),
    "rt_scene_create": (_i, [_p, _pp]),
    "rt_scene_add_model": (_i, [_p, _p, _p]),
    "rt_scene_get_num_instances": (_i, [_p, _pu]),
    "rt_scene_build": (_i, [_p, _u32]),
    "rt_scene_destroy": (_i, [_p]),
    "rt_scene_bvh_info": (_i, [_p, _i, _pu, _pu, _pu]),
    "rt_scene_bvh_read": (_i, [_p, _i, _p, _p, _p]),
    "rt_scene_instance_info": (_i, [_p, _u32, _p, _p]),
    "rt_wide_layout_info": (_i, [_pu, _pu]),
    "rt_scene_wide_info": (_i, [_p, _i, _pu, C.POINTER(C.c_int32), _pu]),
    "rt_scene_wide_read": (_i, [_p, _i, _p, _p]),
    "rt_debug_wide_write": (_i, [_p, _i, _p, _u32]),
    "rt_scene_build_ms": (_i, [_p, C.POINTER(_f)]),
    "rt_trace_batch": (_i, [_p, _p, _p, _p, _sz, _u32, _u32, _u32, _p, _p, _p, _p, _p, _p, _p]),
    "rt_trace_last_ms": (_i, [_p, C.POINTER(_f)]),
    "rt_pipeline_create": (_i, [_p, _u32, _pp]),
    "rt_pipeline_destroy": (_i, [_p]),
    "rt_pipeline_get_name": (C.c_char_p, [_p]),
    "rt_pipeline_set_scene": (_i, [_p, _p]),
    "rt_pipeline_add_material": (_i, [_p, _p]),
    "rt_pipeline_set_material": (_i, [_p, _u32, _p]),
    "rt_pipeline_set_environment_cube": (_i, [_p, _p, _u32]),
    "rt_pipeline_set_environment_constant": (_i, [_p, _p]),
    "rt_pipeline_load_environment_dds": (_i, [_p, C.c_char_p]),
    "rt_pipeline_create_output": (_i, [_p, _u32, _u32, _u32]),
    "rt_pipeline_bind_output": (_i, [_p, _p, _u32, _u32]),
    "rt_pipeline_build_acceleration_structures": (_i, [_p]),
    "rt_pipeline_set_depth_limits": (_i, [_p, _u32, _u32]),
    "rt_pipeline_set_skip_unlit_shadow_rays": (_i, [_p, _i]),
    "rt_pipeline_set_accumulation_mode": (_i, [_p, _u32]),
    "rt_pipeline_set_accumulation_storage": (_i, [_p, _u32, _u32]),
    "rt_pipeline_clear_output": (_i, [_p]),
    "rt_pipeline_update": (_i, [_p, _p]),
    "rt_pipeline_render": (_i, [_p, _u32, _u32]),
    "rt_pipeline_set_shadow_cache": (_i, [_p, _i]),
    "rt_pipeline_get_shadow_cache": (_i, [_p, _p]),
    "rt_pipeline_get_free_sphere": (_i, [_p, _p]),
    "rt_pipeline_render_batch": (_i, [_p, _u32, _u32, _p, _u32]),
    "rt_pipeline_reserve_batch": (_i, [_p, _u32, _u32, _u32]),
    "rt_pipeline_render_tile": (_i, [_p, _u32, _u32, _u32, _u32, _u32, _u32]),
    "rt_pipeline_get_num_outputs": (_i, [_p, C.POINTER(_i)]),
    "rt_pipeline_get_output_device_ptr": (_i, [_p, _u32, _pp]),
    "rt_pipeline_read_output": (_i, [_p, _p, _sz]),
    "rt_pipeline_read_output_n": (_i, [_p, _u32, _p, _sz]),
    "rt_pipeline_write_output": (_i, [_p, _p, _sz]),
    "rt_pipeline_save_checkpoint": (_i, [_p, _p, C.c_char_p]),
    "rt_pipeline_load_checkpoint": (_i, [_p, _p, C.c_char_p]),
    "rt_realtime_host_update": (_i, [_p, _p, _f, _u32, _u32, _u32, _p]),
    "rt_denoiser_create": (_i, [_p, _pp]),
    "rt_denoiser_destroy": (_i, [_p]),
    "rt_denoiser_get_params": (_i, [_p, _pp]),
    "rt_denoiser_create_output": (_i, [_p, _u32, _u32, _u32]),
    "rt_denoiser_dispatch": (_i, [_p, _p, _p, _u32, _u32]),
    "rt_denoiser_get_output_device_ptr": (_i, [_p, _pp]),
    "rt_denoiser_read_output": (_i, [_p, _p, _sz]),
    "rt_denoiser_read_intermediate": (_i, [_p, _p, _sz]),
    "rt_denoiser_last_ms": (_i, [_p, C.POINTER(_f)]),
    "rt_pipeline_get_stats": (_i, [_p, C.POINTER(Stats)]),
    "rt_pipeline_enable_timing": (_i, [_p, _i]),
    "rt_pipeline_read_primary_hits": (_i, [_p, _p, _p, _p]),
    "rt_pipeline_get_totals": (_i, [_p, C.POINTER(Stats)]),
    "rt_pipeline_reset_totals": (_i, [_p]),
    "rt_pipeline_count_work": (_i, [_p, C.POINTER(StageWork)]),
    "rt_pipeline_count_walk": (_i, [_p, C.POINTER(StageWalk)]),
    "rt_pipeline_render_bands": (_i, [_p, _u32, _u32, _u32, _u32, _u32]),
    "rt_pipeline_render_bands_batch": (_i, [_p, _u32, _u32, _u32, _u32, _u32, _p, _u32]),
    "rt_pipeline_set_deferred": (_i, [_p, _u32]),
    "rt_pipeline_get_deferred": (_i, [_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_pipeline_flush": (_i, [_p]),
    "rt_pipeline_set_queue_budget": (_i, [_p, _sz]),
    "rt_pipeline_get_queue_memory": (_i, [_p, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    "rt_scene_refs_info": (_i, [_p, _i, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_scene_refs_read": (_i, [_p, _i, _p, _p, _p]),
    "rt_debug_set_alloc_limit": (_i, [_sz]),
    "rt_debug_set_option": (_i, [_p, C.c_char_p, C.c_char_p]),
    "rt_debug_repack_stats": (_i, [_p, _p]),
    "rt_debug_read_secondary_ray": (_i, [_p, _u32, _p, _p]),
    "rt_camera_look": (_i, [_p, _p, _p, _p, _p]),
    "rt_camera_basis": (_i, [_p, _p, _f, _f, _p, _p, _p]),
    "rt_progressive_host_create": (_i, [_u32, _pp]),
    "rt_progressive_host_destroy": (_i, [_p]),
    "rt_progressive_host_options": (_i, [_p, _pp]),
    "rt_progressive_host_set_flags": (_i, [_p, _i, _i]),
    "rt_image_write_pfm": (_i, [C.c_char_p, _p, _u32, _u32]),
    "rt_image_write_exr": (_i, [C.c_char_p, _p, _u32, _u32]),
    "rt_image_write_png": (_i, [C.c_char_p, _p, _u32, _u32, _f, _f, _i]),
    "rt_progressive_host_reset": (_i, [_p]),
    "rt_progressive_host_save_state": (_i, [_p, _p, _sz, C.POINTER(_sz)]),
    "rt_progressive_host_load_state": (_i, [_p, _p, _sz]),
    "rt_progressive_host_update": (_i, [_p, _p, _f, _u32, _u32, _u32, _p]),
    "rt_debug_math": (_i, [_p, _i, _p, _p, _p, _sz]),
    "rt_debug_sample": (_i, [_p, _i, _p, _p, _f, _p, _p, _p, _sz]),
    "rt_debug_sample_cube": (_i, [_p, _p, _u32, _u32, _p, _p, _sz]),
    "rt_pipeline_set_environment_filter": (_i, [_p, _u32]),
    "rt_shard_frame_count": (_i, [_u32, _u32, _u32, C.POINTER(C.c_uint32)]),
    "rt_tile_bands": (_i, [_u32, _u32, _u32, _u32, _p, _p, _u32, C.POINTER(C.c_uint32)]),
    "rt_tile_gather_layout": (_i, [_u32, _u32, _u32, _u32, C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]),
    "rt_dist_get_unique_id": (_i, [_p]),
    "rt_dist_create": (_i, [_p, _i, _i, _p, _pp]),
    "rt_dist_destroy": (_i, [_p]),
    "rt_dist_get_rank": (_i, [_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rt_dist_all_reduce_sum": (_i, [_p, _p, _sz]),
    "rt_dist_gather_bands": (_i, [_p, _p, _u32, _u32, _u32]),
    "rt_dist_last_collective_ms": (_i, [_p, C.POINTER(C.c_float)]),
    "rt_dist_device_pci_bus_id": (_i, [_p, C.c_char_p, _sz]),
    "rt_dds_read_cube": (_i, [C.c_char_p, _p, _sz, C.POINTER(C.c_uint32)]),
    "rt_obj_read": (_i, [C.c_char_p, _p, _u32, _p, _u32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_fbx_read": (_i, [C.c_char_p, _p, _u32, _p, _u32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_model_create_from_file": (_i, [_p, C.c_char_p, _pp]),
}


def lib():
    """Load the HIP library; raises if it has not been built (there is no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("HIP library %s not found: run `make` (or __graft_entry__.build()) first; "
                              "dxrexperiments_amd has no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise RtError(rc, lib().rt_last_error().decode(errors="replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


def device_count():
    n = lib().rt_device_count()
    if n < 0:
        raise RtError(n, lib().rt_last_error().decode(errors="replace"))
    return n


def wide_layout():
    """(children per traversal node, bytes per node) of the loaded library: (4, 64), or (8, 128) for a -DRT_WIDE=8 build"""
    w, b = C.c_uint32(), C.c_uint32()
    _check(lib().rt_wide_layout_info(C.byref(w), C.byref(b)))
    return w.value, b.value


class Context:
    """RtContext (libs/DXRFramework/RtContext.h:15)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        if stream is None:
            _check(lib().rt_context_create(device, C.byref(h)))
        else:
            _check(lib().rt_context_create_on_stream(device, C.c_void_p(stream), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().rt_context_destroy(self.h)
            self.h = None

    __del__ = close

    def synchronize(self):
        _check(lib().rt_context_synchronize(self.h))

    def set_option(self, name, value):
        """rt_debug_set_option: the context's experiment / test knobs (include/dxr_amd.h lists them)"""
        _check(lib().rt_debug_set_option(self.h, str(name).encode(), str(value).encode()))

    def repack_stats(self):
        """rt_debug_repack_stats: the re-packed engine's tallies since the last call (option repack=1)"""
        out = (C.c_ulonglong * 8)()
        _check(lib().rt_debug_repack_stats(self.h, out))
        names = ("node_steps", "node_step_lanes", "leaf_passes", "leaf_pass_lanes", "rays_to_leaf_queue", "rays_to_node_queue", "refills", "watchdog_aborts")
        return dict(zip(names, [int(x) for x in out]))

    def stack_memory(self):
        """bytes of the traversal kernels' global stack rows held by the context"""
        n = C.c_size_t(0)
        _check(lib().rt_context_get_stack_memory(self.h, C.byref(n)))
        return n.value

    def pci_bus_id(self):
        """PCI bus id of the context's device ("0000:c1:00.0"): what tells two ranks on one GPU apart from two GPUs"""
        buf = C.create_string_buffer(64)
        _check(lib().rt_dist_device_pci_bus_id(self.h, buf, 64))
        return buf.value.decode()

    @property
    def stream(self):
        s = C.c_void_p()
        _check(lib().rt_context_get_stream(self.h, C.byref(s)))
        return s.value or 0

    def upload(self, array):
        """Copy a numpy array into a fresh device buffer; returns a DeviceBuffer (freed on close / GC)."""
        a = np.ascontiguousarray(array)
        p = C.c_void_p()
        _check(lib().rt_device_alloc(self.h, a.nbytes, C.byref(p)))
        _check(lib().rt_device_upload(self.h, p, _ptr(a), a.nbytes))
        return DeviceBuffer(self, p.value, a.nbytes)

    # device math probes (tests)
    def math(self, fn, x, y=None):
        x = _f32(x)
        y = _f32(y) if y is not None else x
        out = np.empty_like(x)
        _check(lib().rt_debug_math(self.h, fn, _ptr(x), _ptr(y), _ptr(out), x.size))
        return out

    def sample(self, kind, seeds, vecs, exponent=0.0):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        vecs = _f32(vecs, (-1, 3))
        n = seeds.size
        out = np.empty((n, 3), np.float32)
        pb = np.empty((n, 2), np.float32)
        so = np.empty(n, np.uint32)
        _check(lib().rt_debug_sample(self.h, kind, _ptr(seeds), _ptr(vecs), exponent, _ptr(out), _ptr(pb), _ptr(so), n))
        return out, pb, so

    def sample_cube(self, faces, dirs, seamless=True):
        faces = _f32(faces)
        dirs = _f32(dirs, (-1, 3))
        out = np.empty_like(dirs)
        _check(lib().rt_debug_sample_cube(self.h, _ptr(faces), faces.shape[1], CUBE_SEAMLESS if seamless else CUBE_FACE_CLAMP,
                                          _ptr(dirs), _ptr(out), dirs.shape[0]))
        return out


class DeviceBuffer:
    def __init__(self, ctx, ptr, nbytes):
        self.ctx, self.ptr, self.nbytes = ctx, ptr, nbytes

    def download(self, dtype=np.float32):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype)
        _check(lib().rt_device_download(self.ctx.h, _ptr(out), C.c_void_p(self.ptr), self.nbytes))
        return out

    def close(self):
        if getattr(self, "ptr", None) and getattr(self.ctx, "h", None):
            lib().rt_device_free(self.ctx.h, C.c_void_p(self.ptr))
        self.ptr = None

    __del__ = close


class Model:
    """RtModel (libs/DXRFramework/RtModel.h:13)."""

    def __init__(self, ctx, verts=None, indices=None, path=None):
        self.ctx = ctx
        h = C.c_void_p()
        if path is not None:
            _check(lib().rt_model_create_from_file(ctx.h, os.fsencode(path), C.byref(h)))
        else:
            v = np.ascontiguousarray(verts, dtype=T.VERTEX)
            i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
            _check(lib().rt_model_create_from_arrays(ctx.h, _ptr(v), v.shape[0], _ptr(i), i.shape[0], C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().rt_model_destroy(self.h)
            self.h = None

    __del__ = close

    def counts(self):
        nv, nt = C.c_uint32(), C.c_uint32()
        _check(lib().rt_model_get_counts(self.h, C.byref(nv), C.byref(nt)))
        return nv.value, nt.value

    def geometry(self):
        nv, nt = self.counts()
        v = np.empty(nv, T.VERTEX)
        i = np.empty((nt, 3), np.uint32)
        _check(lib().rt_model_read_geometry(self.h, _ptr(v), _ptr(i)))
        return v, i


class Scene:
    """RtScene (libs/DXRFramework/RtScene.h:14-37)."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        _check(lib().rt_scene_create(ctx.h, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().rt_scene_destroy(self.h)
            self.h = None

    __del__ = close

    def add_model(self, model, transform=None):
        x = _f32(T.IDENTITY_3X4 if transform is None else transform, 12)
        _check(lib().rt_scene_add_model(self.h, model.h, _ptr(x)))

    @property
    def num_instances(self):
        n = C.c_uint32()
        _check(lib().rt_scene_get_num_instances(self.h, C.byref(n)))
        return n.value

    def build(self, hit_group_count=2):
        _check(lib().rt_scene_build(self.h, hit_group_count))

    def build_ms(self):
        ms = C.c_float()
        _check(lib().rt_scene_build_ms(self.h, C.byref(ms)))
        return ms.value

    def bvh(self, which):
        n, nn, md = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().rt_scene_bvh_info(self.h, which, C.byref(n), C.byref(nn), C.byref(md)))
        nodes = np.empty(nn.value, T.BVH_NODE)
        keys = np.empty(n.value, np.uint64)
        parents = np.empty(nn.value, np.uint32)
        _check(lib().rt_scene_bvh_read(self.h, which, _ptr(nodes), _ptr(keys), _ptr(parents)))
        return nodes, keys, parents, md.value

    def refs(self, which=0):
        """Split references of instance `which`'s model (rt_refs.h): (ref_off uint32[n_tris + 1], ref_boxes float32[n_refs, 6], record_boxes
        float32[n_records, 6]); (None, None, None) when no triangle of the model is split."""
        nt, nr, nn, root, nrec = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int32(), C.c_uint32()
        _check(lib().rt_scene_refs_info(self.h, which, C.byref(nt), C.byref(nr)))
        if nr.value == 0:
            return None, None, None
        _check(lib().rt_scene_wide_info(self.h, which, C.byref(nn), C.byref(root), C.byref(nrec)))
        off = np.empty(nt.value + 1, np.uint32)
        boxes = np.empty((nr.value, 6), np.float32)
        rec = np.zeros((nrec.value, 6), np.float32)
        _check(lib().rt_scene_refs_read(self.h, which, _ptr(off), _ptr(boxes), _ptr(rec)))
        return off, boxes, rec

    def wide_read(self, which=0):
        """The production traversal layout: (nodes uint32[n, 16] (64-B four-wide nodes, raw words; [n, 32] from a library built
        with -DRT_WIDE=8), root_code, records float32[m, 12] (BLAS triangle records; empty for the TLAS))."""
        n, root, m = C.c_uint32(), C.c_int32(), C.c_uint32()
        _check(lib().rt_scene_wide_info(self.h, which, C.byref(n), C.byref(root), C.byref(m)))
        nodes = np.empty((n.value, wide_layout()[1] // 4), np.uint32)
        recs = np.empty((m.value, 12), np.float32)
        _check(lib().rt_scene_wide_read(self.h, which, _ptr(nodes), _ptr(recs)))
        return nodes, root.value, recs

    def wide_counts(self, which=0):
        """(nodes, records) of the production traversal layout: which = -1 the TLAS, k >= 0 the BLAS of instance k's model"""
        n, root, m = C.c_uint32(), C.c_int32(), C.c_uint32()
        _check(lib().rt_scene_wide_info(self.h, which, C.byref(n), C.byref(root), C.byref(m)))
        return n.value, m.value

    def wide_write(self, nodes, which=0):
        nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
        _check(lib().rt_debug_wide_write(self.h, which, _ptr(nodes), nodes.shape[0]))

    def instance_info(self, i):
        box = np.empty(6, np.float32)
        inv = np.empty(12, np.float32)
        _check(lib().rt_scene_instance_info(self.h, i, _ptr(box), _ptr(inv)))
        return box, inv

    def trace(self, origin_tmin, dir_tmax, flags=0, canonical=False):
        """Batch TraceRay from host arrays; returns dict of numpy arrays."""
        o = _f32(origin_tmin, (-1, 4))
        d = _f32(dir_tmax, (-1, 4))
        n = o.shape[0]
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32); inst = np.empty(n, np.uint32)
        cn = np.zeros(n, np.uint32); ct = np.zeros(n, np.uint32)
        _check(lib().rt_trace_batch(self.ctx.h, self.h, _ptr(o), _ptr(d), n, flags, 1 if canonical else 0, 0,
                                    _ptr(t), _ptr(u), _ptr(v), _ptr(prim), _ptr(inst),
                                    _ptr(cn) if canonical else None, _ptr(ct) if canonical else None))
        return dict(t=t, u=u, v=v, prim=prim, inst=inst, nodes=cn, tris=ct)

    def trace_device(self, o_ptr, d_ptr, n, flags=0, canonical=False, t=0, u=0, v=0, prim=0, inst=0, nodes=0, tris=0):
        """Batch TraceRay on device pointers (ints); asynchronous on the context stream."""
        cv = lambda x: C.c_void_p(x) if x else None
        _check(lib().rt_trace_batch(self.ctx.h, self.h, cv(o_ptr), cv(d_ptr), n, flags, 1 if canonical else 0, 1,
                                    cv(t), cv(u), cv(v), cv(prim), cv(inst), cv(nodes), cv(tris)))

    def trace_last_ms(self):
        ms = C.c_float()
        _check(lib().rt_trace_last_ms(self.ctx.h, C.byref(ms)))
        return ms.value


class Pipeline:
    """ProgressiveRaytracingPipeline (include/ProgressiveRaytracingPipeline.h:15-78) or, with
    kind=PIPELINE_REALTIME, RealtimeRaytracingPipeline (include/RealtimeRaytracingPipeline.h:15-74)."""

    def __init__(self, ctx, kind=PIPELINE_PROGRESSIVE):
        self.ctx = ctx
        self.kind = kind
        h = C.c_void_p()
        _check(lib().rt_pipeline_create(ctx.h, kind, C.byref(h)))
        self.h = h
        self.width = self.height = 0
        self.format = T.FORMAT_R32G32B32A32_FLOAT
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            lib().rt_pipeline_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def name(self):
        return lib().rt_pipeline_get_name(self.h).decode()

    def set_scene(self, scene):
        self._keep.append(scene)
        _check(lib().rt_pipeline_set_scene(self.h, scene.h))

    def add_material(self, m):
        m = np.ascontiguousarray(m, dtype=T.MATERIAL_PARAMS)
        _check(lib().rt_pipeline_add_material(self.h, _ptr(m)))

    def set_material(self, index, m):
        m = np.ascontiguousarray(m, dtype=T.MATERIAL_PARAMS)
        _check(lib().rt_pipeline_set_material(self.h, index, _ptr(m)))

    def set_environment_cube(self, faces):
        f = _f32(faces)
        assert f.ndim == 4 and f.shape[0] == 6 and f.shape[1] == f.shape[2] and f.shape[3] == 4
        _check(lib().rt_pipeline_set_environment_cube(self.h, _ptr(f), f.shape[1]))

    def set_skip_unlit_shadow_rays(self, on=True):
        """Shadow rays of lights with N.L == 0 (their visibility is multiplied by zero).  Off (the default, as in dxr_amd.h): every
        shadow ray the reference traces is traversed.  On: such rays are emitted and counted (rays_shadow_skipped) but not
        walked; the image is bit-identical either way."""
        _check(lib().rt_pipeline_set_skip_unlit_shadow_rays(self.h, 1 if on else 0))

    def set_environment_filter(self, seamless=True):
        """Cube-map filtering: seamless (cross-face taps, what D3D12 hardware does; default) or clamped to the face."""
        _check(lib().rt_pipeline_set_environment_filter(self.h, CUBE_SEAMLESS if seamless else CUBE_FACE_CLAMP))

    def set_environment_constant(self, rgb):
        c = _f32(rgb, 3)
        _check(lib().rt_pipeline_set_environment_constant(self.h, _ptr(c)))

    def load_environment_dds(self, path):
        _check(lib().rt_pipeline_load_environment_dds(self.h, os.fsencode(path)))

    def create_output(self, width, height, fmt=T.FORMAT_R32G32B32A32_FLOAT):
        _check(lib().rt_pipeline_create_output(self.h, fmt, width, height))
        self.width, self.height, self.format = width, height, fmt

    def bind_output(self, device_ptr, width, height):
        _check(lib().rt_pipeline_bind_output(self.h, C.c_void_p(device_ptr), width, height))
        self.width, self.height, self.format = width, height, T.FORMAT_R32G32B32A32_FLOAT

    def build_acceleration_structures(self):
        _check(lib().rt_pipeline_build_acceleration_structures(self.h))

    def set_depth_limits(self, max_radiance_depth=1, max_shadow_depth=2):
        _check(lib().rt_pipeline_set_depth_limits(self.h, max_radiance_depth, max_shadow_depth))

    def set_accumulation_mode(self, mode):
        _check(lib().rt_pipeline_set_accumulation_mode(self.h, mode))

    def set_accumulation_storage(self, fmt, rounding=0):
        """RT_FORMAT_R16G16B16A16_FLOAT: the running mean is rounded to fp16 every frame, as the reference's RGBA16F texture does (DXRExperimentsApp.cpp:28)"""
        _check(lib().rt_pipeline_set_accumulation_storage(self.h, fmt, rounding))

    def clear_output(self):
        _check(lib().rt_pipeline_clear_output(self.h))

    def update(self, constants):
        c = np.ascontiguousarray(constants)
        assert c.nbytes == 188
        _check(lib().rt_pipeline_update(self.h, _ptr(c)))

    def render(self, tile=None):
        if tile is None:
            _check(lib().rt_pipeline_render(self.h, self.width, self.height))
        else:
            _check(lib().rt_pipeline_render_tile(self.h, self.width, self.height, *tile))

    def render_batch(self, constants):
        """len(constants) frames = len x (update, render), bit for bit, through shared sets of launches (sample batches,
        BASELINE configs[2]).  constants: a sequence of 188-byte per-frame constant buffers (as ProgressiveHost.update returns)."""
        buf = np.ascontiguousarray(np.stack([np.frombuffer(np.asarray(c).tobytes(), np.uint8) for c in constants]))
        assert buf.shape[1] == 188
        _check(lib().rt_pipeline_render_batch(self.h, self.width, self.height, _ptr(buf), buf.shape[0]))

    def set_shadow_cache(self, cells_per_side):
        """Light buffer of occluders for shadow rays (same image, less time): -1 automatic, 0 off, else cells per side."""
        _check(lib().rt_pipeline_set_shadow_cache(self.h, int(cells_per_side)))

    def shadow_cache(self):
        """cells per side of the shadow cache the last frame ran with (0: none)"""
        n = C.c_int(0)
        _check(lib().rt_pipeline_get_shadow_cache(self.h, C.byref(n)))
        return n.value

    def free_sphere(self):
        """radius of the empty sphere around the point light of the last update() that its shadow rays stop at (0: not known yet)"""
        r = C.c_float(0.0)
        _check(lib().rt_pipeline_get_free_sphere(self.h, C.byref(r)))
        return r.value

    def reserve_batch(self, frames, rows=None):
        """Reserve the work memory of sets of `frames` frames now (the first set of that size then allocates nothing); rows: the
        packed rows of a rank's bands (render_bands_batch) instead of the whole height."""
        _check(lib().rt_pipeline_reserve_batch(self.h, self.width, self.height if rows is None else int(rows), int(frames)))

    def render_bands(self, band_rows, rank, world):
        """One frame over this rank's interleaved row bands (tile-partitioned multi-GPU runs)."""
        _check(lib().rt_pipeline_render_bands(self.h, self.width, self.height, band_rows, rank, world))

    def render_bands_batch(self, band_rows, rank, world, constants):
        """len(constants) frames over this rank's interleaved row bands through shared sets of launches: render_batch for a
        tile-partitioned run (bit for bit the same rows of the whole frames)."""
        buf = np.ascontiguousarray(np.stack([np.frombuffer(np.asarray(c).tobytes(), np.uint8) for c in constants]))
        assert buf.shape[1] == 188
        _check(lib().rt_pipeline_render_bands_batch(self.h, self.width, self.height, band_rows, rank, world, _ptr(buf), buf.shape[0]))

    def set_deferred(self, max_frames):
        """Sets of frames behind update() + render(): render() only records the frame; the recorded frames go through ONE set
        of launches when max_frames (<= 32) have gathered or when anything reads or changes what they produce.  0 / 1: off."""
        _check(lib().rt_pipeline_set_deferred(self.h, int(max_frames)))

    def deferred(self):
        """(max frames per set, frames recorded and not rendered yet)"""
        m, n = C.c_uint32(0), C.c_uint32(0)
        _check(lib().rt_pipeline_get_deferred(self.h, C.byref(m), C.byref(n)))
        return m.value, n.value

    def flush(self):
        _check(lib().rt_pipeline_flush(self.h))

    def set_queue_budget(self, nbytes):
        """Worst-case queue bytes a set of launches may reserve up front; above it the levels are sized by count (0: default)."""
        _check(lib().rt_pipeline_set_queue_budget(self.h, int(nbytes)))

    def queue_memory(self):
        """(bytes of ray / hit / shadow queues reserved, whether the last set sized its levels by count)"""
        b, c = C.c_size_t(0), C.c_uint32(0)
        _check(lib().rt_pipeline_get_queue_memory(self.h, C.byref(b), C.byref(c)))
        return b.value, bool(c.value)

    @property
    def num_outputs(self):
        n = C.c_int()
        _check(lib().rt_pipeline_get_num_outputs(self.h, C.byref(n)))
        return n.value

    def output_device_ptr(self, output=0):
        p = C.c_void_p()
        _check(lib().rt_pipeline_get_output_device_ptr(self.h, output, C.byref(p)))
        return p.value

    def write_output(self, image):
        """Inverse of read_output() for the fp32 accumulation image."""
        img = np.ascontiguousarray(image, np.float32)
        _check(lib().rt_pipeline_write_output(self.h, _ptr(img), img.nbytes))

    def save_checkpoint(self, path, host=None):
        _check(lib().rt_pipeline_save_checkpoint(self.h, host.h if host is not None else None, os.fsencode(path)))

    def load_checkpoint(self, path, host=None):
        _check(lib().rt_pipeline_load_checkpoint(self.h, host.h if host is not None else None, os.fsencode(path)))

    def read_output(self, output=0):
        if self.format == T.FORMAT_R16G16B16A16_FLOAT:
            out = np.empty((self.height, self.width, 4), np.float16)
        else:
            out = np.empty((self.height, self.width, 4), np.float32)
        _check(lib().rt_pipeline_read_output_n(self.h, output, _ptr(out), out.nbytes))
        return out

    def enable_timing(self, frames=1):
        """Record HIP events around every stage kernel, remembering the last `frames` frames (0 = off)."""
        _check(lib().rt_pipeline_enable_timing(self.h, int(frames)))

    def totals(self):
        s = Stats()
        _check(lib().rt_pipeline_get_totals(self.h, C.byref(s)))
        return s.as_dict()

    def reset_totals(self):
        _check(lib().rt_pipeline_reset_totals(self.h))

    def count_work(self):
        """Canonical-traversal work of the last frame per stage: {stage: dict(rays, nodes, tris)}."""
        w = (StageWork * len(STAGES))()
        _check(lib().rt_pipeline_count_work(self.h, w))
        return {n: dict(rays=int(w[i].rays), nodes=int(w[i].nodes), tris=int(w[i].tris)) for i, n in enumerate(STAGES)}

    def secondary_ray(self, index):
        o = np.zeros(4, np.float32); d = np.zeros(4, np.float32)
        _check(lib().rt_debug_read_secondary_ray(self.h, int(index), _ptr(o), _ptr(d)))
        return o, d

    def count_walk(self):
        """What the PRODUCTION traversal fetched for the last frame per stage:
        {stage: dict(rays, nodes_global, nodes_lds, tris, instance_entries, lines, longest_walk, longest_walk_ray,
        wave_node_steps, wave_leaf_phases, wave_tri_steps)}."""
        w = (StageWalk * len(STAGES))()
        _check(lib().rt_pipeline_count_walk(self.h, w))
        return {n: {f: int(getattr(w[i], f)) for f, _ in StageWalk._fields_} for i, n in enumerate(STAGES)}

    def stats(self):
        s = Stats()
        _check(lib().rt_pipeline_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def primary_hits(self, n):
        t = np.empty(n, np.float32); prim = np.empty(n, np.uint32); inst = np.empty(n, np.uint32)
        _check(lib().rt_pipeline_read_primary_hits(self.h, _ptr(t), _ptr(prim), _ptr(inst)))
        return t, prim, inst


def shard_frame_count(rank, world, n_frames):
    """|{f < n_frames : f mod world == rank}| (partition A of SURVEY 8(e)); host logic of the C ABI, no device."""
    n = C.c_uint32(0)
    _check(lib().rt_shard_frame_count(rank, world, n_frames, C.byref(n)))
    return n.value


def tile_bands(height, band_rows, rank, world):
    """Interleaved row bands of `rank` as (y0, y1) pairs (partition B); host logic of the C ABI, no device."""
    n = C.c_uint32(0)
    _check(lib().rt_tile_bands(height, band_rows, rank, world, None, None, 0, C.byref(n)))
    y0 = np.zeros(max(n.value, 1), np.uint32)
    y1 = np.zeros(max(n.value, 1), np.uint32)
    _check(lib().rt_tile_bands(height, band_rows, rank, world, _ptr(y0), _ptr(y1), n.value, C.byref(n)))
    return [(int(a), int(b)) for a, b in zip(y0[:n.value], y1[:n.value])]


def tile_gather_layout(width, height, band_rows, world):
    """(band slots per rank, floats per rank) of the all-gather that combines the bands."""
    slots, floats = C.c_uint32(0), C.c_size_t(0)
    _check(lib().rt_tile_gather_layout(width, height, band_rows, world, C.byref(slots), C.byref(floats)))
    return slots.value, floats.value


class Dist:
    """rt_dist: RCCL from the C ABI (one process per GPU).  uid = Dist.unique_id() on rank 0, handed to the other ranks."""

    @staticmethod
    def unique_id():
        buf = np.zeros(128, np.uint8)
        _check(lib().rt_dist_get_unique_id(_ptr(buf)))
        return buf.tobytes()

    def __init__(self, ctx, rank, world, uid):
        self.ctx = ctx
        h = C.c_void_p()
        buf = np.frombuffer(uid, np.uint8).copy()
        _check(lib().rt_dist_create(ctx.h, int(rank), int(world), _ptr(buf), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().rt_dist_destroy(self.h)
            self.h = None

    __del__ = close

    def all_reduce_sum(self, device_ptr, count):
        _check(lib().rt_dist_all_reduce_sum(self.h, C.c_void_p(device_ptr), int(count)))

    def gather_bands(self, device_ptr, width, height, band_rows):
        _check(lib().rt_dist_gather_bands(self.h, C.c_void_p(device_ptr), width, height, band_rows))


def dds_read_cube(path):
    """The product's DDS cube-map reader, no device needed: (6, size, size, 4) float32 faces of mip 0."""
    n = C.c_uint32(0)
    _check(lib().rt_dds_read_cube(os.fsencode(path), None, 0, C.byref(n)))
    faces = np.empty((6, n.value, n.value, 4), np.float32)
    _check(lib().rt_dds_read_cube(os.fsencode(path), _ptr(faces), faces.size, C.byref(n)))
    return faces


def obj_read(path):
    """The product's OBJ reader, no device needed: (verts[VERTEX], tris[n,3] uint32)."""
    from .rtypes import VERTEX
    nv, nt = C.c_uint32(0), C.c_uint32(0)
    _check(lib().rt_obj_read(os.fsencode(path), None, 0, None, 0, C.byref(nv), C.byref(nt)))
    v = np.zeros(nv.value, VERTEX)
    t = np.zeros((nt.value, 3), np.uint32)
    _check(lib().rt_obj_read(os.fsencode(path), _ptr(v), nv.value, _ptr(t), nt.value, C.byref(nv), C.byref(nt)))
    return v, t


def fbx_read(path):
    """The product's binary-FBX mesh reader, no device needed: (verts[VERTEX], tris[n,3] uint32)."""
    from .rtypes import VERTEX
    nv, nt = C.c_uint32(0), C.c_uint32(0)
    _check(lib().rt_fbx_read(os.fsencode(path), None, 0, None, 0, C.byref(nv), C.byref(nt)))
    v = np.zeros(nv.value, VERTEX)
    t = np.zeros((nt.value, 3), np.uint32)
    _check(lib().rt_fbx_read(os.fsencode(path), _ptr(v), nv.value, _ptr(t), nt.value, C.byref(nv), C.byref(nt)))
    return v, t


def camera_look(eye, at, up):
    eye, at, up = _f32(eye, 3), _f32(at, 3), _f32(up, 3)
    f = np.empty(3, np.float32); u = np.empty(3, np.float32)
    _check(lib().rt_camera_look(_ptr(eye), _ptr(at), _ptr(up), _ptr(f), _ptr(u)))
    return f, u


def camera_basis(forward, up, fov, aspect):
    forward, up = _f32(forward, 3), _f32(up, 3)
    U = np.empty(4, np.float32); V = np.empty(4, np.float32); W = np.empty(4, np.float32)
    _check(lib().rt_camera_basis(_ptr(forward), _ptr(up), fov, aspect, _ptr(U), _ptr(V), _ptr(W)))
    return U, V, W


def camera_array(eye, at, up, fov, aspect):
    """Pack the 11 floats rt_progressive_host_update takes."""
    return np.array([*eye, *at, *up, fov, aspect], np.float32)


class ProgressiveHost:
    """Host-side frame logic of ProgressiveRaytracingPipeline::update (.cpp:177-213)."""

    def __init__(self, rng_seed=1234):
        h = C.c_void_p()
        _check(lib().rt_progressive_host_create(rng_seed, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().rt_progressive_host_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def options(self):
        """Live numpy view (DEBUG_OPTIONS record) of mShaderDebugOptions."""
        p = C.c_void_p()
        _check(lib().rt_progressive_host_options(self.h, C.byref(p)))
        buf = (C.c_uint8 * T.DEBUG_OPTIONS.itemsize).from_address(p.value)
        return np.frombuffer(buf, dtype=T.DEBUG_OPTIONS, count=1)

    def set_flags(self, accumulation_enabled=True, animation_paused=True):
        _check(lib().rt_progressive_host_set_flags(self.h, int(accumulation_enabled), int(animation_paused)))

    def reset(self):
        _check(lib().rt_progressive_host_reset(self.h))

    def save_state(self):
        n = C.c_size_t()
        _check(lib().rt_progressive_host_save_state(self.h, None, 0, C.byref(n)))
        buf = (C.c_uint8 * n.value)()
        _check(lib().rt_progressive_host_save_state(self.h, buf, n.value, C.byref(n)))
        return bytes(buf)

    def load_state(self, blob):
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        _check(lib().rt_progressive_host_load_state(self.h, buf, len(blob)))

    def update(self, camera11, elapsed_time, elapsed_frames, width, height):
        cam = _f32(camera11, 11)
        out = np.zeros((), T.PER_FRAME_CONSTANTS)
        _check(lib().rt_progressive_host_update(self.h, _ptr(cam), elapsed_time, elapsed_frames, width, height, _ptr(out)))
        return out

    def update_realtime(self, camera11, elapsed_time, elapsed_frames, width, height):
        """RealtimeRaytracingPipeline::update (src/RealtimeRaytracingPipeline.cpp:168-199)."""
        cam = _f32(camera11, 11)
        out = np.zeros((), T.PER_FRAME_CONSTANTS)
        _check(lib().rt_realtime_host_update(self.h, _ptr(cam), elapsed_time, elapsed_frames, width, height, _ptr(out)))
        return out


def write_pfm(path, image):
    """fp32 RGB portable float map of an (H, W, 4) float32 image."""
    img = np.ascontiguousarray(image, np.float32)
    _check(lib().rt_image_write_pfm(os.fsencode(path), _ptr(img), img.shape[1], img.shape[0]))


def write_exr(path, image):
    """(H, W, 4) float32 -> lossless OpenEXR (FLOAT channels A B G R, no compression)."""
    img = np.ascontiguousarray(image, np.float32)
    assert img.ndim == 3 and img.shape[2] == 4
    _check(lib().rt_image_write_exr(os.fsencode(path), _ptr(img), img.shape[1], img.shape[0]))


def write_png(path, image, exposure=1.0, gamma=2.2, tonemap=True):
    """8-bit RGB PNG of an (H, W, 4) float32 image: exposure, optional Reinhard, gamma."""
    img = np.ascontiguousarray(image, np.float32)
    _check(lib().rt_image_write_png(os.fsencode(path), _ptr(img), img.shape[1], img.shape[0], exposure, gamma, int(tonemap)))


class Denoiser:
    """DenoiseCompositor (include/DenoiseCompositor.h:5-59)."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        _check(lib().rt_denoiser_create(ctx.h, C.byref(h)))
        self.h = h
        self.width = self.height = 0
        self.format = T.FORMAT_R32G32B32A32_FLOAT

    def close(self):
        if getattr(self, "h", None):
            lib().rt_denoiser_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def params(self):
        """Live numpy view (DENOISER_PARAMS record) of the constant buffer."""
        p = C.c_void_p()
        _check(lib().rt_denoiser_get_params(self.h, C.byref(p)))
        buf = (C.c_uint8 * DENOISER_PARAMS.itemsize).from_address(p.value)
        return np.frombuffer(buf, dtype=DENOISER_PARAMS, count=1)

    def create_output(self, width, height, fmt=T.FORMAT_R32G32B32A32_FLOAT):
        _check(lib().rt_denoiser_create_output(self.h, fmt, width, height))
        self.width, self.height, self.format = width, height, fmt

    def dispatch(self, direct_ptr, indirect_ptr):
        _check(lib().rt_denoiser_dispatch(self.h, C.c_void_p(direct_ptr), C.c_void_p(indirect_ptr), self.width, self.height))

    def _read(self, fn):
        dt = np.float16 if self.format == T.FORMAT_R16G16B16A16_FLOAT else np.float32
        out = np.empty((self.height, self.width, 4), dt)
        _check(fn(self.h, _ptr(out), out.nbytes))
        return out

    def read_output(self):
        return self._read(lib().rt_denoiser_read_output)

    def read_intermediate(self):
        return self._read(lib().rt_denoiser_read_intermediate)

    def output_device_ptr(self):
        p = C.c_void_p()
        _check(lib().rt_denoiser_get_output_device_ptr(self.h, C.byref(p)))
        return p.value

    def last_ms(self):
        ms = C.c_float()
        _check(lib().rt_denoiser_last_ms(self.h, C.byref(ms)))
        return ms.value
