"""ctypes binding of the C ABI in include/srt.h (libsrt_hip.so).

The product path has NO CPU fallback: if the HIP library is missing or no GPU is visible, this
raises.  (The CPU restatement under oracle/ is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsrt_hip.so")

# every symbol include/srt.h declares
ABI_SYMBOLS = ("srt_params_default", "srt_light_staircase", "srt_rows_owned", "srt_cols_owned", "srt_scene_create", "srt_scene_destroy", "srt_scene_update", "srt_scene_share",
               "srt_render_device", "srt_render_device_batch", "srt_render", "srt_render_async", "srt_host_alloc", "srt_host_free", "srt_sync", "srt_scene_device_bytes", "srt_strerror",
               "srt_last_hip_error", "srt_abi_version", "srt_kat_ray_aabb", "srt_kat_ray_triangle", "srt_kat_phong", "srt_kat_tonemap", "srt_kat_interp_normal", "srt_kat_pow",
               "srt_debug_fail_host_allocs", "srt_debug_valu_rate", "srt_debug_scene_records", "srt_scene_set_source", "srt_scene_update_frame",
               "srt_scene_pipeline", "srt_scene_overlap_estimate")

_f32p, _i32p, _u8p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
_lib = None


class SrtError(RuntimeError):
    def __init__(self, code, where):
        L = load()
        msg = L.srt_strerror(code).decode()
        super().__init__(f"{where}: {msg} (code {code}, hip error {L.srt_last_hip_error()})")
        self.code = code


def load():
    """Load libsrt_hip.so; fail loudly if it has not been built (python -m simple_raytracer_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: the HIP extension must be built "
                               "(python -m simple_raytracer_amd.build); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.srt_params_default.argtypes = [C.POINTER(abi.Params), C.c_uint32, C.c_uint32]
        L.srt_params_default.restype = None
        L.srt_light_staircase.argtypes = [_f32p, C.c_uint32, _f32p]
        L.srt_light_staircase.restype = None
        L.srt_rows_owned.argtypes = [C.POINTER(abi.Params)]
        L.srt_rows_owned.restype = C.c_uint32
        L.srt_cols_owned.argtypes = [C.POINTER(abi.Params)]
        L.srt_cols_owned.restype = C.c_uint32
        L.srt_scene_create.argtypes = [C.c_int, C.POINTER(abi.SceneDesc), C.POINTER(C.c_void_p)]
        L.srt_scene_create.restype = C.c_int
        L.srt_scene_update.argtypes = [C.c_void_p, C.POINTER(abi.SceneDesc), C.c_void_p]
        L.srt_scene_update.restype = C.c_int
        L.srt_scene_share.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.srt_scene_share.restype = C.c_int
        L.srt_scene_destroy.argtypes = [C.c_void_p]
        L.srt_scene_destroy.restype = C.c_int
        L.srt_render_device.argtypes = [C.c_void_p, C.POINTER(abi.Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.srt_render_device.restype = C.c_int
        L.srt_render_device_batch.argtypes = [C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(abi.Params), C.c_void_p] + [C.POINTER(C.c_void_p)] * 4
        L.srt_render_device_batch.restype = C.c_int
        L.srt_render.argtypes = [C.c_void_p, C.POINTER(abi.Params), _i32p, _f32p, _f32p, _u8p, C.POINTER(abi.Stats)]
        L.srt_render.restype = C.c_int
        L.srt_render_async.argtypes = [C.c_void_p, C.POINTER(abi.Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.srt_render_async.restype = C.c_int
        L.srt_host_alloc.argtypes = [C.c_size_t]
        L.srt_host_alloc.restype = C.c_void_p
        L.srt_host_free.argtypes = [C.c_void_p]
        L.srt_host_free.restype = None
        L.srt_sync.argtypes = [C.c_void_p, C.POINTER(abi.Stats)]
        L.srt_sync.restype = C.c_int
        L.srt_scene_device_bytes.argtypes = [C.c_void_p]
        L.srt_scene_device_bytes.restype = C.c_uint64
        L.srt_scene_pipeline.argtypes = [C.c_void_p]
        L.srt_scene_pipeline.restype = C.c_char_p
        L.srt_scene_overlap_estimate.argtypes = [C.c_void_p]
        L.srt_scene_overlap_estimate.restype = C.c_double
        L.srt_strerror.argtypes = [C.c_int]
        L.srt_strerror.restype = C.c_char_p
        L.srt_last_hip_error.restype = C.c_int
        L.srt_abi_version.restype = C.c_uint32
        L.srt_debug_fail_host_allocs.argtypes = [C.c_int]
        L.srt_debug_fail_host_allocs.restype = None
        L.srt_kat_ray_aabb.argtypes = [C.c_int, C.c_uint32, _f32p, _f32p, _u8p, _u8p, _u8p, _u8p]
        L.srt_kat_ray_triangle.argtypes = [C.c_int, C.c_uint32, _f32p, _f32p, _f32p]
        L.srt_kat_phong.argtypes = [C.c_int, C.c_uint32, _f32p, _f32p]
        L.srt_kat_interp_normal.argtypes = [C.c_int, C.c_uint32, _f32p, _f32p]
        L.srt_kat_pow.argtypes = [C.c_int, C.c_uint32, _f32p, _f32p, _f32p, _f32p]
        L.srt_kat_tonemap.argtypes = [C.c_int, C.c_uint32, _f32p, C.c_float, C.c_float, _f32p, _i32p]
        _lib = L
    return _lib


def _check(rc, where):
    if rc != abi.SRT_OK:
        raise SrtError(rc, where)


class DeviceScene:
    """A flat scene resident on one HIP device (opaque srt_scene handle)."""

    def __init__(self, flat: abi.FlatScene, device: int = 0):
        self.L = load()
        self.flat = flat
        self.device = device
        h = C.c_void_p()
        d = flat.desc()
        _check(self.L.srt_scene_create(device, C.byref(d), C.byref(h)), "srt_scene_create")
        self.h = h

    def share(self):
        """srt_scene_share: another handle (own workspace and counters) on this scene's device records."""
        o = object.__new__(DeviceScene)
        o.L, o.flat, o.device = self.L, self.flat, self.device
        h = C.c_void_p()
        _check(self.L.srt_scene_share(self.h, C.byref(h)), "srt_scene_share")
        o.h = h
        return o

    def update(self, flat: abi.FlatScene, stream=0):
        """srt_scene_update: new geometry with the same counts into the existing device allocations (asynchronous on `stream`)."""
        d = flat.desc()
        _check(self.L.srt_scene_update(self.h, C.byref(d), C.c_void_p(stream)), "srt_scene_update")
        self.flat = flat

    def set_source(self, tri_texcoord=None, tri_normals=None, tri_tex=None):
        """srt_scene_set_source: per-triangle attributes in SOURCE order (objects concatenated), for update_frame."""
        a = [None if x is None else np.ascontiguousarray(x, ty) for x, ty in ((tri_texcoord, np.float32), (tri_normals, np.float32), (tri_tex, np.int32))]
        self.L.srt_scene_set_source.argtypes = [C.c_void_p, _f32p, _f32p, _i32p]
        _check(self.L.srt_scene_set_source(self.h, a[0].ctypes.data_as(_f32p) if a[0] is not None else None, a[1].ctypes.data_as(_f32p) if a[1] is not None else None,
                                           a[2].ctypes.data_as(_i32p) if a[2] is not None else None), "srt_scene_set_source")

    def update_frame(self, points, order, node_min, node_max, obj_color=None, obj_material=None, stream=0):
        """srt_scene_update_frame: per object the transformed points in source order (n x 3 x 4), the build's permutation (n), the node
        boxes in DFS pre-order (m x 3 each); the records are derived on the device."""
        n = len(points)
        pts = [np.ascontiguousarray(x, np.float32).reshape(-1) for x in points]
        ords = [np.ascontiguousarray(x, np.uint32) for x in order]
        mn = [np.ascontiguousarray(x, np.float32).reshape(-1) for x in node_min]
        mx = [np.ascontiguousarray(x, np.float32).reshape(-1) for x in node_max]
        g = abi.FrameGeometry()
        g.n_objects = n
        nt = (C.c_uint32 * n)(*[o.shape[0] for o in ords]); nn = (C.c_uint32 * n)(*[m.shape[0] // 3 for m in mn])
        pp = (_f32p * n)(*[p.ctypes.data_as(_f32p) for p in pts]); po = (C.POINTER(C.c_uint32) * n)(*[o.ctypes.data_as(C.POINTER(C.c_uint32)) for o in ords])
        pmn = (_f32p * n)(*[m.ctypes.data_as(_f32p) for m in mn]); pmx = (_f32p * n)(*[m.ctypes.data_as(_f32p) for m in mx])
        g.obj_n_tris, g.obj_n_nodes, g.obj_points, g.obj_order, g.obj_node_min, g.obj_node_max = nt, nn, pp, po, pmn, pmx
        col = None if obj_color is None else np.ascontiguousarray(obj_color, np.float32)
        mat = None if obj_material is None else np.ascontiguousarray(obj_material, np.float32)
        g.obj_color = col.ctypes.data_as(_f32p) if col is not None else None
        g.obj_material = mat.ctypes.data_as(_f32p) if mat is not None else None
        self.L.srt_scene_update_frame.argtypes = [C.c_void_p, C.POINTER(abi.FrameGeometry), C.c_void_p]
        _check(self.L.srt_scene_update_frame(self.h, C.byref(g), C.c_void_p(stream)), "srt_scene_update_frame")

    def records(self):
        """srt_debug_scene_records: the device records as raw numpy arrays (dict)."""
        nN, nT, nO = self.flat.n_nodes, self.flat.n_tris, self.flat.n_objects
        out = {"nodes": np.zeros((nN, 8), np.uint32), "tris": np.zeros((nT, 12), np.uint32), "tris_o": np.zeros((nT, 12), np.uint32),
               "wide": np.zeros(((nN - nO) // 2, 16), np.uint32), "root_nodes": np.zeros((nO, 8), np.uint32),
               "tri_texcoord": np.zeros((nT, 6), np.float32), "tri_normals": np.zeros((nT, 9), np.float32), "tri_tex": np.full(nT, -7, np.int32)}
        self.L.srt_debug_scene_records.argtypes = [C.c_void_p] * 6 + [_f32p, _f32p, _i32p]
        v = lambda k: out[k].ctypes.data_as(C.c_void_p)
        _check(self.L.srt_debug_scene_records(self.h, v("nodes"), v("tris"), v("tris_o"), v("wide"), v("root_nodes"), out["tri_texcoord"].ctypes.data_as(_f32p),
                                              out["tri_normals"].ctypes.data_as(_f32p), out["tri_tex"].ctypes.data_as(_i32p)), "srt_debug_scene_records")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.L.srt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self):
        return int(self.L.srt_scene_device_bytes(self.h))

    @property
    def pipeline(self):
        """Kernels of the last render, in launch order."""
        return self.L.srt_scene_pipeline(self.h).decode()

    @property
    def overlap_estimate(self):
        return float(self.L.srt_scene_overlap_estimate(self.h))

    def rows(self, params):
        return int(self.L.srt_rows_owned(C.byref(params)))

    def cols(self, params):
        """Width of the rows a call with these params writes (params.width unless the frame is dealt in tiles)."""
        return int(self.L.srt_cols_owned(C.byref(params)))

    def render(self, params: abi.Params, want=("hit_id", "t", "rgb_linear", "rgb8")):
        """srt_render: host buffers out.  Returns dict of numpy arrays + 'stats'."""
        rows, W = self.rows(params), self.cols(params)
        out = {}
        if "hit_id" in want: out["hit_id"] = np.empty((rows, W), np.int32)
        if "t" in want: out["t"] = np.empty((rows, W), np.float32)
        if "rgb_linear" in want: out["rgb_linear"] = np.empty((rows, W, 3), np.float32)
        if "rgb8" in want: out["rgb8"] = np.empty((rows, W, 3), np.uint8)
        st = abi.Stats()
        g = lambda k, ty: out[k].ctypes.data_as(ty) if k in out else ty()
        _check(self.L.srt_render(self.h, C.byref(params), g("hit_id", _i32p), g("t", _f32p), g("rgb_linear", _f32p), g("rgb8", _u8p),
                                 C.byref(st)), "srt_render")
        out["stats"] = st.as_dict()
        return out

    def render_device(self, params: abi.Params, stream=0, hit_id=0, t=0, rgb_linear=0, rgb8=0):
        """srt_render_device: raw device pointers (ints, e.g. torch.Tensor.data_ptr()) in, async on `stream`."""
        _check(self.L.srt_render_device(self.h, C.byref(params), C.c_void_p(stream), C.c_void_p(hit_id), C.c_void_p(t),
                                        C.c_void_p(rgb_linear), C.c_void_p(rgb8)), "srt_render_device")

    def sync(self):
        st = abi.Stats()
        _check(self.L.srt_sync(self.h, C.byref(st)), "srt_sync")
        return st.as_dict()


class FrameBatch:
    """The argument tables of one srt_render_device_batch call, built once (a step of a bench or an orbit re-issues the same call):
    `scenes` are distinct DeviceScenes, `params` one abi.Params per frame, the outputs lists of raw device pointers (ints) or None."""

    def __init__(self, scenes, params, hit_id=None, t=None, rgb_linear=None, rgb8=None):
        n = len(scenes)
        assert len(params) == n
        self.L, self.n, self.scenes = load(), n, list(scenes)
        self.h = (C.c_void_p * n)(*[s.h for s in scenes])
        self.p = (abi.Params * n)()
        for i, q in enumerate(params):
            C.memmove(C.byref(self.p[i]), C.byref(q), C.sizeof(abi.Params))
        self._keep = list(params)             # the light arrays the params point to
        def table(v):
            if v is None:
                return None
            assert len(v) == n
            return (C.c_void_p * n)(*[C.c_void_p(int(x) if x else 0) for x in v])
        self.out = [table(v) for v in (hit_id, t, rgb_linear, rgb8)]

    def render(self, stream=0):
        _check(self.L.srt_render_device_batch(self.n, self.h, self.p, C.c_void_p(stream), *self.out), "srt_render_device_batch")


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def kat_ray_aabb(ray_od, box, device=0):
    """Device slab test on vectors: returns (literal, branch-free, filtered, ambiguous) uint8 arrays."""
    L = load(); ray_od, box = _f(ray_od), _f(box); n = ray_od.shape[0]
    outs = [np.empty(n, np.uint8) for _ in range(4)]
    _check(L.srt_kat_ray_aabb(device, n, ray_od.ctypes.data_as(_f32p), box.ctypes.data_as(_f32p), *[o.ctypes.data_as(_u8p) for o in outs]), "srt_kat_ray_aabb")
    return outs


def kat_ray_triangle(ray_od, tri_points, device=0):
    L = load(); ray_od, tri_points = _f(ray_od), _f(tri_points); n = ray_od.shape[0]; t = np.empty(n, np.float32)
    _check(L.srt_kat_ray_triangle(device, n, ray_od.ctypes.data_as(_f32p), tri_points.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p)), "srt_kat_ray_triangle")
    return t


def kat_phong(in28, device=0):
    L = load(); in28 = _f(in28); n = in28.shape[0]; rgb = np.empty((n, 3), np.float32)
    _check(L.srt_kat_phong(device, n, in28.ctypes.data_as(_f32p), rgb.ctypes.data_as(_f32p)), "srt_kat_phong")
    return rgb


def kat_tonemap(lin, reinhard=0.5, gamma=1.1, device=0):
    L = load(); lin = _f(lin).reshape(-1, 3); n = lin.shape[0]
    tone = np.empty((n, 3), np.float32); q = np.empty((n, 3), np.int32)
    _check(L.srt_kat_tonemap(device, n, lin.ctypes.data_as(_f32p), reinhard, gamma, tone.ctypes.data_as(_f32p), q.ctypes.data_as(_i32p)), "srt_kat_tonemap")
    return tone, q


def kat_interp_normal(in12, device=0):
    L = load(); in12 = _f(in12); n = in12.shape[0]; out = np.empty((n, 3), np.float32)
    _check(L.srt_kat_interp_normal(device, n, in12.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p)), "srt_kat_interp_normal")
    return out


def valu_rate(iters=2000, device=0):
    """(wave-instructions per SIMD-cycle, shader clock in GHz, the same rate over the whole launch span, waves per SIMD): srt_debug_valu_rate."""
    L = load()
    out = (C.c_double * 4)()
    L.srt_debug_valu_rate.argtypes = [C.c_int, C.c_uint32, C.POINTER(C.c_double)]
    _check(L.srt_debug_valu_rate(device, iters, out), "srt_debug_valu_rate")
    return float(out[0]), float(out[1]), float(out[2]), float(out[3])


def kat_pow(x, y, device=0):
    """The device powf on vectors: (shipped form, (float)pow(double)(x, y))."""
    L = load(); x, y = _f(x), _f(y); n = x.shape[0]; a = np.empty(n, np.float32); b = np.empty(n, np.float32)
    _check(L.srt_kat_pow(device, n, x.ctypes.data_as(_f32p), y.ctypes.data_as(_f32p), a.ctypes.data_as(_f32p), b.ctypes.data_as(_f32p)), "srt_kat_pow")
    return a, b
