"""ctypes mirror of include/srt.h (PODs only) and a numpy container for the flat scene.

The flat scene is the reference's ObjectManager state (Object.h:59-89 in the reference) written out as
arrays; see include/srt.h for the layout contract.  This module holds no compute.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

SRT_OK = 0
SRT_ERR_ARG, SRT_ERR_LAYOUT, SRT_ERR_DEVICE, SRT_ERR_NO_GPU, SRT_ERR_TEXTURE, SRT_ERR_LIMIT, SRT_ERR_OOM = 1, 2, 3, 4, 5, 6, 7
SRT_FLAG_SMOOTH_NORMALS = 1 << 0
SRT_FLAG_COUNT_WORK = 1 << 1
SRT_FLAG_NO_TIMING = 1 << 2
SRT_FLAG_FRAMES_IN_FLIGHT = 1 << 3      # hint: other frames are in flight on the device (srt.h)

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_objects", C.c_uint32), ("n_nodes", C.c_uint32), ("n_tris", C.c_uint32), ("n_textures", C.c_uint32),
        ("node_min", _f32p), ("node_max", _f32p),
        ("node_left", _i32p), ("node_right", _i32p), ("node_first", _i32p), ("node_count", _i32p),
        ("obj_root", _u32p),
        ("tri_points", _f32p), ("tri_obj", _i32p), ("tri_tex", _i32p), ("tri_texcoord", _f32p), ("tri_normals", _f32p),
        ("obj_color", _f32p), ("obj_material", _f32p),
        ("tex_rgb", _u8p), ("tex_off", _u64p), ("tex_w", _u32p), ("tex_h", _u32p),
    ]


class FrameGeometry(C.Structure):
    """srt_frame_geometry (include/srt.h, f1 device half): per-object pointer tables."""
    _fields_ = [
        ("n_objects", C.c_uint32),
        ("obj_n_tris", _u32p), ("obj_n_nodes", _u32p),
        ("obj_points", C.POINTER(_f32p)), ("obj_order", C.POINTER(_u32p)), ("obj_node_min", C.POINTER(_f32p)), ("obj_node_max", C.POINTER(_f32p)),
        ("obj_color", _f32p), ("obj_material", _f32p),
    ]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("block_rows", C.c_uint32), ("block_first", C.c_uint32), ("block_stride", C.c_uint32), ("block_cols", C.c_uint32),
        ("focal", C.c_float),
        ("n_lights", C.c_uint32), ("light_pos", _f32p), ("ray_matrix", _f32p),
        ("shadow_div", C.c_float), ("reinhard", C.c_float), ("gamma", C.c_float),
        ("background", C.c_uint8 * 4),
        ("spp", C.c_uint32), ("flags", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64), ("hit_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
        ("node_tests_primary", C.c_uint64), ("tri_tests_primary", C.c_uint64),
        ("node_tests_shadow", C.c_uint64), ("tri_tests_shadow", C.c_uint64),
        ("ms_primary", C.c_float), ("ms_shadow", C.c_float), ("ms_shade", C.c_float), ("ms_total", C.c_float),
        ("launches", C.c_uint32), ("rows", C.c_uint32),
    ]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["node_tests"] = d["node_tests_primary"] + d["node_tests_shadow"]
        d["tri_tests"] = d["tri_tests_primary"] + d["tri_tests_shadow"]
        return d


def _ptr(a, ty):
    return a.ctypes.data_as(ty) if a is not None else ty()


@dataclass
class FlatScene:
    """Host-side flat scene (numpy, C-contiguous).  Field names follow include/srt.h."""
    node_min: np.ndarray
    node_max: np.ndarray
    node_left: np.ndarray
    node_right: np.ndarray
    node_first: np.ndarray
    node_count: np.ndarray
    obj_root: np.ndarray
    tri_points: np.ndarray
    tri_obj: np.ndarray
    obj_color: np.ndarray
    obj_material: np.ndarray
    tri_tex: np.ndarray | None = None
    tri_texcoord: np.ndarray | None = None
    tri_normals: np.ndarray | None = None
    tex_rgb: np.ndarray | None = None
    tex_off: np.ndarray | None = None
    tex_w: np.ndarray | None = None
    tex_h: np.ndarray | None = None
    names: list = field(default_factory=list)

    _SPEC = {
        "node_min": np.float32, "node_max": np.float32,
        "node_left": np.int32, "node_right": np.int32, "node_first": np.int32, "node_count": np.int32,
        "obj_root": np.uint32, "tri_points": np.float32, "tri_obj": np.int32,
        "obj_color": np.float32, "obj_material": np.float32,
        "tri_tex": np.int32, "tri_texcoord": np.float32, "tri_normals": np.float32,
        "tex_rgb": np.uint8, "tex_off": np.uint64, "tex_w": np.uint32, "tex_h": np.uint32,
    }

    def __post_init__(self):
        for k, dt in self._SPEC.items():
            v = getattr(self, k)
            if v is not None:
                setattr(self, k, np.ascontiguousarray(v, dtype=dt))
        if self.tri_tex is None:
            self.tri_tex = np.full(self.n_tris, -1, np.int32)

    @property
    def n_objects(self):
        return int(self.obj_root.shape[0])

    @property
    def n_nodes(self):
        return int(self.node_left.shape[0])

    @property
    def n_tris(self):
        return int(self.tri_obj.shape[0])

    @property
    def n_textures(self):
        return 0 if self.tex_w is None else int(self.tex_w.shape[0])

    def desc(self) -> SceneDesc:
        """ctypes descriptor pointing into this object's arrays (keep `self` alive while in use)."""
        d = SceneDesc()
        d.n_objects, d.n_nodes, d.n_tris, d.n_textures = self.n_objects, self.n_nodes, self.n_tris, self.n_textures
        d.node_min, d.node_max = _ptr(self.node_min, _f32p), _ptr(self.node_max, _f32p)
        d.node_left, d.node_right = _ptr(self.node_left, _i32p), _ptr(self.node_right, _i32p)
        d.node_first, d.node_count = _ptr(self.node_first, _i32p), _ptr(self.node_count, _i32p)
        d.obj_root = _ptr(self.obj_root, _u32p)
        d.tri_points = _ptr(self.tri_points, _f32p)
        d.tri_obj, d.tri_tex = _ptr(self.tri_obj, _i32p), _ptr(self.tri_tex, _i32p)
        d.tri_texcoord, d.tri_normals = _ptr(self.tri_texcoord, _f32p), _ptr(self.tri_normals, _f32p)
        d.obj_color, d.obj_material = _ptr(self.obj_color, _f32p), _ptr(self.obj_material, _f32p)
        d.tex_rgb, d.tex_off = _ptr(self.tex_rgb, _u8p), _ptr(self.tex_off, _u64p)
        d.tex_w, d.tex_h = _ptr(self.tex_w, _u32p), _ptr(self.tex_h, _u32p)
        return d

    ARRAYS = tuple(_SPEC.keys())

    def to_npz_dict(self, prefix="scene_"):
        out = {prefix + k: getattr(self, k) for k in self.ARRAYS if getattr(self, k) is not None}
        out[prefix + "names"] = np.array(self.names, dtype="U")
        return out

    @classmethod
    def from_npz_dict(cls, z, prefix="scene_"):
        kw = {k: z[prefix + k] for k in cls.ARRAYS if (prefix + k) in z}
        names = [str(s) for s in z[prefix + "names"]] if (prefix + "names") in z else []
        return cls(names=names, **kw)


REFERENCE_BACKGROUND = (173, 216, 230)   # drawImage, simple_raytracer.cpp:476


def light_staircase(base, n):
    """softShadow's light table (simple_raytracer.cpp:363-383), accumulated in f32 like the reference.
    Pure data preparation for srt_params.light_pos (the C ABI's srt_light_staircase does the same)."""
    out = np.zeros((n, 3), np.float32)
    L = np.array(base[:3], np.float32)
    for i in range(n):
        out[i] = L
        L[i % 3] = np.float32(L[i % 3] + np.float32(3.0))
    return out


def make_params(width, height, lights, *, block_rows=None, block_first=0, block_stride=1, block_cols=0,
                focal=400.0, shadow_div=5.0, reinhard=0.5, gamma=1.1, background=REFERENCE_BACKGROUND,
                spp=1, flags=0, ray_matrix=None):
    """srt_params with the reference's literals; `lights` is an (n,3) f32 array kept alive on the
    returned object (attribute _lights)."""
    p = Params()
    p.width, p.height = int(width), int(height)
    p.block_rows = int(block_rows if block_rows else height)
    p.block_first, p.block_stride, p.block_cols = int(block_first), int(block_stride), int(block_cols)
    p.focal = focal
    lights = np.ascontiguousarray(np.asarray(lights, np.float32).reshape(-1, 3))
    p.n_lights = lights.shape[0]
    p.light_pos = _ptr(lights, _f32p)
    p._lights = lights
    p.shadow_div, p.reinhard, p.gamma = shadow_div, reinhard, gamma
    p.background[0], p.background[1], p.background[2], p.background[3] = background[0], background[1], background[2], 0
    p.spp, p.flags = spp, flags
    if ray_matrix is not None:      # camera mode (extension): 16 floats, column-major
        m = np.ascontiguousarray(np.asarray(ray_matrix, np.float32).reshape(16))
        p.ray_matrix = _ptr(m, _f32p)
        p._ray_matrix = m
    return p


def owned_pixels(width, height, block_rows, block_first, block_stride, block_cols=0):
    """Image pixel (flat index y * width + x) of every local output pixel of a call with these block params, as an int64 array
    [rows_local, cols_local]; -1 = padding.  Mirrors include/srt.h (srt_params.block_rows / block_cols)."""
    if not block_cols:
        ys = rows_owned(height, block_rows, block_first, block_stride)
        return ys[:, None] * width + np.arange(width, dtype=np.int64)[None, :]
    n_bx = (width + block_cols - 1) // block_cols
    wl = (n_bx + block_stride - 1) // block_stride * block_cols
    y = np.arange(height, dtype=np.int64)[:, None]
    xl = np.arange(wl, dtype=np.int64)[None, :]
    off = (block_first + block_stride - (y // block_rows) % block_stride) % block_stride
    x = ((xl // block_cols) * block_stride + off) * block_cols + xl % block_cols
    return np.where(x < width, y * width + x, -1)


def rows_owned(height, block_rows, block_first, block_stride):
    """Image rows (ascending, in local-row order) a call with these block params writes."""
    ys = []
    nblocks = (height + block_rows - 1) // block_rows
    for b in range(block_first, nblocks, block_stride):
        ys.extend(range(b * block_rows, min((b + 1) * block_rows, height)))
    return np.array(ys, np.int64)
