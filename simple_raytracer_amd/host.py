"""ctypes binding of libsrt_host.so: the host-side C++ mirror of the reference's scene interface
(ObjectManager / Transformation / sendRaysAndIntersectPointsColors; csrc/host/srt_host.h).

Scene construction, transforms, the hierarchy builder and the flattener run on the host CPU as they
do in the reference; only `render()` (the drop-in for sendRaysAndIntersectPointsColors) goes to the
GPU, through the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsrt_host.so")
_f32p, _i32p, _u8p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run python -m simple_raytracer_amd.build")
        L = C.CDLL(LIB_PATH)
        L.srth_last_error.restype = C.c_char_p
        L.srth_om_new.restype = C.c_void_p
        L.srth_om_free.argtypes = [C.c_void_p]
        for fn in ("srth_om_load_obj", "srth_om_build_bvh"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_char_p]
        L.srth_om_add_object.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, _f32p]
        L.srth_om_clone.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.srth_sort_keys_both_ways.argtypes = [_f32p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.srth_decode_image.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _u8p]
        L.srth_om_add_texture.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, _u8p]
        L.srth_om_add_textured_object.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, _f32p, _f32p, C.c_char_p]
        L.srth_om_add_multi_textured_object.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, _f32p, _f32p, _i32p, C.c_char_p]
        L.srth_om_set_color.argtypes = [C.c_void_p, C.c_char_p] + [C.c_float] * 3
        L.srth_om_set_props.argtypes = [C.c_void_p, C.c_char_p] + [C.c_float] * 3
        L.srth_om_transform.argtypes = [C.c_void_p, C.c_char_p, _f32p]
        L.srth_om_num_tris.argtypes = [C.c_void_p, C.c_char_p]
        L.srth_om_num_tris.restype = C.c_int64
        L.srth_om_get_points.argtypes = [C.c_void_p, C.c_char_p, _f32p]
        L.srth_om_get_tri_attrs.argtypes = [C.c_void_p, C.c_char_p, _f32p, _f32p, _i32p, _f32p]
        L.srth_om_hierarchy_nodes.argtypes = [C.c_void_p, C.c_char_p]
        L.srth_om_hierarchy_nodes.restype = C.c_int64
        L.srth_om_hierarchy.argtypes = [C.c_void_p, C.c_char_p, _f32p, C.POINTER(C.c_uint32), _f32p, _f32p]
        L.srth_renderer_fast_frames.argtypes = [C.c_void_p]
        L.srth_renderer_fast_frames.restype = C.c_uint64
        L.srth_renderer_set_fast_path.argtypes = [C.c_void_p, C.c_int]
        L.srth_flatten.argtypes = [C.c_void_p]
        L.srth_flatten.restype = C.c_void_p
        L.srth_flat_free.argtypes = [C.c_void_p]
        L.srth_flat_desc.argtypes = [C.c_void_p, C.POINTER(abi.SceneDesc)]
        L.srth_flat_names.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        L.srth_flat_names.restype = C.c_uint32
        L.srth_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, _f32p, C.c_int, C.c_int, _f32p]
        L.srth_render.restype = C.c_int64
        L.srth_renderer_new.argtypes = [C.c_int]
        L.srth_renderer_new.restype = C.c_void_p
        L.srth_renderer_free.argtypes = [C.c_void_p]
        L.srth_renderer_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _f32p, C.c_int, _f32p]
        L.srth_renderer_render.restype = C.c_int64
        L.srth_renderer_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _f32p, C.c_int]
        L.srth_renderer_collect.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, _f32p]
        L.srth_renderer_collect.restype = C.c_int64
        L.srth_renderer_render_from_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.srth_renderer_render_from_camera.restype = C.c_int64
        L.srth_write_bmp.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, _u8p]
        L.srth_radians.argtypes = [C.c_float]
        L.srth_radians.restype = C.c_float
        for name in ("srth_mat_rotx", "srth_mat_roty", "srth_mat_rotz"):
            getattr(L, name).argtypes = [C.c_float, _f32p]
        L.srth_mat_scale.argtypes = [C.c_float] * 3 + [_f32p]
        L.srth_mat_translate.argtypes = [C.c_float] * 3 + [_f32p]
        L.srth_mat_shear.argtypes = [C.c_float] * 6 + [_f32p]
        L.srth_mat_mirror.argtypes = [C.c_int] * 3 + [_f32p]
        L.srth_mat_view.argtypes = [_f32p, _f32p, _f32p]
        L.srth_mat_inverse.argtypes = [_f32p, _f32p]
        L.srth_mat_mul.argtypes = [_f32p, _f32p, _f32p]
        L.srth_mat_mul_vec4.argtypes = [_f32p, _f32p, _f32p]
        _lib = L
    return _lib


def _p(a, ty=_f32p):
    return a.ctypes.data_as(ty)


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def set_build_tasks(on):
    """createBoundingHierarchy cut into pool tasks (True, default) or on the calling thread alone (False: for several frames built concurrently)."""
    load().srth_set_build_tasks(int(bool(on)))


class HostError(RuntimeError):
    pass


def _ok(rc):
    if rc != 0:
        raise HostError(load().srth_last_error().decode())


class Transformation:
    """Transformation.h:10-20 of the reference (same factory names) + the glm ops main() applies.
    Matrices are 16 floats, column-major (glm::mat4 memory order)."""

    @staticmethod
    def _m(fn, *a):
        out = np.empty(16, np.float32); fn(*a, _p(out)); return out

    @staticmethod
    def radians(d): return float(load().srth_radians(d))
    @staticmethod
    def scaleObj(x, y, z): return Transformation._m(load().srth_mat_scale, x, y, z)
    @staticmethod
    def rotateObjX(a): return Transformation._m(load().srth_mat_rotx, a)
    @staticmethod
    def rotateObjY(a): return Transformation._m(load().srth_mat_roty, a)
    @staticmethod
    def rotateObjZ(a): return Transformation._m(load().srth_mat_rotz, a)
    @staticmethod
    def mirrorObj(x, y, z): return Transformation._m(load().srth_mat_mirror, int(x), int(y), int(z))
    @staticmethod
    def shearObj(*s): return Transformation._m(load().srth_mat_shear, *s)
    @staticmethod
    def changeObjPosition(x, y, z): return Transformation._m(load().srth_mat_translate, x, y, z)

    @staticmethod
    def createViewMatrix(pos, rot):
        pos, rot, out = _f(pos), _f(rot), np.empty(16, np.float32)
        load().srth_mat_view(_p(pos), _p(rot), _p(out)); return out

    @staticmethod
    def inverse(m):
        m, out = _f(m), np.empty(16, np.float32); load().srth_mat_inverse(_p(m), _p(out)); return out

    @staticmethod
    def mul(a, b):
        a, b, out = _f(a), _f(b), np.empty(16, np.float32); load().srth_mat_mul(_p(a), _p(b), _p(out)); return out

    @staticmethod
    def mul_vec4(a, v):
        a, v, out = _f(a), _f(v), np.empty(4, np.float32); load().srth_mat_mul_vec4(_p(a), _p(v), _p(out)); return out

    # aliases with the provider names tests/scenes.py uses for both this class and the reference's
    scale, rotx, roty, rotz, mirror, shear, translate, view = scaleObj, rotateObjX, rotateObjY, rotateObjZ, mirrorObj, shearObj, changeObjPosition, createViewMatrix


class ObjectManager:
    """Object.h:59-89 of the reference: string-keyed objects, loadObjFile / transformTriangles /
    createBoundingHierarchy / setColor, plus flatten() (the flat scene of include/srt.h) and render()
    (drop-in for sendRaysAndIntersectPointsColors, on the GPU)."""

    def __init__(self):
        self.L = load()
        self.om = C.c_void_p(self.L.srth_om_new())

    def __del__(self):
        try:
            if self.om:
                self.L.srth_om_free(self.om); self.om = None
        except Exception:
            pass

    def loadObjFile(self, name): _ok(self.L.srth_om_load_obj(self.om, name.encode()))

    def add_object(self, name, points):
        pts = _f(points).reshape(-1, 12)
        _ok(self.L.srth_om_add_object(self.om, name.encode(), pts.shape[0], _p(pts)))

    def add_textured_object(self, name, points, texcoord, texname, texture):
        pts, tc = _f(points).reshape(-1, 12), _f(texcoord).reshape(-1, 6)
        tex = np.ascontiguousarray(texture, np.uint8)
        _ok(self.L.srth_om_add_texture(self.om, texname.encode(), tex.shape[1], tex.shape[0], _p(tex, _u8p)))
        _ok(self.L.srth_om_add_textured_object(self.om, name.encode(), pts.shape[0], _p(pts), _p(tc), texname.encode()))

    def add_multi_textured_object(self, name, points, texcoord, tri_tex, tex_names, textures):
        """Object whose triangles use several textures (house.obj): tri_tex[i] indexes tex_names / textures, -1 = untextured."""
        pts, tc = _f(points).reshape(-1, 12), _f(texcoord).reshape(-1, 6)
        tt = np.ascontiguousarray(tri_tex, np.int32)
        for nm, tex in zip(tex_names, textures):
            tex = np.ascontiguousarray(tex, np.uint8)
            _ok(self.L.srth_om_add_texture(self.om, nm.encode(), tex.shape[1], tex.shape[0], _p(tex, _u8p)))
        _ok(self.L.srth_om_add_multi_textured_object(self.om, name.encode(), pts.shape[0], _p(pts), _p(tc), _p(tt, _i32p), "\n".join(tex_names).encode()))

    def clone(self, src, dst): _ok(self.L.srth_om_clone(self.om, src.encode(), dst.encode()))
    def setColor(self, name, rgb): _ok(self.L.srth_om_set_color(self.om, name.encode(), *[float(x) for x in rgb]))
    def set_props(self, name, p): _ok(self.L.srth_om_set_props(self.om, name.encode(), *[float(x) for x in p]))
    def transformTriangles(self, name, m): m = _f(m); _ok(self.L.srth_om_transform(self.om, name.encode(), _p(m)))
    def createBoundingHierarchy(self, name): _ok(self.L.srth_om_build_bvh(self.om, name.encode()))
    set_color, transform, build_bvh, load_obj = setColor, transformTriangles, createBoundingHierarchy, loadObjFile

    def num_tris(self, name):
        n = self.L.srth_om_num_tris(self.om, name.encode())
        if n < 0:
            raise KeyError(name)          # std::out_of_range from objTriangles.at(), Object.cpp:174
        return int(n)

    def points(self, name):
        out = np.empty((self.num_tris(name), 3, 4), np.float32)
        _ok(self.L.srth_om_get_points(self.om, name.encode(), _p(out))); return out

    def hierarchy(self, name):
        """(points n x 3 x 4 in source order at build time, order n, node_min m x 3, node_max m x 3): what srt_scene_update_frame takes."""
        n, m = self.num_tris(name), int(self.L.srth_om_hierarchy_nodes(self.om, name.encode()))
        if m < 0:
            raise KeyError(name)
        pts, order = np.empty((n, 3, 4), np.float32), np.empty(n, np.uint32)
        mn, mx = np.empty((m, 3), np.float32), np.empty((m, 3), np.float32)
        _ok(self.L.srth_om_hierarchy(self.om, name.encode(), _p(pts), order.ctypes.data_as(C.POINTER(C.c_uint32)), _p(mn), _p(mx)))
        return pts, order, mn, mx

    def tri_attrs(self, name):
        n = self.num_tris(name)
        tc, col, ht, nrm = np.empty((n, 6), np.float32), np.empty((n, 3), np.float32), np.empty(n, np.int32), np.empty((n, 9), np.float32)
        _ok(self.L.srth_om_get_tri_attrs(self.om, name.encode(), _p(tc), _p(col), _p(ht, _i32p), _p(nrm))); return tc, col, ht, nrm

    def flatten(self) -> abi.FlatScene:
        h = self.L.srth_flatten(self.om)
        if not h:
            raise HostError(self.L.srth_last_error().decode())
        try:
            d = abi.SceneDesc(); self.L.srth_flat_desc(h, C.byref(d))
            buf = C.create_string_buffer(1 << 16); self.L.srth_flat_names(h, buf, 1 << 16)
            names = [s for s in buf.value.decode().split("\n") if s]

            def arr(ptr, n, dt):
                return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
            no, nn, nt, nx = d.n_objects, d.n_nodes, d.n_tris, d.n_textures
            kw = {}
            if nx:
                tex_off = arr(d.tex_off, nx, np.uint64); tex_w = arr(d.tex_w, nx, np.uint32); tex_h = arr(d.tex_h, nx, np.uint32)
                total = int(max(int(o) + int(w) * int(h_) * 3 for o, w, h_ in zip(tex_off, tex_w, tex_h)))
                kw = dict(tex_rgb=arr(d.tex_rgb, total, np.uint8), tex_off=tex_off, tex_w=tex_w, tex_h=tex_h)
            return abi.FlatScene(
                node_min=arr(d.node_min, nn * 3, np.float32).reshape(nn, 3), node_max=arr(d.node_max, nn * 3, np.float32).reshape(nn, 3),
                node_left=arr(d.node_left, nn, np.int32), node_right=arr(d.node_right, nn, np.int32),
                node_first=arr(d.node_first, nn, np.int32), node_count=arr(d.node_count, nn, np.int32),
                obj_root=arr(d.obj_root, no, np.uint32),
                tri_points=arr(d.tri_points, nt * 12, np.float32).reshape(nt, 3, 4), tri_obj=arr(d.tri_obj, nt, np.int32),
                tri_tex=arr(d.tri_tex, nt, np.int32), tri_texcoord=arr(d.tri_texcoord, nt * 6, np.float32).reshape(nt, 6),
                tri_normals=arr(d.tri_normals, nt * 9, np.float32).reshape(nt, 9),
                obj_color=arr(d.obj_color, no * 3, np.float32).reshape(no, 3), obj_material=arr(d.obj_material, no * 3, np.float32).reshape(no, 3),
                names=names, **kw)
        finally:
            self.L.srth_flat_free(h)

    def render(self, W, H, light4, light_amount=1, device=0):
        """sendRaysAndIntersectPointsColors drop-in (GPU): dense H x W x 3 float image of the emitted
        (px,py,rgb) list, 0 where nothing was emitted; returns (image, n_emitted)."""
        light4 = _f(light4)
        rgb = np.empty((H, W, 3), np.float32)
        n = self.L.srth_render(self.om, W, H, _p(light4), light_amount, device, _p(rgb))
        if n < 0:
            raise HostError(self.L.srth_last_error().decode())
        return rgb, int(n)


class Renderer:
    """srt_host::Renderer: a device scene kept across frames (in-place scene updates, pinned buffers, asynchronous frames, camera mode)."""

    def __init__(self, device=0):
        self.L = load()
        self.h = C.c_void_p(self.L.srth_renderer_new(device))
        if not self.h:
            raise HostError("srth_renderer_new failed")

    def __del__(self):
        try:
            if self.h:
                self.L.srth_renderer_free(self.h); self.h = None
        except Exception:
            pass

    def _out(self, W, H, want):
        return np.empty((H, W, 3), np.float32) if want else None

    def _ret(self, n, rgb):
        if n < 0:
            raise HostError(self.L.srth_last_error().decode())
        return (rgb, int(n)) if rgb is not None else int(n)

    def render(self, om, W, H, light4, light_amount=1, image=True):
        light4 = _f(light4); rgb = self._out(W, H, image)
        return self._ret(self.L.srth_renderer_render(self.h, om.om, W, H, _p(light4), light_amount, _p(rgb) if image else None), rgb)

    @property
    def fast_frames(self):
        """Uploads that took the device half of the rebuild (srt_scene_update_frame) instead of flatten + srt_scene_update."""
        return int(self.L.srth_renderer_fast_frames(self.h))

    def set_fast_path(self, on):
        self.L.srth_renderer_set_fast_path(self.h, int(bool(on)))

    def submit(self, om, W, H, light4, light_amount=1):
        light4 = _f(light4)
        _ok(self.L.srth_renderer_submit(self.h, om.om, W, H, _p(light4), light_amount))

    def collect(self, W, H, image=True):
        rgb = self._out(W, H, image)
        return self._ret(self.L.srth_renderer_collect(self.h, W, H, _p(rgb) if image else None), rgb)

    def render_from_camera(self, om, W, H, light4_world, view16, light_amount=1, scene_changed=False, image=True):
        light4_world, view16 = _f(light4_world), _f(view16); rgb = self._out(W, H, image)
        return self._ret(self.L.srth_renderer_render_from_camera(self.h, om.om, W, H, _p(light4_world), _p(view16), light_amount, int(scene_changed),
                                                                 _p(rgb) if image else None), rgb)


class MultiRenderer:
    """srt_host::MultiRenderer: the framebuffer split over several devices from ONE host process (scene replicated, scanline blocks dealt
    block-cyclically, rows put together on the host)."""

    def __init__(self, devices, block_rows=8):
        self.L = load()
        self.L.srth_multi_new.restype = C.c_void_p
        self.L.srth_multi_new.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.c_uint32]
        self.L.srth_multi_free.argtypes = [C.c_void_p]
        self.L.srth_multi_render.restype = C.c_int64
        self.L.srth_multi_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]
        dv = (C.c_int * len(devices))(*devices)
        self.h = self.L.srth_multi_new(dv, len(devices), block_rows)
        if not self.h:
            raise HostError(self.L.srth_last_error().decode())

    def __del__(self):
        try:
            if self.h:
                self.L.srth_multi_free(self.h); self.h = None
        except Exception:
            pass

    def render(self, om, W, H, light4, light_amount=1):
        light4 = _f(light4); rgb = np.empty((H, W, 3), np.float32)
        n = self.L.srth_multi_render(self.h, om.om, W, H, _p(light4), light_amount, _p(rgb))
        if n < 0:
            raise HostError(self.L.srth_last_error().decode())
        return rgb, int(n)


def sort_keys_both_ways(keys):
    """Test hook: (permutation by the hierarchy builder's parallel sort, permutation by std::sort) of float keys."""
    keys = _f(keys)
    a = np.empty(keys.size, np.uint32); b = np.empty(keys.size, np.uint32)
    _ok(load().srth_sort_keys_both_ways(_p(keys), keys.size, _p(a, C.POINTER(C.c_uint32)), _p(b, C.POINTER(C.c_uint32))))
    return a, b


def decode_image(path):
    """The texture loader's decode of an image file (PNG / JPEG / PPM / BMP): H x W x 3 uint8, or None."""
    L = load()
    w, h = C.c_int32(), C.c_int32()
    if L.srth_decode_image(path.encode(), C.byref(w), C.byref(h), None):
        return None
    rgb = np.empty((h.value, w.value, 3), np.uint8)
    L.srth_decode_image(path.encode(), C.byref(w), C.byref(h), _p(rgb, _u8p))
    return rgb


def write_bmp(path, rgb8):
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    _ok(load().srth_write_bmp(path.encode(), rgb8.shape[1], rgb8.shape[0], _p(rgb8, _u8p)))


def build_flat_scene(recipe, meshes) -> abi.FlatScene:
    """Replay a tests/scenes.Recipe on the host ObjectManager and flatten it."""
    om = ObjectManager()
    recipe.replay(om, meshes)
    return om.flatten()
