"""Scene scripts shared by the golden generator, the parity tests and bench.py.

Each script restates one of the reference's edit-and-recompile scene blocks in main()
(/root/reference/simple_raytracer.cpp:553-769) as DATA: a list of ops over a scene-builder interface

    load(name, mesh_key) | clone(src, dst) | color(name, rgb) | props(name, (ka, ks, shin))
    transform(name, mat16_column_major) | bvh(name)

so that the same recipe can be replayed on the compiled reference (oracle/pyoracle.RefScene, golden
generation) and on the product's host ObjectManager mirror (simple_raytracer_amd.host).  Matrices come
from a `M` provider with the reference's Transformation.h factory names; recipes store the concrete
matrices so that replay does not depend on the provider.
"""
from __future__ import annotations

import ctypes
import json

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
_libm.cosf.restype = _libm.sinf.restype = ctypes.c_float
_libm.cosf.argtypes = _libm.sinf.argtypes = [ctypes.c_float]

LIGHT_DEFAULT = (500.0, -300.0, -200.0, 1.0)   # simple_raytracer.cpp:776


class Recipe:
    def __init__(self):
        self.ops = []
        self.light = None          # 3 floats, already in view space
        self.meshes = set()

    def load(self, name, mesh_key): self.ops.append(["load", name, mesh_key]); self.meshes.add(mesh_key)
    def clone(self, src, dst): self.ops.append(["clone", src, dst])
    def color(self, name, rgb): self.ops.append(["color", name, [float(x) for x in rgb]])
    def props(self, name, p): self.ops.append(["props", name, [float(x) for x in p]])
    def transform(self, name, m): self.ops.append(["transform", name, [float(np.float32(x)) for x in np.asarray(m).reshape(16)]])
    def bvh(self, name): self.ops.append(["bvh", name])

    def to_json(self):
        return json.dumps({"ops": self.ops, "light": [float(np.float32(x)) for x in self.light]})

    @classmethod
    def from_json(cls, s):
        d = json.loads(s); r = cls(); r.ops = d["ops"]; r.light = d["light"]
        r.meshes = {op[2] for op in r.ops if op[0] == "load"}
        return r

    def replay(self, builder, meshes):
        """builder: add_object(name, points[n,3,4]) / clone / set_color / set_props / transform / build_bvh."""
        for op in self.ops:
            apply_op(builder, op, meshes)


def apply_op(builder, op, meshes):
    k = op[0]
    if k == "load":
        m = meshes[op[2]]
        if isinstance(m, dict) and "tri_tex" in m:
            builder.add_multi_textured_object(op[1], m["points"], m["texcoord"], m["tri_tex"], m["texture_names"], m["textures"])
        elif isinstance(m, dict): builder.add_textured_object(op[1], m["points"], m["texcoord"], m["texture_name"], m["texture"])
        else: builder.add_object(op[1], m)
    elif k == "clone": builder.clone(op[1], op[2])
    elif k == "color": builder.set_color(op[1], op[2])
    elif k == "props": builder.set_props(op[1], op[2])
    elif k == "transform": builder.transform(op[1], np.array(op[2], np.float32))
    elif k == "bvh": builder.build_bvh(op[1])
    else: raise ValueError(k)


def _orbit_view(M, radius, angle_deg, height, pitch_deg):
    """Camera of main(): simple_raytracer.cpp:546-551 / :707-711."""
    rad = np.float32(M.radians(angle_deg))
    cx = np.float32(radius) * np.float32(_libm.cosf(float(rad)))     # std::cos(float) == cosf
    cz = np.float32(radius) * np.float32(_libm.sinf(float(rad)))
    view = M.view([cx, height, cz], [M.radians(pitch_deg), M.radians(angle_deg + 90.0), M.radians(0.0)])
    return M.inverse(view)


def orbit_view_matrix(M, radius, angle_deg, height, pitch_deg):
    """The viewMatrix of main() itself (simple_raytracer.cpp:546-551), whose INVERSE the reference applies to every triangle: what
    camera mode (srt_params.ray_matrix, an extension) takes as the camera-space -> scene-space ray matrix."""
    rad = np.float32(M.radians(angle_deg))
    cx = np.float32(radius) * np.float32(_libm.cosf(float(rad)))
    cz = np.float32(radius) * np.float32(_libm.sinf(float(rad)))
    return M.view([cx, height, cz], [M.radians(pitch_deg), M.radians(angle_deg + 90.0), M.radians(0.0)])


def in_world_space(recipe, inv):
    """The same scene script without its last step per object -- the transform into camera space (`inv`, simple_raytracer.cpp:558
    etc.): the scene where it stands, for camera mode.  The light stays at its world position."""
    inv16 = [float(np.float32(x)) for x in np.asarray(inv).reshape(16)]
    r = Recipe()
    r.ops = [op for op in recipe.ops if not (op[0] == "transform" and op[2] == inv16)]
    assert len(r.ops) < len(recipe.ops)
    r.meshes = set(recipe.meshes)
    r.light = LIGHT_DEFAULT[:3]
    return r


def one_cube(M, angle=0.0):
    """'One Sample Cube for Testing', simple_raytracer.cpp:703-722."""
    r = Recipe(); inv = _orbit_view(M, 100.0, angle, 0.0, 0.0)
    r.load("cube.obj", "cube")
    r.transform("cube.obj", M.scale(20.0, 20.0, 20.0))
    r.transform("cube.obj", M.roty(M.radians(25.0)))
    r.transform("cube.obj", inv)
    r.bvh("cube.obj")
    r.light = M.mul_vec4(inv, LIGHT_DEFAULT)[:3]
    return r


def sphere(M):
    """First object of the 6-sphere scene, simple_raytracer.cpp:640-641 (BASELINE config 1)."""
    r = Recipe()
    r.load("sphere.obj", "sphere")
    r.transform("sphere.obj", M.translate(0.0, 6.0, 30.0))
    r.bvh("sphere.obj")
    r.light = LIGHT_DEFAULT[:3]            # ':625 To use this comment out: lightPos = inverse(view)*lightPos'
    return r


def six_spheres(M):
    """'Scene: 6 Sphere Triangles transformed', simple_raytracer.cpp:622-673.  Clones get colours but
    no material entry -> objProperties default-inserts (0,0,0) on the hot path."""
    r = Recipe()
    r.load("sphere.obj", "sphere")
    r.transform("sphere.obj", M.translate(0.0, 6.0, 30.0))
    for k, pos in enumerate([(6.0, 0.0, 0.0), (-6.0, 0.0, 0.0), (0.0, -12.0, 0.0), (6.0, -12.0, 0.0), (-6.0, -12.0, 0.0)], start=1):
        nm = f"sphere{k}.obj"
        r.clone("sphere.obj", nm)
        r.color(nm, (1.0, 0.0, 0.0))
        r.transform(nm, M.translate(*pos))
    for nm in ["sphere.obj"] + [f"sphere{k}.obj" for k in range(1, 6)]:
        r.bvh(nm)
    r.light = LIGHT_DEFAULT[:3]
    return r


def four_cubes(M, angle=0.0):
    """'Scene with 4 Cubes in different colors', simple_raytracer.cpp:726-769 (cross-object shadows)."""
    r = Recipe(); inv = _orbit_view(M, 100.0, angle, 0.0, 0.0)
    r.load("cube.obj", "cube")
    r.color("cube.obj", (1.0, 1.0, 0.0))
    r.transform("cube.obj", M.scale(10.0, 10.0, 10.0))
    for nm, col, pos in [("cube1.obj", (1.0, 0.0, 1.0), (0.0, -15.0, -15.0)), ("cube2.obj", (1.0, 0.0, 0.0), (0.0, -15.0, 15.0)),
                         ("cube3.obj", (0.0, 1.0, 0.0), (0.0, 15.0, 15.0))]:
        r.clone("cube.obj", nm); r.color(nm, col); r.transform(nm, M.translate(*pos))
    r.transform("cube.obj", M.translate(0.0, 15.0, -15.0))
    for nm in ["cube.obj", "cube1.obj", "cube2.obj", "cube3.obj"]:
        r.transform(nm, inv)
    for nm in ["cube.obj", "cube1.obj", "cube2.obj", "cube3.obj"]:
        r.bvh(nm)
    r.light = M.mul_vec4(inv, LIGHT_DEFAULT)[:3]
    return r


def main_scene_no_cats(M, angle=0.0):
    """The scene the reference's checked-in main() builds (simple_raytracer.cpp:546-618: ground cube, bunny, three
    textured trees) WITHOUT the two cats, whose cat.obj is a missing blob.  Tree clones copy the triangles and the
    material of tree.obj (specular strength set to 0, :595-600); each object goes into view space (:558 etc.)."""
    r = Recipe(); inv = _orbit_view(M, 50.0, angle, -50.0, 30.0)
    r.load("./obj/cube.obj", "cube")
    r.color("./obj/cube.obj", (0.0, 1.0, 0.0))
    r.transform("./obj/cube.obj", M.scale(35.0, 35.0, 35.0))
    r.transform("./obj/cube.obj", M.translate(0.0, 10.0, 0.0))
    r.transform("./obj/cube.obj", inv)
    r.bvh("./obj/cube.obj")
    r.load("./obj/stanford-bunny.obj", "bunny")
    r.color("./obj/stanford-bunny.obj", (0.9, 0.9, 0.9))
    r.transform("./obj/stanford-bunny.obj", M.scale(50.0, 50.0, 50.0))
    r.transform("./obj/stanford-bunny.obj", M.rotx(M.radians(181.0)))
    r.transform("./obj/stanford-bunny.obj", M.roty(M.radians(90.0)))
    r.transform("./obj/stanford-bunny.obj", M.translate(25.0, -23.0, 0.0))
    r.transform("./obj/stanford-bunny.obj", inv)
    r.bvh("./obj/stanford-bunny.obj")
    r.load("./obj/tree/tree.obj", "tree")
    r.props("./obj/tree/tree.obj", (0.2, 0.0, 15.0))                       # objProperties[...].y = 0.0f (:595)
    for nm in ("./obj/tree/tree.obj1", "./obj/tree/tree.obj2"):
        r.clone("./obj/tree/tree.obj", nm); r.props(nm, (0.2, 0.0, 15.0))    # :597-600
    for nm, sc, z in (("./obj/tree/tree.obj", 0.03, -25.0), ("./obj/tree/tree.obj1", 0.035, 0.0), ("./obj/tree/tree.obj2", 0.03, 25.0)):
        r.transform(nm, M.scale(sc, sc, sc))
        r.transform(nm, M.rotx(M.radians(-90.0)))
        r.transform(nm, M.translate(-6.0, -25.0, z))
        r.transform(nm, inv)
        r.bvh(nm)
    r.light = M.mul_vec4(inv, LIGHT_DEFAULT)[:3]
    return r


def composite_k4(M, angle=0.0):
    """BASELINE config 4: the composite scene of the reference's main() (ground cube :554-559, bunny :583-591,
    three textured trees :594-618) with the assets that are present standing in for the two cats (cat.obj is a missing blob): the
    horse in cat 0's place (:567-572, same rotations) and the house -- without its 'Plane' object, whose texture is a missing blob
    -- near cat 1's (:574-579).  Like the cats, the horse gets specular strength 0 (:564); each object goes into view space."""
    r = main_scene_no_cats(M, angle)
    inv = _orbit_view(M, 50.0, angle, -50.0, 30.0)
    r.load("./obj/horse/horse.obj", "horse")
    r.props("./obj/horse/horse.obj", (0.2, 0.0, 15.0))
    r.transform("./obj/horse/horse.obj", M.scale(0.0125, 0.0125, 0.0125))
    r.transform("./obj/horse/horse.obj", M.rotx(M.radians(-90.0)))
    r.transform("./obj/horse/horse.obj", M.roty(M.radians(125.0)))
    r.transform("./obj/horse/horse.obj", M.translate(25.0, -25.0, -16.0))
    r.transform("./obj/horse/horse.obj", inv)
    r.bvh("./obj/horse/horse.obj")
    r.load("./obj/house/house.obj", "house_noplane")
    r.transform("./obj/house/house.obj", M.scale(0.03, 0.03, 0.03))
    r.transform("./obj/house/house.obj", M.rotx(M.radians(180.0)))
    r.transform("./obj/house/house.obj", M.translate(14.0, -25.0, 20.0))
    r.transform("./obj/house/house.obj", inv)
    r.bvh("./obj/house/house.obj")
    return r


def cube_over_ground(M):
    """BASELINE config 2 ('cube.obj, 1920x1080, Phong + 1 hard-shadow ray'): a single object can never
    be shadowed (shadowIntersection:331), so the cube of :714-717 stands over a second ground cube
    (SURVEY.md s8d K2)."""
    r = Recipe()
    r.load("cube.obj", "cube")
    r.color("cube.obj", (0.9, 0.3, 0.2))
    r.transform("cube.obj", M.scale(60.0, 60.0, 60.0))
    r.transform("cube.obj", M.roty(M.radians(25.0)))
    r.transform("cube.obj", M.translate(-40.0, 40.0, 330.0))
    r.bvh("cube.obj")
    r.load("ground.obj", "cube")
    r.color("ground.obj", (0.3, 0.6, 0.9))
    r.transform("ground.obj", M.scale(500.0, 10.0, 400.0))
    r.transform("ground.obj", M.translate(0.0, 115.0, 400.0))
    r.bvh("ground.obj")
    r.light = (300.0, -600.0, -100.0)
    return r


def ground_bunny(M):
    """BASELINE config 3 bench placement: frame-filling bunny over a ground slab."""
    r = Recipe()
    r.load("./obj/stanford-bunny.obj", "bunny")
    r.color("./obj/stanford-bunny.obj", (0.9, 0.9, 0.9))          # :584
    r.transform("./obj/stanford-bunny.obj", M.scale(1500.0, 1500.0, 1500.0))
    r.transform("./obj/stanford-bunny.obj", M.rotx(M.radians(180.0)))
    r.transform("./obj/stanford-bunny.obj", M.translate(20.0, 170.0, 300.0))
    r.bvh("./obj/stanford-bunny.obj")
    r.load("cube.obj", "cube")
    r.color("cube.obj", (0.2, 0.7, 0.3))
    r.transform("cube.obj", M.scale(400.0, 10.0, 400.0))
    r.transform("cube.obj", M.translate(0.0, 130.0, 350.0))
    r.bvh("cube.obj")
    r.light = (300.0, -600.0, -100.0)
    return r


# ---- synthetic N-triangle soup (BASELINE config 5), fully specified ------------
def _splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def soup_points(n_tris, seed=0x5EED, extent=2000.0, z0=500.0, z1=1500.0, size=8.0):
    """n_tris x 3 x 4 homogeneous points: centre uniform in [-extent,extent]^2 x [z0,z1], vertices
    centre + (u-0.5)*2*size per component; uniforms are the top 24 bits of SplitMix64 as f32 in [0,1)."""
    # vectorised SplitMix64: stream position k is state0 + (k+1)*gamma
    k = np.arange(1, n_tris * 12 + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        st = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = st
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = ((z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).reshape(n_tris, 12)
    c = np.empty((n_tris, 3), np.float32)
    c[:, 0] = (u[:, 0] * np.float32(2.0) - np.float32(1.0)) * np.float32(extent)
    c[:, 1] = (u[:, 1] * np.float32(2.0) - np.float32(1.0)) * np.float32(extent)
    c[:, 2] = np.float32(z0) + u[:, 2] * np.float32(z1 - z0)
    pts = np.ones((n_tris, 3, 4), np.float32)
    off = (u[:, 3:12].reshape(n_tris, 3, 3) - np.float32(0.5)) * np.float32(2.0 * size)
    pts[:, :, :3] = c[:, None, :] + off
    return pts


SOUP_PALETTE = [(0.9, 0.2, 0.2), (0.2, 0.9, 0.2), (0.2, 0.3, 0.9), (0.9, 0.8, 0.2)]


def soup(n_tris, n_objects=4, **kw):
    """Recipe + meshes for the soup: object id = triangle index mod n_objects."""
    pts = soup_points(n_tris, **kw)
    r = Recipe(); meshes = {}
    for k in range(n_objects):
        key = f"soup{k}"
        meshes[key] = pts[k::n_objects]
        r.load(f"soup{k}.obj", key)
        r.color(f"soup{k}.obj", SOUP_PALETTE[k % len(SOUP_PALETTE)])
        r.bvh(f"soup{k}.obj")
    r.light = LIGHT_DEFAULT[:3]
    return r, meshes


def mesh_points(npz):
    """(verts[n,3], faces[m,3]) -> points[m,3,4] with w = 1 (Object.cpp:82-89); textured meshes come back as a dict
    with the loader's per-vertex integer texel coordinates and the decoded texture (Object.cpp:113-119,57)."""
    v, f = npz["v"], npz["f"]
    pts = np.ones((f.shape[0], 3, 4), np.float32)
    pts[:, :, :3] = v[f]
    if "tri_tex" in npz.files:      # several textures per object (house.obj): per-triangle index, -1 = untextured
        names = [str(x) for x in npz["texture_names"]]
        return {"points": pts, "texcoord": npz["texcoord"].astype(np.float32), "tri_tex": npz["tri_tex"].astype(np.int32),
                "texture_names": names, "textures": [npz[f"texture_{k}"] for k in range(len(names))]}
    if "texcoord" in npz.files:
        return {"points": pts, "texcoord": npz["texcoord"].astype(np.float32), "texture_name": str(npz["texture_name"]),
                "texture": npz["texture"]}
    return pts
