#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the COMPILED REFERENCE.

Runs only in the build container (needs /root/reference and oracle/_ref/libsrt_ref.so, built by
`make -C oracle ref`).  Nothing here is read at test time except the .npz files it writes.

    python tests/golden/make_golden.py            # regenerate everything

Fixtures (all produced by the reference's own functions through oracle/ref_harness.cpp):
  kat.npz                  leaf-function known-answer vectors (a4, a5, a8, a8b, a9, light staircase,
                           Transformation.h factories + glm inverse / mat*vec)
  meshes/<key>.npz         triangle meshes as the reference's loader (tinyobjloader) produced them
  jpeg.npz                 small synthetic JPEG files (encoded here with Pillow) and the bytes the reference's
                           stbi_load decodes them to; sha256 of the decode of the reference's own JPEG textures
  polygons.npz             an OBJ of 5..120-corner faces and the triangles the reference's loader (tinyobjloader +
                           mapbox earcut) makes of them
  scene_<name>.npz         recipe (JSON), the flat scene exported from the reference's Node* trees,
                           and per-resolution outputs: rgb8 of sendRaysAndIntersectPointsColors +
                           drawImage's background rule, closest-hit ids, and t / pre-tone-map /
                           post-tone-map floats (full arrays when small, sha256 + subsample when big)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle as po          # noqa: E402
from simple_raytracer_amd import abi       # noqa: E402
import scenes                              # noqa: E402

M = po.RefMat
BG = np.array(abi.REFERENCE_BACKGROUND, np.int32)
REF_OBJ = {"cube": "cube.obj", "sphere": "sphere.obj", "bunny": "./obj/stanford-bunny.obj", "tree": "./obj/tree/tree.obj",
           "horse": "./obj/horse/horse.obj"}
# house.obj (six textures) without the two triangles of its 'Plane' object, whose texture is a missing blob: the reference
# null-derefs when such a triangle is hit (simple_raytracer.cpp:354-358), so no golden can contain them
REF_OBJ_MULTI = {"house_noplane": "./obj/house/house.obj"}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def export_mesh(key):
    s = po.RefScene(); s.load_obj(REF_OBJ[key])
    pts = s.points(REF_OBJ[key])
    assert np.all(pts[..., 3] == 1.0)
    v = pts[..., :3].reshape(-1, 3)
    uv, inv = np.unique(v, axis=0, return_inverse=True)
    f = inv.astype(np.int32).reshape(-1, 3)
    assert np.array_equal(uv[f], pts[..., :3])
    os.makedirs(os.path.join(HERE, "meshes"), exist_ok=True)
    extra = {}
    tc, col, ht, nrm = s.tri_attrs(REF_OBJ[key])
    if ht.any():      # textured mesh: the loader's integer texel coordinates and the texture as stb_image decoded it
        assert ht.all() and tc.max() < 65536 and np.all(tc == np.floor(tc))
        texname = s.tri_texture_name(REF_OBJ[key], 0)
        extra = dict(texcoord=tc.astype(np.uint16), texture_name=np.array(texname), texture=s.texture(texname))
    np.savez_compressed(os.path.join(HERE, "meshes", key + ".npz"), v=uv.astype(np.float32), f=f, **extra)
    if extra:
        return {"points": pts, "texcoord": tc, "texture_name": texname, "texture": extra["texture"]}
    return pts


def export_multi_textured_mesh(key):
    """A mesh whose triangles use several textures, as the reference's loader produced it (per-triangle textureName), minus the
    triangles whose texture failed to load."""
    name = REF_OBJ_MULTI[key]
    s = po.RefScene(); s.load_obj(name)
    pts = s.points(name)
    tc, col, ht, nrm = s.tri_attrs(name)
    tn = [s.tri_texture_name(name, i) for i in range(len(pts))]
    loaded = {n: s.texture(n) for n in sorted(set(tn)) if n}
    keep = np.array([(not n) or loaded[n] is not None for n in tn])
    names = [n for n in loaded if loaded[n] is not None]
    tri_tex = np.array([names.index(n) if n in names else -1 for n in tn], np.int32)[keep]
    pts, tc = pts[keep], tc[keep]
    print(f"   {key}: {len(keep)} triangles loaded, {int((~keep).sum())} dropped (texture missing), {len(names)} textures")
    assert np.all(pts[..., 3] == 1.0) and tc.max() < 65536 and np.all(tc == np.floor(tc))
    v = pts[..., :3].reshape(-1, 3)
    uv, inv = np.unique(v, axis=0, return_inverse=True)
    f = inv.astype(np.int32).reshape(-1, 3)
    assert np.array_equal(uv[f], pts[..., :3])
    np.savez_compressed(os.path.join(HERE, "meshes", key + ".npz"), v=uv.astype(np.float32), f=f, texcoord=tc.astype(np.uint16),
                        tri_tex=tri_tex.astype(np.int8), texture_names=np.array(names), **{f"texture_{k}": loaded[n] for k, n in enumerate(names)})
    return {"points": pts, "texcoord": tc, "tri_tex": tri_tex, "texture_names": names, "textures": [loaded[n] for n in names]}


class RefBuilder:
    """Replays a recipe on the reference's ObjectManager; 'load' goes through the REAL loader."""
    def __init__(self, real_loader=True, meshes=None):
        self.s = po.RefScene(); self.real = real_loader; self.meshes = meshes or {}
    def add_object(self, name, pts): self.s.add_object(name, pts)
    def add_textured_object(self, name, pts, tc, texname, tex): self.s.add_textured_object(name, pts, tc, texname, tex)
    def add_multi_textured_object(self, name, pts, tc, tt, names, texs): self.s.add_multi_textured_object(name, pts, tc, tt, names, texs)
    def clone(self, a, b): self.s.clone(a, b)
    def set_color(self, n, c): self.s.set_color(n, c)
    def set_props(self, n, p): self.s.set_props(n, p)
    def transform(self, n, m): self.s.transform(n, m)
    def build_bvh(self, n): self.s.build_bvh(n)


def replay_on_ref(recipe, meshes, real_loader):
    b = RefBuilder()
    for op in recipe.ops:
        if op[0] == "load" and real_loader and op[1] == REF_OBJ.get(op[2]):
            b.s.load_obj(op[1])        # the reference keys an object by the file name it loaded
        else:
            scenes.apply_op(b, op, meshes)
    return b.s


def outputs(s, flat, light3, W, H, n_lights, full):
    """Reference outputs for one resolution."""
    out = {}
    hit, t, tone, lin = s.trace(W, H, np.array(light3, np.float32), n_lights)
    if n_lights == 1:
        # the real entry point (lightAmount is hard-wired to 1 at simple_raytracer.cpp:445)
        img, n = s.render(W, H, np.array(list(light3) + [1.0], np.float32))
        q = img.astype(np.int32)
        # cross-check harness trace against the real entry point before trusting it
        tq = np.clip((tone * np.float32(255.0)).astype(np.int32), 0, 255)
        assert np.array_equal(q, tq), "harness trace disagrees with sendRaysAndIntersectPointsColors"
    else:
        q = np.clip((tone * np.float32(255.0)).astype(np.int32), 0, 255)
    rgb8 = q.copy(); rgb8[q.sum(-1) == 0] = BG
    pre = f"{W}x{H}_L{n_lights}_"
    out[pre + "hit_id"] = hit
    out[pre + "rgb8"] = rgb8.astype(np.uint8)
    out[pre + "sha_t"] = np.array(sha(t)); out[pre + "sha_lin"] = np.array(sha(lin)); out[pre + "sha_tone"] = np.array(sha(tone))
    if full and W * H <= 160 * 120:
        out[pre + "t"] = t; out[pre + "lin"] = lin; out[pre + "tone"] = tone
    else:
        out[pre + "sub_stride"] = np.array(61)
        out[pre + "sub_t"] = t.reshape(-1)[::61].copy()
        out[pre + "sub_lin"] = lin.reshape(-1, 3)[::61].copy()
        out[pre + "sub_tone"] = tone.reshape(-1, 3)[::61].copy()
    print(f"   {W}x{H} L={n_lights}: {int((hit >= 0).sum())} hit px")
    return out


def band_outputs(s, light3, W, H, n_lights, y0, y1):
    """Reference outputs for rows [y0, y1) of a W x H frame (a whole 3840x2160 frame with 64 light samples would be hours of
    reference time): keys band_<W>x<H>_L<n>_y<y0>_<y1>_*; same contents as outputs()."""
    hit, t, tone, lin = s.trace(W, H, np.array(light3, np.float32), n_lights, rows=(y0, y1))
    q = np.clip((tone * np.float32(255.0)).astype(np.int32), 0, 255)
    rgb8 = q.copy(); rgb8[q.sum(-1) == 0] = BG
    pre = f"band_{W}x{H}_L{n_lights}_y{y0}_{y1}_"
    print(f"   {W}x{H} rows {y0}..{y1} L={n_lights}: {int((hit >= 0).sum())} hit px")
    return {pre + "hit_id": hit, pre + "rgb8": rgb8.astype(np.uint8), pre + "sha_t": np.array(sha(t)), pre + "sha_lin": np.array(sha(lin)),
            pre + "sub_stride": np.array(61), pre + "sub_t": t.reshape(-1)[::61].copy(), pre + "sub_lin": lin.reshape(-1, 3)[::61].copy()}


def make_scene(name, recipe, meshes, renders, textures=None, prebuilt=None, compact=False, bands=()):
    print("scene", name)
    s = prebuilt if prebuilt is not None else replay_on_ref(recipe, meshes, real_loader=True)
    flat = s.export(textures=textures)
    if recipe is not None and prebuilt is None:
        # array-fed replay must give the identical flat scene (this is what tests replay)
        s2 = replay_on_ref(recipe, meshes, real_loader=False)
        f2 = s2.export(textures=textures)
        for k in abi.FlatScene.ARRAYS:
            a, b = getattr(flat, k), getattr(f2, k)
            assert (a is None and b is None) or np.array_equal(a, b), (name, k)
        assert flat.names == f2.names
    if compact:
        # big scene: keep the recipe, the object order and a sha256 of every flat array of the reference's export; tests
        # rebuild the scene with the host mirror (proven to reproduce the reference's trees) and check the hashes
        d = {"scene_names": np.array(flat.names, dtype="U")}
        for k in abi.FlatScene.ARRAYS:
            if getattr(flat, k) is not None:
                d["sha_scene_" + k] = np.array(sha(getattr(flat, k)))
        d["scene_counts"] = np.array([flat.n_objects, flat.n_nodes, flat.n_tris, flat.n_textures])
    else:
        d = flat.to_npz_dict()
    d["recipe"] = np.array(recipe.to_json() if recipe is not None else "")
    d["light"] = np.array(recipe.light if recipe is not None else renders[0][4], np.float32)
    import time
    for (W, H, nl, full, *rest) in renders:
        light3 = recipe.light if recipe is not None else rest[0]
        d.update(outputs(s, flat, light3, W, H, nl, full))
    for (W, H, nl, y0, y1) in bands:
        d.update(band_outputs(s, recipe.light, W, H, nl, y0, y1))
    np.savez_compressed(os.path.join(HERE, f"scene_{name}.npz"), **d)
    print("   ->", os.path.getsize(os.path.join(HERE, f"scene_{name}.npz")) // 1024, "KiB;",
          flat.n_objects, "objects", flat.n_nodes, "nodes", flat.n_tris, "tris; order", flat.names)


def make_texquad():
    """Textured path (softShadow:350-361, Object.cpp:52-68,98-128) on a small synthetic asset that
    goes through the reference's REAL loader (tinyobj + stb_image), plus an untextured occluder."""
    from PIL import Image
    tmp = "/tmp/srt_texquad"; os.makedirs(tmp, exist_ok=True)
    rng = np.random.default_rng(7)
    tex = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    tex[::8] //= 3
    Image.fromarray(tex).save(os.path.join(tmp, "tex.png"))
    with open(os.path.join(tmp, "quad.mtl"), "w") as f:
        f.write(f"newmtl m\nKd 1 1 1\nmap_Kd {tmp}/tex.png\n")
    # 6x6 grid of quads (72 triangles), wavy so that normals differ
    n = 6; lines = ["mtllib quad.mtl", "usemtl m"]
    for j in range(n + 1):
        for i in range(n + 1):
            x = -60 + 120 * i / n; y = -45 + 90 * j / n; z = 300 + 12 * np.sin(i * 1.3) * np.cos(j * 0.9)
            lines.append(f"v {x:.6f} {y:.6f} {z:.6f}")
    for j in range(n + 1):
        for i in range(n + 1):
            lines.append(f"vt {0.02 + 0.96 * i / n:.6f} {0.02 + 0.96 * j / n:.6f}")
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i + 1; b = a + 1; c = a + n + 1; d = c + 1
            lines.append(f"f {a}/{a} {b}/{b} {d}/{d} {c}/{c}")      # quads: exercises tinyobj triangulation
    with open(os.path.join(tmp, "quad.obj"), "w") as f:
        f.write("\n".join(lines) + "\n")
    s = po.RefScene()
    objname = os.path.join(tmp, "quad.obj")
    s.load_obj(objname, cwd=tmp)
    assert s.L.ref_om_num_tris(s.om, objname.encode()) == 72
    texname = s.tri_texture_name(objname, 0)
    s.build_bvh(objname)
    s.load_obj("cube.obj")
    s.set_color("cube.obj", (0.1, 0.5, 0.9))
    s.transform("cube.obj", M.scale(14.0, 14.0, 14.0))
    s.transform("cube.obj", M.roty(M.radians(30.0)))
    s.transform("cube.obj", M.translate(25.0, -30.0, 240.0))
    s.build_bvh("cube.obj")
    light = (260.0, -420.0, -60.0)
    make_scene("texquad", None, {}, [(120, 90, 1, True, light), (64, 48, 5, True, light)],
               textures={objname: texname}, prebuilt=s)


def make_kat():
    rng = np.random.default_rng(20250225)
    d = {}
    # ---- a5 ray/triangle -------------------------------------------------------------------
    n = 1536
    tri = np.ones((n, 3, 4), np.float32)
    c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32); c[..., 2] += 300
    tri[..., :3] = c + rng.uniform(-20, 20, (n, 3, 3)).astype(np.float32)
    ray = np.zeros((n, 6), np.float32)
    tgt = (tri[:, :, :3] * rng.dirichlet([1, 1, 1], n).astype(np.float32)[:, :, None]).sum(1)
    ray[:, 3:] = tgt + rng.normal(0, 6, (n, 3)).astype(np.float32)
    k = n // 6
    ray[:k, :3] = rng.uniform(-30, 30, (k, 3)).astype(np.float32); ray[:k, 3:] = tgt[:k] - ray[:k, :3]      # shadow-like rays
    ray[k:2 * k, 3:] = tri[k:2 * k, 0, :3]                          # exactly through vertex 1 (u = v = 0)
    ray[2 * k:3 * k, 3:] = (tri[2 * k:3 * k, 0, :3] + tri[2 * k:3 * k, 1, :3]) * np.float32(0.5)   # edge midpoints
    tri[3 * k:3 * k + 32, 2] = tri[3 * k:3 * k + 32, 1]             # degenerate (det == 0)
    tri[3 * k + 32:3 * k + 64, :, 3] = rng.uniform(0.5, 2.0, (32, 3)).astype(np.float32)   # w != 1
    ray[3 * k + 64:3 * k + 96, 3:] *= -1                             # behind the origin (t < 0)
    ray[3 * k + 96:3 * k + 128, 3] = 0.0                             # zero direction components
    ray[3 * k + 128:3 * k + 160, 3:5] = 0.0
    ray[3 * k + 160:3 * k + 164, 3:] = 0.0                           # zero direction: NaN path
    ray[3 * k + 164:3 * k + 196, :3] = tri[3 * k + 164:3 * k + 196, 0, :3]   # origin on the triangle (t = 0)
    sel = slice(3 * k + 196, 3 * k + 212)                           # overflow: inf * 0 -> NaN falls through every test
    tri[sel, :, :3] *= np.float32(1e25); ray[sel, 3:] *= np.float32(1e20)
    d["rt_ray"] = ray; d["rt_tri"] = tri.reshape(n, 12)
    d["rt_t"] = po.ref_kat_ray_triangle(ray, tri.reshape(n, 12))
    # integer pixel rays against scene-scale triangles, like the hot path
    n2 = 1024
    ray2 = np.zeros((n2, 6), np.float32)
    ray2[:, 3] = rng.integers(-960, 960, n2); ray2[:, 4] = rng.integers(-540, 540, n2); ray2[:, 5] = 400.0
    tri2 = np.ones((n2, 3, 4), np.float32)
    tt = rng.uniform(0.5, 3.0, (n2, 1, 1)).astype(np.float32)
    tri2[..., :3] = ray2[:, None, 3:] * tt + rng.uniform(-40, 40, (n2, 3, 3)).astype(np.float32)
    d["rt2_ray"] = ray2; d["rt2_tri"] = tri2.reshape(n2, 12); d["rt2_t"] = po.ref_kat_ray_triangle(ray2, tri2.reshape(n2, 12))
    # ---- a4 ray/box ------------------------------------------------------------------------
    n = 2048
    lo = rng.uniform(-100, 100, (n, 3)).astype(np.float32); lo[:, 2] += 300
    hi = lo + rng.uniform(0, 80, (n, 3)).astype(np.float32)
    ray = np.zeros((n, 6), np.float32)
    ray[:, 3:] = (lo + hi) * np.float32(0.5) + rng.normal(0, 45, (n, 3)).astype(np.float32)
    k = n // 8
    ray[:k, :3] = rng.uniform(-80, 80, (k, 3)).astype(np.float32)                 # non-zero origins
    ray[k:2 * k, 3] = 0.0                                                           # d.x == 0  -> +-inf / NaN
    ray[2 * k:3 * k, 4] = 0.0
    ray[3 * k:3 * k + 64, 3:5] = 0.0
    hi[3 * k + 64:3 * k + 192, 1] = lo[3 * k + 64:3 * k + 192, 1]                   # flat boxes
    sel = slice(3 * k + 192, 3 * k + 320)
    ray[sel, :3] = 0; ray[sel, 3:] = hi[sel]                                        # through the max corner
    sel = slice(3 * k + 320, 3 * k + 448)
    ray[sel, :3] = 0; ray[sel, 3:] = lo[sel]; ray[sel, 3] = 0.0; lo[sel, 0] = 0.0    # 0/0 NaN on x
    sel = slice(3 * k + 448, 3 * k + 512)
    lo[sel] = np.float32(3.4028235e38); hi[sel] = np.float32(-3.4028235e38)         # empty box (Object.cpp:207-208)
    sel = slice(3 * k + 512, 3 * k + 640)
    ray[sel, 3:] *= -1                                                              # box behind the origin still passes
    box = np.concatenate([lo, hi], 1)
    d["ab_ray"] = ray; d["ab_box"] = box
    d["ab_hit"], d["ab_hit_origin0"] = po.ref_kat_ray_aabb(ray, box)
    # ---- a8 phong ----------------------------------------------------------------------------
    n = 512
    inp = np.zeros((n, 28), np.float32)
    tri = np.ones((n, 3, 4), np.float32)
    c = rng.uniform(-80, 80, (n, 1, 3)).astype(np.float32); c[..., 2] += 300
    tri[..., :3] = c + rng.uniform(-25, 25, (n, 3, 3)).astype(np.float32)
    bary = rng.dirichlet([1, 1, 1], n).astype(np.float32)
    P = (tri[:, :, :3] * bary[:, :, None]).sum(1)
    tt = rng.uniform(0.3, 2.5, n).astype(np.float32)
    inp[:, 3:6] = P / tt[:, None]; inp[:, 27] = tt
    inp[:, 6:18] = tri.reshape(n, 12)
    inp[:, 18:21] = rng.uniform(-600, 600, (n, 3)).astype(np.float32)
    inp[:, 21:24] = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    inp[:, 24] = rng.uniform(0, 0.5, n); inp[:, 25] = rng.uniform(0, 1, n); inp[:, 26] = rng.choice([0.0, 1.0, 5.0, 15.0, 15.0, 32.5, 100.0], n)
    inp[:16, 24:27] = 0.0                                  # default-inserted (0,0,0) material: pow(x, 0)
    d["ph_in"] = inp; d["ph_rgb"] = po.ref_kat_phong(inp)
    # ---- a8b barycentric ---------------------------------------------------------------------
    inb = np.zeros((n, 15), np.float32); inb[:, :12] = tri.reshape(n, 12); inb[:, 12:] = P
    d["bc_in"] = inb; d["bc_uvw"] = po.ref_kat_barycentric(inb)
    # ---- a9 tone map + quantiser -------------------------------------------------------------
    lin = np.concatenate([rng.uniform(0, 2, (700, 3)), rng.uniform(0, 64, (200, 3)), rng.uniform(0, 1e-3, (100, 3)),
                          np.zeros((8, 3)), np.full((8, 3), 0.5)]).astype(np.float32)
    d["tm_lin"] = lin; d["tm_tone"], d["tm_q"] = po.ref_kat_tonemap(lin)
    # ---- softShadow light staircase (:363-383) via the reference's float adds -----------------
    # (the harness reproduces the loop with glm::vec3 += 3.0f; recorded here through ref_trace's
    #  own table by construction: rebuild it with f32 numpy adds and pin 64 entries)
    d["ls_base"] = np.array(scenes.LIGHT_DEFAULT[:3], np.float32)
    d["ls_table"] = abi.light_staircase(scenes.LIGHT_DEFAULT[:3], 64)
    # ---- Transformation.h factories + glm ops --------------------------------------------------
    angs = np.array([0.0, 25.0, 30.0, 90.0, 125.0, 180.0, 181.0, -90.0, 70.0, 350.0], np.float32)
    d["tf_deg"] = angs
    d["tf_rad"] = np.array([M.radians(float(a)) for a in angs], np.float32)
    d["tf_rotx"] = np.stack([M.rotx(float(r)) for r in d["tf_rad"]])
    d["tf_roty"] = np.stack([M.roty(float(r)) for r in d["tf_rad"]])
    d["tf_rotz"] = np.stack([M.rotz(float(r)) for r in d["tf_rad"]])
    sc = rng.uniform(0.01, 50, (8, 3)).astype(np.float32)
    d["tf_scale_in"] = sc; d["tf_scale"] = np.stack([M.scale(*map(float, v)) for v in sc])
    d["tf_translate"] = np.stack([M.translate(*map(float, v)) for v in sc])
    d["tf_mirror"] = np.stack([M.mirror(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1)])
    sh = rng.uniform(-1, 1, (4, 6)).astype(np.float32)
    d["tf_shear_in"] = sh; d["tf_shear"] = np.stack([M.shear(*map(float, v)) for v in sh])
    vp = rng.uniform(-60, 60, (10, 3)).astype(np.float32); vr = rng.uniform(-3, 3, (10, 3)).astype(np.float32)
    d["tf_view_pos"] = vp; d["tf_view_rot"] = vr
    d["tf_view"] = np.stack([M.view(vp[i], vr[i]) for i in range(10)])
    d["tf_view_inv"] = np.stack([M.inverse(m) for m in d["tf_view"]])
    d["tf_mul"] = np.stack([M.mul(d["tf_view"][i], d["tf_view"][(i + 1) % 10]) for i in range(10)])
    v4 = np.concatenate([rng.uniform(-500, 500, (10, 3)), np.ones((10, 1))], 1).astype(np.float32)
    d["tf_vec"] = v4; d["tf_mulvec"] = np.stack([M.mul_vec4(d["tf_view_inv"][i], v4[i]) for i in range(10)])
    # ---- interpolateNormal (:132-140), drawn AFTER everything above so that earlier vectors keep their values ----
    nin = np.zeros((256, 12), np.float32)
    nv = rng.normal(0, 1, (256, 3, 3)).astype(np.float32)
    nv /= np.linalg.norm(nv, axis=2, keepdims=True).astype(np.float32)
    nin[:, :9] = nv.reshape(256, 9); nin[:, 9:] = rng.dirichlet([1, 1, 1], 256).astype(np.float32)
    nin[:8, :9] = 0.0                                   # missing normals (loader default 0): 0 * inf -> NaN
    d["in_in"] = nin; d["in_out"] = po.ref_kat_interp_normal(nin)
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **d)
    print("kat.npz", os.path.getsize(os.path.join(HERE, "kat.npz")) // 1024, "KiB")


def make_jpeg():
    """Small synthetic JPEGs (encoded here with Pillow: our own inputs) and what the reference's stbi_load(path, ..., 3)
    (Object.cpp:57) decodes them to: baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0 / grey / CMYK, restart intervals,
    odd sizes down to one pixel, low and high quality.  Plus sha256 of the decode of the reference's own JPEG assets."""
    import io
    import tempfile
    from PIL import Image
    rng = np.random.default_rng(7)
    def picture(w, h):
        y, x = np.mgrid[0:h, 0:w]
        base = np.stack([(x * 255) // max(w - 1, 1), (y * 255) // max(h - 1, 1), ((x + y) * 127) // max(w + h - 2, 1)], -1)
        blobs = (rng.integers(0, 2, (h // 4 + 1, w // 4 + 1, 3)) * 255).repeat(4, 0).repeat(4, 1)[:h, :w]
        return np.clip(base * 0.6 + blobs * 0.4 + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)
    cases = [
        ("base444", 48, 40, dict(quality=90, subsampling=0)),
        ("base422", 50, 33, dict(quality=85, subsampling=1)),
        ("base420", 57, 39, dict(quality=75, subsampling=2)),
        ("base420_q20", 64, 48, dict(quality=20, subsampling=2)),
        ("base444_q100", 31, 17, dict(quality=100, subsampling=0)),
        ("prog444", 40, 40, dict(quality=88, subsampling=0, progressive=True)),
        ("prog420", 61, 35, dict(quality=70, subsampling=2, progressive=True)),
        ("prog422_q40", 33, 50, dict(quality=40, subsampling=1, progressive=True)),
        ("rst444", 64, 40, dict(quality=90, subsampling=0, restart_marker_blocks=3)),
        ("rst420", 70, 45, dict(quality=80, subsampling=2, restart_marker_rows=1)),
        ("prog_rst420", 52, 52, dict(quality=80, subsampling=2, progressive=True, restart_marker_blocks=2)),
        ("grey", 45, 29, dict(quality=85, mode="L")),
        ("grey_prog", 23, 41, dict(quality=60, mode="L", progressive=True)),
        ("cmyk", 36, 28, dict(quality=90, mode="CMYK")),
        ("one_pixel", 1, 1, dict(quality=90, subsampling=2)),
        ("one_column", 1, 19, dict(quality=90, subsampling=2)),
        ("one_row", 21, 1, dict(quality=90, subsampling=1)),
        ("optimized", 40, 30, dict(quality=85, subsampling=2, optimize=True)),
    ]
    d = {"names": np.array([c[0] for c in cases])}
    with tempfile.TemporaryDirectory() as tmp:
        for name, w, h, kw in cases:
            kw = dict(kw)
            mode = kw.pop("mode", "RGB")
            im = Image.fromarray(picture(w, h), "RGB").convert(mode)
            buf = io.BytesIO(); im.save(buf, "JPEG", **kw)
            path = os.path.join(tmp, name + ".jpg")
            open(path, "wb").write(buf.getvalue())
            ref = po.ref_stbi_load(path)
            assert ref is not None and ref.shape == (h, w, 3), name
            d[f"{name}_file"] = np.frombuffer(buf.getvalue(), np.uint8)
            d[f"{name}_rgb"] = ref
    assets = ["obj/tree/10445_Oak_Tree_v1_diffuse.jpg", "obj/grass/10438_Circular_Grass_Patch_v1_Diffuse.jpg", "obj/horse/Horse_v01.jpg",
              "obj/cat/Cat_diffuse.jpg", "obj/bird/12248_Bird_v1_diff.jpg", "obj/dog/13466_Canaan_Dog_diff.jpg", "obj/cat/Cat_bump.jpg"]
    d["asset_names"] = np.array(assets)
    d["asset_sha"] = np.array([sha(po.ref_stbi_load(os.path.join("/root/reference", a))) for a in assets])
    np.savez_compressed(os.path.join(HERE, "jpeg.npz"), **d)
    print("jpeg.npz", os.path.getsize(os.path.join(HERE, "jpeg.npz")) // 1024, "KiB")


def make_polygons():
    """An OBJ of faces with 5..120 corners (star-shaped, concave combs, collinear and duplicate corners, a bow-tie, a
    figure-eight, tilted planes, slightly non-planar) and the triangles the reference's loader (tinyobjloader built with
    mapbox earcut, simple_raytracer.cpp:15-16) makes of it."""
    import tempfile
    rng = np.random.default_rng(11)
    verts, faces = [], []
    def add(poly2d, frame=None, wobble=0.0):
        poly2d = np.asarray(poly2d, np.float64)
        if frame is None:
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            frame = (q[:, 0], q[:, 1], rng.normal(size=3) * 3)
        u, v, o = frame
        base = len(verts)
        for x, y in poly2d:
            p = o + u * x + v * y + np.cross(u, v) * rng.normal() * wobble
            verts.append(p)
        faces.append(list(range(base + 1, base + 1 + len(poly2d))))
    for n in (5, 6, 7, 8, 13, 32, 79, 80, 81, 120):                       # star-shaped, random radii (concave)
        a = np.sort(rng.uniform(0, 2 * np.pi, n)); r = rng.uniform(0.4, 2.0, n)
        add(np.stack([r * np.cos(a), r * np.sin(a)], 1))
        add(np.stack([r * np.cos(a), r * np.sin(a)], 1)[::-1], wobble=0.01)   # other winding, slightly non-planar
    comb = [(0, 0)] + [p for k in range(6) for p in ((k + 0.2, 3), (k + 0.5, 0.5), (k + 0.8, 3))] + [(6, 0)]
    add(comb); add(comb, frame=(np.array([1., 0, 0]), np.array([0, 1., 0]), np.zeros(3)))
    add([(0, 0), (1, 0), (2, 0), (3, 0), (3, 1), (3, 2), (1.5, 2), (0, 2), (0, 1)])              # collinear corners
    add([(0, 0), (2, 0), (2, 0), (2, 2), (1, 1), (0, 2), (0, 2)])                                # duplicates
    add([(0, 0), (2, 2), (2, 0), (0, 2), (1, 3)])                                                # bow-tie
    add([(0, 0), (1, 1), (2, 0), (3, 1), (4, 0), (4, 2), (3, 1.2), (2, 2), (1, 1.2), (0, 2)])   # pinched
    add([(0, 0), (4, 0), (4, 4), (0, 4), (0, 1), (3, 1), (3, 3), (1, 3), (1, 2), (2, 2), (2, 1.5), (0.5, 1.5), (0.5, 3.5), (3.5, 3.5), (3.5, 0.5), (0, 0.5)])  # spiral
    add([(0, 0), (1, 0), (2, 0), (3, 0), (4, 0)])                                                # degenerate: zero area
    add([(0, 0), (1, 0), (1, 1), (0, 1), (0.5, 0.5)], frame=(np.array([0, 0, 1.]), np.array([0, 1., 0]), np.array([5., 0, 0])))   # normal along x
    text = "".join("v %.6f %.6f %.6f\n" % tuple(p) for p in verts) + "".join("f " + " ".join(map(str, f)) + "\n" for f in faces)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "polys.obj")
        open(path, "w").write(text)
        r = po.RefScene(); r.load_obj(path, cwd=tmp)
        pts = r.points(path)
    np.savez_compressed(os.path.join(HERE, "polygons.npz"), obj=np.frombuffer(text.encode(), np.uint8), points=pts)
    print("polygons.npz", len(faces), "faces ->", len(pts), "triangles")


def make_k4(meshes=None):
    """BASELINE configs[3] at its own shape: the composite scene with horse and house, 64 light samples
    (lightAmount = 64, softShadow:348,366-383): a 320x180 frame in full and a band of scanlines of the 3840x2160 frame."""
    if meshes is None:
        meshes = {k: export_mesh(k) for k in ("cube", "bunny", "tree", "horse")}
        meshes["house_noplane"] = export_multi_textured_mesh("house_noplane")
    make_scene("k4", scenes.composite_k4(M, 0.0), meshes, [(320, 180, 64, False), (160, 90, 9, True)], compact=True,
               bands=[(3840, 2160, 64, 1000, 1008), (3840, 2160, 64, 1120, 1128)])


def main():
    assert po.ref_available(), "build oracle/_ref first: make -C oracle ref"
    if len(sys.argv) > 1 and sys.argv[1] in ("jpeg", "polygons", "k4"):
        {"jpeg": make_jpeg, "polygons": make_polygons, "k4": make_k4}[sys.argv[1]]()
        return
    make_jpeg()
    make_polygons()
    meshes = {k: export_mesh(k) for k in REF_OBJ}
    make_kat()
    make_scene("cube", scenes.one_cube(M, 0.0), meshes, [(256, 256, 1, True), (37, 23, 1, True)])
    make_scene("sphere", scenes.sphere(M), meshes, [(256, 256, 1, True)])
    make_scene("cubes4_a0", scenes.four_cubes(M, 0.0), meshes, [(256, 256, 1, True), (128, 96, 8, True)])
    make_scene("cubes4_a40", scenes.four_cubes(M, 40.0), meshes, [(200, 150, 1, True), (121, 91, 3, True)])
    make_scene("spheres6", scenes.six_spheres(M), meshes, [(160, 120, 1, True)])
    make_scene("cube_ground", scenes.cube_over_ground(M), meshes, [(240, 135, 1, True), (1920, 1080, 1, False)])
    make_scene("ground_bunny", scenes.ground_bunny(M), meshes, [(192, 108, 1, True), (96, 54, 4, True), (1920, 1080, 1, False)])
    make_texquad()
    # the scene of the reference's checked-in main() (minus the cats, a missing blob) at the reference's own 600x400
    tex = meshes["tree"]["texture_name"]
    make_scene("main_nocats", scenes.main_scene_no_cats(M, 0.0), meshes, [(600, 400, 1, False), (150, 100, 4, True)],
               textures={"./obj/tree/tree.obj": tex, "./obj/tree/tree.obj1": tex, "./obj/tree/tree.obj2": tex}, compact=True)
    meshes["house_noplane"] = export_multi_textured_mesh("house_noplane")
    make_k4(meshes)


if __name__ == "__main__":
    main()
