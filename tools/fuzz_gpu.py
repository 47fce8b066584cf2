#!/usr/bin/env python3
"""Randomised parity run on the GPU: small scenes of many shapes (few big objects, many small ones, cubes, slivers, shared triangles),
random frame sizes, light-sample counts 1..12, whole frames and scanline-block shares -- the SHIPPED (non-counting) pipelines
against the CPU oracle: hit ids and t bit for bit, linear colour within the tolerance, rgb8 within 1 LSB on a bounded number of
pixels; and the batch call against the single renders.  Usage (GPU box): python tools/fuzz_gpu.py [--seeds 40] [--first 0]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simple_raytracer_amd import abi, host, lib      # noqa: E402
from oracle import pyoracle                          # noqa: E402
import golden_util as gu                             # noqa: E402
import scenes                                        # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def random_scene(rng, cube, bunny, pose=None):
    """pose = (angle, dx): every object turned and moved once more before its hierarchy is built (the next frame of an orbit)"""
    recipe = scenes.Recipe(); meshes = {"cube": cube}
    T = host.Transformation
    n_obj = int(rng.integers(1, 9))
    for k in range(n_obj):
        kind = rng.integers(0, 5)
        name = f"obj{k}"
        if kind == 4:                                  # a textured soup: per-vertex integer texel coordinates into a random image
            n = int(rng.integers(1, 120))
            c = rng.uniform(-150, 150, (n, 1, 3)); c[..., 2] += 380
            pts = np.ones((n, 3, 4), np.float32)
            pts[..., :3] = c + rng.uniform(-1, 1, (n, 3, 3)) * rng.choice([25.0, 120.0])
            tw, th = int(rng.integers(2, 40)), int(rng.integers(2, 40))
            tcs = np.stack([rng.integers(0, tw, (n, 3)), rng.integers(0, th, (n, 3))], -1).reshape(n, 6).astype(np.float32)
            meshes[f"m{k}"] = {"points": pts, "texcoord": tcs, "texture_name": f"tex{k}.png", "texture": rng.integers(0, 256, (th, tw, 3), dtype=np.uint8)}
            recipe.load(name, f"m{k}")
        elif kind == 3:                                  # a piece of the bunny: a deep tree of small triangles
            n = int(rng.integers(200, 6000))
            meshes[f"m{k}"] = bunny[int(rng.integers(0, bunny.shape[0] - n)):][:n]
            recipe.load(name, f"m{k}")
            recipe.transform(name, T.scaleObj(1500.0, 1500.0, 1500.0)); recipe.transform(name, T.rotateObjX(T.radians(180.0)))
            recipe.transform(name, T.changeObjPosition(float(rng.uniform(-80, 80)), float(rng.uniform(100, 220)), float(rng.uniform(250, 400))))
        elif kind == 0:                                  # a transformed cube (root + two leaves): the ground of the reference's scenes
            recipe.load(name, "cube")
            recipe.transform(name, T.scaleObj(*[float(x) for x in rng.uniform(5, 300, 3)]))
            recipe.transform(name, T.rotateObjY(T.radians(float(rng.uniform(0, 90)))))
            recipe.transform(name, T.changeObjPosition(float(rng.uniform(-150, 150)), float(rng.uniform(-100, 150)), float(rng.uniform(200, 600))))
        else:                                          # a soup of n triangles, some of them big
            n = int(rng.integers(1, 400 if kind == 1 else 40))
            c = rng.uniform(-150, 150, (n, 1, 3)); c[..., 2] += 380
            pts = np.ones((n, 3, 4), np.float32)
            pts[..., :3] = c + rng.uniform(-1, 1, (n, 3, 3)) * rng.choice([3.0, 25.0, 120.0])
            meshes[f"m{k}"] = pts
            recipe.load(name, f"m{k}")
        if pose is not None:
            recipe.transform(name, T.rotateObjY(T.radians(pose[0]))); recipe.transform(name, T.changeObjPosition(pose[1], 0.0, 0.0))
        recipe.color(name, rng.uniform(0, 1, 3)); recipe.bvh(name)
    recipe.light = tuple(float(x) for x in rng.uniform(-500, 500, 3))
    return recipe, meshes


def replay(recipe, meshes):
    om = host.ObjectManager(); recipe.replay(om, meshes)
    return om


def source_attrs(om, flat):
    tc, nrm, tex = np.zeros_like(flat.tri_texcoord), np.zeros_like(flat.tri_normals), np.full(flat.n_tris, -1, np.int32)
    base = 0
    for nme in flat.names:
        order = om.hierarchy(nme)[1]
        src = base + order.astype(np.int64); vis = base + np.arange(order.shape[0])
        tc.reshape(-1, 6)[src] = flat.tri_texcoord.reshape(-1, 6)[vis]; nrm.reshape(-1, 9)[src] = flat.tri_normals.reshape(-1, 9)[vis]
        tex[src] = flat.tri_tex[vis]
        base += order.shape[0]
    return tc, nrm, tex


def update_frame_leg(seed, cube, bunny, ds, flat, p):
    rng0 = np.random.default_rng(1000 + seed)
    recipe0, meshes0 = random_scene(rng0, cube, bunny)
    ds.set_source(*source_attrs(replay(recipe0, meshes0), flat))
    rng1 = np.random.default_rng(1000 + seed)
    recipe1, meshes1 = random_scene(rng1, cube, bunny, pose=(float(seed % 90) + 3.5, float(seed % 7) - 3.0))
    om1 = replay(recipe1, meshes1)
    flat1 = om1.flatten()
    hs = [om1.hierarchy(nme) for nme in flat1.names]
    ds.update_frame([h[0] for h in hs], [h[1] for h in hs], [h[2] for h in hs], [h[3] for h in hs], obj_color=flat1.obj_color, obj_material=flat1.obj_material)
    fresh = lib.DeviceScene(flat1)
    got, want = ds.records(), fresh.records()
    ok = all(np.array_equal(np.ascontiguousarray(got[k]).view(np.uint8), np.ascontiguousarray(want[k]).view(np.uint8)) for k in ("nodes", "wide", "root_nodes", "tris", "tris_o", "tri_tex", "tri_texcoord", "tri_normals"))
    o, f = ds.render(p), fresh.render(p)
    return ok and np.array_equal(o["hit_id"], f["hit_id"]) and np.array_equal(bits(o["t"]), bits(f["t"])) and np.array_equal(bits(o["rgb_linear"]), bits(f["rgb_linear"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--update-frames", action="store_true", help="also check srt_scene_update_frame against srt_scene_create on a second pose of every scene")
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    pyoracle.oracle_lib()
    cube = gu.load_mesh("cube"); bunny = gu.load_mesh("bunny")
    bad = 0
    for seed in range(a.first, a.first + a.seeds):
        rng = np.random.default_rng(1000 + seed)
        recipe, meshes = random_scene(rng, cube, bunny)
        flat = host.build_flat_scene(recipe, meshes)
        W, H = int(rng.integers(17, 200)), int(rng.integers(9, 150))
        L = int(rng.choice([1, 1, 2, 3, 4, 7, 8, 9, 12, 20, 64, 70]))
        lights = abi.light_staircase(recipe.light, L)
        kw = {}
        if rng.random() < 0.5:
            stride = int(rng.integers(2, 5)); kw = dict(block_rows=8 * int(rng.integers(1, 3)), block_first=int(rng.integers(0, stride)), block_stride=stride)
        if rng.random() < 0.25:                        # camera mode (extension): the rays are taken into the scene's space by a view matrix
            kw["ray_matrix"] = scenes.orbit_view_matrix(host.Transformation, float(rng.uniform(0, 60)), float(rng.uniform(-40, 40)), float(rng.uniform(-30, 30)), float(rng.uniform(-10, 10)))
        if rng.random() < 0.3:                         # the in-flight hint (scheduling only): results must not depend on it
            kw["flags"] = abi.SRT_FLAG_FRAMES_IN_FLIGHT
        p = abi.make_params(W, H, lights, **kw)
        ds = lib.DeviceScene(flat)
        if ds.rows(p) == 0:
            continue
        o = ds.render(p)
        c = pyoracle.render(flat, p)
        ok = np.array_equal(o["hit_id"], c["hit_id"]) and np.array_equal(bits(o["t"]), bits(c["t"]))
        fin = np.isfinite(c["rgb_linear"]).all(-1)
        tol = 1e-4 * max(1.0, float(np.abs(c["rgb_linear"][fin]).max())) if fin.any() else 1e-4
        ok = ok and (not fin.any() or float(np.abs(o["rgb_linear"][fin] - c["rgb_linear"][fin]).max()) < tol)
        d8 = np.abs(o["rgb8"].astype(np.int32) - c["rgb8"].astype(np.int32))
        ok = ok and d8.max() <= 1 and int((d8.max(-1) > 0).sum()) <= max(2, W * H // 2000)
        # the other forms of the node-queue kernels (64 B records everywhere, 32 B everywhere, the round-2 queue order): the same frame bit for bit
        if "ray_matrix" not in kw:
            for variant in (40, 41, 42):
                ov = ds.render(abi.make_params(W, H, lights, flags=variant << 8, **{k: v for k, v in kw.items() if k != "flags"}))
                ok = ok and np.array_equal(ov["hit_id"], o["hit_id"]) and np.array_equal(bits(ov["t"]), bits(o["t"])) and np.array_equal(bits(ov["rgb_linear"]), bits(o["rgb_linear"]))
        # the batch call: this frame and a second one with the light moved, on shared records, against the single renders
        twin = ds.share()
        light2 = np.asarray(recipe.light, np.float32) + np.float32(17.0)
        p2 = abi.make_params(W, H, abi.light_staircase(light2, L), **kw)
        import ctypes as C
        shape = o["hit_id"].shape
        def pinned(dtype, tail=()):
            n = int(np.prod(shape + tail)) * np.dtype(dtype).itemsize
            ptr = lib.load().srt_host_alloc(n)
            return ptr, np.frombuffer((C.c_uint8 * n).from_address(ptr), dtype=dtype).reshape(shape + tail)
        bufs = [(pinned(np.int32), pinned(np.float32, (3,)), pinned(np.uint8, (3,))) for _ in range(2)]
        fb = lib.FrameBatch([ds, twin], [p, p2], [b[0][0] for b in bufs], None, [b[1][0] for b in bufs], [b[2][0] for b in bufs])
        o2 = lib.DeviceScene(flat).render(p2)
        for rep in range(2):          # twice: the second call's quadrant lists are cut by the first call's cost map (16+ samples; SRT_HEAVY_STEPS)
            for b in bufs:
                b[0][1][...] = -9
            fb.render()
            ds.sync(); twin.sync()
            for b, want in zip(bufs, (o, o2)):
                ok = ok and np.array_equal(b[0][1], want["hit_id"]) and np.array_equal(bits(b[1][1]), bits(want["rgb_linear"])) and np.array_equal(b[2][1], want["rgb8"])
        for b in bufs:
            for ptr, _ in b:
                lib.load().srt_host_free(ptr)
        print(f"seed {seed:3d}: {len(flat.names)} objects, {flat.tri_points.shape[0]:5d} triangles, {W}x{H}, L={L:2d}, {({k: v for k, v in kw.items() if k != 'ray_matrix'} or 'whole')}{' camera' if 'ray_matrix' in kw else ''}: "
              f"{ds.pipeline:45s} hits {int((c['hit_id'] >= 0).sum()):6d}  {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
    print(f"fuzz: {bad} mismatching configuration(s)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
