#!/usr/bin/env python3
"""Where the K3 frame's time goes by part of the image: the bench scene (bunny over the ground slab) against the same frame with
only the slab and only the bunny, each at 1920x1080 with one light sample -- per-kernel HIP-event times of eager launches and
the wall time per frame with frames overlapped on 4 streams (srt_scene_share handles)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                     # noqa: E402
from simple_raytracer_amd import abi, host, lib      # noqa: E402
import golden_util as gu                               # noqa: E402
import scenes                                          # noqa: E402


def recipe(parts):
    MM = host.Transformation
    r = scenes.Recipe()
    if "bunny" in parts:
        r.load("./obj/stanford-bunny.obj", "bunny"); r.color("./obj/stanford-bunny.obj", (0.9, 0.9, 0.9))
        r.transform("./obj/stanford-bunny.obj", MM.scale(1500.0, 1500.0, 1500.0)); r.transform("./obj/stanford-bunny.obj", MM.rotx(MM.radians(180.0)))
        r.transform("./obj/stanford-bunny.obj", MM.translate(20.0, 170.0, 300.0)); r.bvh("./obj/stanford-bunny.obj")
    if "far" in parts:                                 # a cube no ray can reach: a frame of background tiles only
        r.load("cube.obj", "cube"); r.color("cube.obj", (0.2, 0.7, 0.3))
        r.transform("cube.obj", MM.translate(0.0, 50000.0, 350.0)); r.bvh("cube.obj")
    if "slab" in parts:
        r.load("cube.obj", "cube"); r.color("cube.obj", (0.2, 0.7, 0.3))
        r.transform("cube.obj", MM.scale(400.0, 10.0, 400.0)); r.transform("cube.obj", MM.translate(0.0, 130.0, 350.0)); r.bvh("cube.obj")
    r.light = (300.0, -600.0, -100.0)
    return r


def main():
    import torch
    W, H = 1920, 1080
    meshes = {"bunny": gu.load_mesh("bunny"), "cube": gu.load_mesh("cube")}
    for parts in (("bunny", "slab"), ("slab",), ("bunny",), ("far",)):
        r = recipe(parts)
        flat = host.build_flat_scene(r, {k: meshes[k] for k in r.meshes})
        p = abi.make_params(W, H, abi.light_staircase(np.array(r.light, np.float32), 1))
        hs = [lib.DeviceScene(flat)]; hs += [hs[0].share() for _ in range(3)]
        for _ in range(3):
            hs[0].render_device(p)
        hs[0].sync()
        for _ in range(20):
            hs[0].render_device(p)
        st = hs[0].sync()
        pc = abi.make_params(W, H, abi.light_staircase(np.array(r.light, np.float32), 1), flags=abi.SRT_FLAG_COUNT_WORK)
        hs[0].render_device(pc); sc = hs[0].sync()
        sc.setdefault('hit_rays', 0)
        streams = [torch.cuda.Stream() for _ in range(4)]
        pq = abi.make_params(W, H, abi.light_staircase(np.array(r.light, np.float32), 1), flags=abi.SRT_FLAG_NO_TIMING)
        def frames(n):
            for f in range(n):
                hs[f % 4].render_device(pq, stream=streams[f % 4].cuda_stream)
        frames(8); torch.cuda.synchronize()
        t0 = time.perf_counter(); frames(200); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 200 * 1e3
        print(f"{'+'.join(parts):12s} hit px {sc['hit_rays']:8d}  tests {sc['node_tests'] + sc['tri_tests']:10d}   trace {st['ms_primary'] + st['ms_shadow']:.4f}  shade {st['ms_shade']:.4f} ms alone;"
              f"  {wall:.4f} ms per frame on 4 streams (eager)", flush=True)


if __name__ == "__main__":
    main()
