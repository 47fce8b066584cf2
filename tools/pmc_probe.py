#!/usr/bin/env python3
"""A lean target for rocprofv3 --pmc passes: one scene, N eager renders through the C ABI (srt_render, host buffers), no torch, no
graph, no parity -- a pass takes seconds instead of a bench.py run.  Build first (python -m simple_raytracer_amd.build): nothing is
compiled here.  Usage: pmc_probe.py --workload ground_bunny|k4|soup|cube_ground|main_nocats [--variant V] [--frames N] [--width W
--height H --lights L] [--flags F]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ground_bunny")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--lights", type=int, default=1); ap.add_argument("--tris", type=int, default=1000000)
    a = ap.parse_args()
    from simple_raytracer_amd import abi, lib
    if a.workload == "soup":
        import scenes
        from simple_raytracer_amd import host
        recipe, meshes = scenes.soup(a.tris)
        flat, light = host.build_flat_scene(recipe, meshes), np.array(recipe.light, np.float32)
    else:
        import golden_util as gu
        g = gu.GoldenScene(a.workload)
        flat, light = g.flat, g.light
    ds = lib.DeviceScene(flat)
    p = abi.make_params(a.width, a.height, abi.light_staircase(light, a.lights), flags=a.variant << 8)
    for i in range(a.frames):
        o = ds.render(p, want=("rgb8",))
    st = o["stats"]
    print(f"{a.workload} variant {a.variant}: {ds.pipeline}  ms primary/shadow/shade {st['ms_primary']:.4f} {st['ms_shadow']:.4f} {st['ms_shade']:.4f}")


if __name__ == "__main__":
    main()
