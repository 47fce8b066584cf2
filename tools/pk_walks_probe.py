#!/usr/bin/env python3
"""Where the packet shadow kernel's time goes (GPU box): the shipped walk built with wave clocks (variant 58) on a frame and on a
scanline-block share of it -- wave slots busy, longest walk, walks by duration.  SRT_DIAG_COUNTERS=1 python tools/pk_walks_probe.py
[--workload k4 --width 3840 --height 2160 --lights 64 --split 4/8]"""
import argparse, os, sys
os.environ.setdefault("SRT_DIAG_COUNTERS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simple_raytracer_amd import abi, lib, tiling     # noqa: E402
import golden_util as gu                               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="k4"); ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--lights", type=int, default=64); ap.add_argument("--split", default="4/8"); ap.add_argument("--variant", type=int, default=58)
a = ap.parse_args()
g = gu.GoldenScene(a.workload)
ds = lib.DeviceScene(g.flat)
lights = abi.light_staircase(g.light, a.lights)
r, n = [int(x) for x in a.split.split("/")]
for name, p in [(f"whole frame, variant {v}", abi.make_params(a.width, a.height, lights, flags=v << 8)) for v in (57, a.variant)] + \
               [(f"share {r}/{n}, variant {v}", tiling.split_params(a.width, a.height, lights, r, n, 8, 0, flags=v << 8)) for v in (57, a.variant)]:
    for rep in range(3):
        o = ds.render(p, outputs=("hit_id",)) if "outputs" in lib.DeviceScene.render.__code__.co_varnames else ds.render(p)
    print(f"{name}: {ds.pipeline}: shadow kernel {o['stats']['ms_shadow']:.3f} ms", flush=True)
