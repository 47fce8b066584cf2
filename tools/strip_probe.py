#!/usr/bin/env python3
"""Why does a GPU that owns an eighth of every frame not run at eight times the frame rate?  

Times `copies` identical shares of the K3 frame issued through srt_render_device_batch (their launches shared, so no per-launch
tail is in the number) for several ways of cutting the frame, and divides by the share's slab + triangle tests from the counting
build: ns per 1000 tests, to be compared with the whole frame's.

    python tools/strip_probe.py [--workload ground_bunny] [--width 1920 --height 1080] [--lights 1]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simple_raytracer_amd import abi, lib      # noqa: E402
import golden_util as gu                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ground_bunny")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--lights", type=int, default=1)
    ap.add_argument("--copies", type=int, default=9)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--only", default="", help="run only the cuts whose label contains this")
    a = ap.parse_args()
    g = gu.GoldenScene(a.workload)
    W, H, L = a.width, a.height, a.lights
    lights = abi.light_staircase(g.light, L)
    handles = [lib.DeviceScene(g.flat) for _ in range(a.copies)]

    def run(label, **kw):
        if a.only and a.only not in label:
            return
        p = abi.make_params(W, H, lights, **kw)
        pc = abi.make_params(W, H, lights, flags=abi.SRT_FLAG_COUNT_WORK, **kw)
        handles[0].render_device(pc); st = handles[0].sync()
        tests = st["node_tests"] + st["tri_tests"]
        out = []
        for n in (1, a.copies):
            fb = lib.FrameBatch(handles[:n], [p] * n)
            for _ in range(3):
                fb.render()
            for h in handles[:n]:
                h.sync()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                fb.render()
            for h in handles[:n]:
                h.sync()
            out.append((time.perf_counter() - t0) / a.reps * 1e3)
        per = out[1] / a.copies
        print(f"{label:34s} rows {st['rows']:5d} rays {st['primary_rays'] + st['shadow_rays']:9d} tests {tests:11d}   alone {out[0]:.4f} ms   "
              f"x{a.copies} shared launch: {per:.4f} ms per share, {per * 1e6 / (tests / 1e3):.3f} ns per 1000 tests", flush=True)

    run("whole frame")
    for rows in (8, 16, 32, 64):
        for rank in (0, 3):
            run(f"rows {rows:3d}, rank {rank}/8", block_rows=rows, block_first=rank, block_stride=8)
    for cols in (64, 128, 256):
        run(f"tiles 8 x {cols}, rank 3/8", block_rows=8, block_first=3, block_stride=8, block_cols=cols)
        run(f"tiles 32 x {cols}, rank 3/8", block_rows=32, block_first=3, block_stride=8, block_cols=cols)


if __name__ == "__main__":
    main()
