#!/usr/bin/env python3
"""Where a drop-in frame goes: the reference rebuilds everything per frame (simple_raytracer.cpp:534-618 -- load,
transform into view space, createBoundingHierarchy, render, draw), so with the HIP path the host stages are the frame.
Times every stage of the K3 scene (bunny + ground slab, 1920x1080) through the host mirror and the C ABI:

    python examples/frame_pipeline.py [--frames 10] [--width 1920] [--height 1080]

Needs a GPU (the render stage calls libsrt_hip.so).  The mesh comes from the committed fixture
tests/golden/meshes/bunny.npz, so the OBJ parse is not part of the table."""
import argparse, ctypes as C, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simple_raytracer_amd import abi, host, lib      # noqa: E402
import golden_util as gu                             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    a = ap.parse_args()
    W, H = a.width, a.height
    bunny, cube = gu.load_mesh("bunny"), gu.load_mesh("cube")
    T = host.Transformation
    HL, GL = host.load(), lib.load()
    stages = ["add_object", "transform", "createBoundingHierarchy", "flatten", "srt_scene_create", "srt_render (sync, rgb8 to host)",
              "srt_scene_destroy", "drop-in call (flatten..ImageData)"]
    acc = {s: [] for s in stages}
    rgb8 = np.empty((H, W, 3), np.uint8)
    for f in range(a.frames + 1):
        t = [time.perf_counter()]
        om = host.ObjectManager()
        om.add_object("bunny", bunny); om.add_object("cube", cube)
        om.setColor("bunny", (0.9, 0.9, 0.9)); om.setColor("cube", (0.2, 0.7, 0.3))
        t.append(time.perf_counter())
        om.transformTriangles("bunny", T.scaleObj(1500.0, 1500.0, 1500.0))
        om.transformTriangles("bunny", T.rotateObjX(T.radians(180.0 + f)))
        om.transformTriangles("bunny", T.changeObjPosition(20.0, 170.0, 300.0))
        om.transformTriangles("cube", T.scaleObj(400.0, 10.0, 400.0))
        om.transformTriangles("cube", T.changeObjPosition(0.0, 130.0, 350.0))
        t.append(time.perf_counter())
        om.createBoundingHierarchy("bunny"); om.createBoundingHierarchy("cube")
        t.append(time.perf_counter())
        fh = HL.srth_flatten(om.om)
        d = abi.SceneDesc(); HL.srth_flat_desc(fh, C.byref(d))
        t.append(time.perf_counter())
        sh = C.c_void_p()
        rc = GL.srt_scene_create(0, C.byref(d), C.byref(sh)); assert rc == 0, rc
        t.append(time.perf_counter())
        p = abi.Params(); GL.srt_params_default(C.byref(p), W, H)
        light = np.array([300.0, -600.0, -100.0], np.float32)
        p.n_lights = 1; p.light_pos = light.ctypes.data_as(C.POINTER(C.c_float))
        rc = GL.srt_render(sh, C.byref(p), None, None, None, rgb8.ctypes.data_as(C.POINTER(C.c_uint8)), None); assert rc == 0, rc
        t.append(time.perf_counter())
        GL.srt_scene_destroy(sh); HL.srth_flat_free(fh)
        t.append(time.perf_counter())
        img, n = om.render(W, H, [300.0, -600.0, -100.0, 1.0])
        t.append(time.perf_counter())
        if f == 0:
            continue                      # first frame: library load, HIP context
        for s, a0, a1 in zip(stages, t[:-1], t[1:]):
            acc[s].append((a1 - a0) * 1e3)
    print(f"K3 scene {W}x{H}, {a.frames} frames, median ms per stage (host mirror + C ABI):")
    tot = 0.0
    for s in stages:
        m = float(np.median(acc[s]))
        if not s.startswith("drop-in"):
            tot += m
        print(f"  {s:40s} {m:9.3f}")
    print(f"  {'sum of the stages above':40s} {tot:9.3f}")


if __name__ == "__main__":
    main()
