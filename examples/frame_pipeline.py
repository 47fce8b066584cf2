#!/usr/bin/env python3
"""Where a drop-in frame goes, and how much of it the host mirror takes back.

The reference rebuilds everything per frame (simple_raytracer.cpp:534-618 -- load, transform into view space,
createBoundingHierarchy, render, draw), so with the HIP path the host stages ARE the frame.  K3 scene (bunny + ground slab,
1920x1080), a small orbit, four ways through the host mirror and the C ABI:

  stages     every stage timed on its own, a device scene created and destroyed per frame (what round 1 shipped)
  renderer   srt_host::Renderer: one device scene for the whole orbit -- srt_scene_update into the existing allocations through
             pinned staging, srt_render_async into a pinned frame buffer (no hipMalloc / hipFree / pageable copy per frame)
  pipelined  the same, with frame n + 1's transform + createBoundingHierarchy on a second host thread while frame n is
             flattened, uploaded, rendered and collected (EXACT: every frame is the reference's frame)
  camera     camera mode (EXTENSION, srt_params.ray_matrix): the scene stays in world space, hierarchies are built once, each frame
             passes the viewMatrix -- different rounding, pinned by the oracle run in the same mode

    python examples/frame_pipeline.py [--frames 12] [--width 1920] [--height 1080]

Needs a GPU.  The mesh comes from the committed fixture tests/golden/meshes/bunny.npz, so the OBJ parse is not in the table."""
import argparse, ctypes as C, os, sys, threading, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simple_raytracer_amd import abi, host, lib      # noqa: E402
import golden_util as gu                             # noqa: E402
import scenes                                        # noqa: E402

T = host.Transformation
LIGHT = [300.0, -600.0, -100.0, 1.0]


def build_frame(f, bunny, cube, timing=None):
    """The per-frame scene script: new ObjectManager, transforms (the bunny turns a degree per frame), hierarchies."""
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
    if timing is not None:
        for k, s in enumerate(("add_object", "transform", "createBoundingHierarchy")):
            timing[s].append((t[k + 1] - t[k]) * 1e3)
    return om


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--builders", type=lambda v: [int(x) for x in v.split(",")], default=[2, 3, 4], help="frames built concurrently in mode 3b, e.g. 2,3,4")
    ap.add_argument("--serial-builders", type=lambda v: [int(x) for x in v.split(",")], default=[6, 8, 12], help="frames built concurrently in mode 3c (one thread per build)")
    a = ap.parse_args()
    W, H, N = a.width, a.height, a.frames
    bunny, cube = gu.load_mesh("bunny"), gu.load_mesh("cube")
    HL, GL = host.load(), lib.load()

    # ---- 1. stage by stage, scene created and destroyed per frame ----------------------------------------------------------
    stages = ["add_object", "transform", "createBoundingHierarchy", "flatten", "srt_scene_create", "srt_render (sync, rgb8 to host)", "srt_scene_destroy"]
    acc = {s: [] for s in stages}
    rgb8 = np.empty((H, W, 3), np.uint8)
    for f in range(N + 1):
        tm = {s: [] for s in stages[:3]}
        om = build_frame(f, bunny, cube, tm)
        t = [time.perf_counter()]
        fh = HL.srth_flatten(om.om)
        d = abi.SceneDesc(); HL.srth_flat_desc(fh, C.byref(d))
        t.append(time.perf_counter())
        sh = C.c_void_p()
        rc = GL.srt_scene_create(0, C.byref(d), C.byref(sh)); assert rc == 0, rc
        t.append(time.perf_counter())
        p = abi.Params(); GL.srt_params_default(C.byref(p), W, H)
        light = np.array(LIGHT[:3], np.float32)
        p.n_lights = 1; p.light_pos = light.ctypes.data_as(C.POINTER(C.c_float))
        rc = GL.srt_render(sh, C.byref(p), None, None, None, rgb8.ctypes.data_as(C.POINTER(C.c_uint8)), None); assert rc == 0, rc
        t.append(time.perf_counter())
        GL.srt_scene_destroy(sh); HL.srth_flat_free(fh)
        t.append(time.perf_counter())
        if f == 0:
            continue                      # first frame: library load, HIP context
        for s in stages[:3]:
            acc[s].append(tm[s][0])
        for s, a0, a1 in zip(stages[3:], t[:-1], t[1:]):
            acc[s].append((a1 - a0) * 1e3)
    print(f"K3 scene {W}x{H}, {N} frames, median ms per frame (host mirror + C ABI)")
    print(" 1. stage by stage, device scene created and destroyed per frame:")
    tot = 0.0
    for s in stages:
        m = float(np.median(acc[s])); tot += m
        print(f"      {s:40s} {m:9.3f}")
    print(f"      {'sum':40s} {tot:9.3f}")

    # ---- 2. Renderer: one device scene for the orbit ---------------------------------------------------------------------------
    r = host.Renderer(0)
    per = []
    for f in range(N + 1):
        t0 = time.perf_counter()
        om = build_frame(f, bunny, cube)
        t1 = time.perf_counter()
        n = r.render(om, W, H, LIGHT, image=False)
        t2 = time.perf_counter()
        if f:
            per.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    b, c = np.median([x[0] for x in per]), np.median([x[1] for x in per])
    print(f" 2. Renderer (scene updated in place, pinned buffers): build {b:.3f} + drop-in call (flatten .. ImageData) {c:.3f} = {b + c:.3f} ms per frame")

    # ---- 3. pipelined: frame n + 1 is built on a second thread while frame n goes through the Renderer -------------------------
    nxt = {}
    def builder(f):
        nxt["om"] = build_frame(f, bunny, cube)
    om = build_frame(0, bunny, cube)
    t_render, t_join = [], []
    t0 = time.perf_counter()
    for f in range(N):
        th = threading.Thread(target=builder, args=(f + 1,)); th.start()      # ctypes calls release the GIL
        a0 = time.perf_counter()
        n = r.render(om, W, H, LIGHT, image=False)
        a1 = time.perf_counter()
        th.join()
        t_render.append((a1 - a0) * 1e3); t_join.append((time.perf_counter() - a1) * 1e3)
        om = nxt["om"]
    wall = (time.perf_counter() - t0) * 1e3 / N
    print(f" 3. pipelined (next frame's hierarchies built during this frame's flatten / upload / render / collect): {wall:.3f} ms per frame, exact "
          f"(drop-in call {np.median(t_render):.3f} ms, then {np.median(t_join):.3f} ms more until the next frame's scene is built)")

    # ---- 3b. several frames being built at any time: the hierarchy build has serial stretches (the top-level sorts), the builds of
    # different frames fill each other's gaps
    import concurrent.futures as cf
    for nb in a.builders:
        with cf.ThreadPoolExecutor(nb) as ex:
            fut = [ex.submit(build_frame, k, bunny, cube) for k in range(nb)]
            t0 = time.perf_counter()
            for f in range(N):
                om = fut[f % nb].result()
                fut[f % nb] = ex.submit(build_frame, f + nb, bunny, cube)
                n = r.render(om, W, H, LIGHT, image=False)
            wall2 = (time.perf_counter() - t0) * 1e3 / N
            for x in fut:
                x.result()
        print(f" 3b. pipelined, {nb} frames being built at any time: {wall2:.3f} ms per frame, exact")

    # ---- 3c. the same with every hierarchy build on its own thread alone (no pool tasks inside a build): more frames in the builders'
    # hands, each build slower, the cores never waiting for each other
    host.set_build_tasks(False)
    for nb in a.serial_builders:
        with cf.ThreadPoolExecutor(nb) as ex:
            fut = [ex.submit(build_frame, k, bunny, cube) for k in range(nb)]
            t0 = time.perf_counter()
            for f in range(3 * N):
                om = fut[f % nb].result()
                fut[f % nb] = ex.submit(build_frame, f + nb, bunny, cube)
                n = r.render(om, W, H, LIGHT, image=False)
            wall3 = (time.perf_counter() - t0) * 1e3 / (3 * N)
            for x in fut:
                x.result()
        print(f" 3c. pipelined, {nb} frames being built at any time, each build on one thread: {wall3:.3f} ms per frame, exact")
    host.set_build_tasks(True)

    # ---- 4. camera mode: world-space scene, hierarchies built once, one matrix per frame ---------------------------------------
    om = build_frame(0, bunny, cube)
    rc_ = host.Renderer(0)
    view = np.eye(4, dtype=np.float32).reshape(16)          # the K3 bench placement has the identity view
    rc_.render_from_camera(om, W, H, LIGHT, view, scene_changed=True, image=False)
    t0 = time.perf_counter()
    for f in range(N):
        view = scenes.orbit_view_matrix(T, 0.0, 0.5 * f, 0.0, 0.0)            # the camera turns half a degree per frame
        n = rc_.render_from_camera(om, W, H, LIGHT, view, image=False)
    wall_c = (time.perf_counter() - t0) * 1e3 / N
    print(f" 4. camera mode (EXTENSION: scene uploaded once, viewMatrix per frame; result incl. ImageData): {wall_c:.3f} ms per frame")


if __name__ == "__main__":
    main()
