"""Run by tests/test_gpu_parity.py in its own process with SRT_HEAVY_STEPS set low (the library reads it once): batch calls of the
8+-sample pipeline rendered again and again on the same handles, so that from the second call on the quadrant lists are cut into a
heavy and an ordinary part by the previous call's cost map (srt_kernels.h) -- every frame must stay the single render bit for bit,
also when the lights move between calls (the map is a prediction, not a promise)."""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simple_raytracer_amd import abi, lib      # noqa: E402
import golden_util as gu                       # noqa: E402
import ctypes as C                             # noqa: E402


def main():
    L_ = lib.load()
    pinned = []
    def buf(shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = L_.srt_host_alloc(n); assert ptr
        pinned.append(ptr)
        return np.frombuffer((C.c_uint8 * n).from_address(ptr), dtype=dtype).reshape(shape)
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    for name, W, H in (("k4", 240, 136), ("ground_bunny", 192, 108), ("main_nocats", 160, 96)):
        g = gu.GoldenScene(name)
        first = lib.DeviceScene(g.flat)
        handles = [first] + [first.share() for _ in range(3)]
        for call in range(4):
            params = []
            for f in range(4):
                light = g.light.copy(); light[0] += 25.0 * f + (60.0 if call == 3 else 0.0)      # the last call moves the lights
                kw = dict(block_rows=8, block_first=1, block_stride=2) if name == "k4" else {}
                params.append(abi.make_params(W, H, abi.light_staircase(light, (16, 20, 64, 33)[f] if name != "k4" else 24), **kw))
            outs = [(buf((handles[0].rows(p), handles[0].cols(p)), np.int32), buf((handles[0].rows(p), handles[0].cols(p), 3), np.float32),
                     buf((handles[0].rows(p), handles[0].cols(p), 3), np.uint8)) for p in params]
            for o in outs:
                o[0][...] = -7
            lib.FrameBatch(handles, params, [o[0].ctypes.data for o in outs], None, [o[1].ctypes.data for o in outs], [o[2].ctypes.data for o in outs]).render()
            for h in handles:
                h.sync()
            for f, p in enumerate(params):
                one = lib.DeviceScene(g.flat)
                o = one.render(p)
                assert "k_shadow_pk" in one.pipeline, one.pipeline
                assert np.array_equal(outs[f][0], o["hit_id"]), (name, call, f, "hit ids")
                assert np.array_equal(bits(outs[f][1]), bits(o["rgb_linear"])), (name, call, f, "linear colour")
                assert np.array_equal(outs[f][2], o["rgb8"]), (name, call, f, "rgb8")
                one.close()
        for h in handles:
            h.close()
        # a frame alone on the device (no in-flight hint): the same lists from the handle's second render on; with the hint: plain order
        lone = lib.DeviceScene(g.flat)
        p16 = abi.make_params(W, H, abi.light_staircase(g.light, 20))
        want = lib.DeviceScene(g.flat).render(p16)
        for rep in range(3):
            got = lone.render(p16 if rep < 2 else abi.make_params(W, H, abi.light_staircase(g.light, 20), flags=abi.SRT_FLAG_FRAMES_IN_FLIGHT))
            assert np.array_equal(got["hit_id"], want["hit_id"]) and np.array_equal(bits(got["rgb_linear"]), bits(want["rgb_linear"])) and np.array_equal(got["rgb8"], want["rgb8"]), (name, rep)
        lone.close()
    for ptr in pinned:
        L_.srt_host_free(ptr)
    print("heavy list case: ok")


if __name__ == "__main__":
    main()
