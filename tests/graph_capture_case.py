"""Run by tests/test_gpu_graph_capture.py in its own process (torch initialises HIP first): the frames of a step captured into a
hipGraph -- frame by frame on forked streams, and as srt_render_device_batch calls -- and replayed; every replayed frame must be the
eager frame bit for bit, also for a batch that is first seen DURING a capture (no table may be made then: frame-by-frame launches)."""
import os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simple_raytracer_amd import abi, lib      # noqa: E402
import golden_util as gu                       # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    g = gu.GoldenScene("ground_bunny")
    W, H, B = 192, 108, 6
    first = lib.DeviceScene(g.flat)
    handles = [first] + [first.share() for _ in range(B - 1)]
    for L in (2, 17):                                          # the fused pipeline and the 16+-sample pipeline (packet shadow kernel, batched)
        params = []
        for f in range(B):
            light = g.light.copy(); light[0] += 30.0 * f
            params.append(abi.make_params(W, H, abi.light_staircase(light, L), block_rows=8, block_first=f % 3, block_stride=3,
                                          flags=abi.SRT_FLAG_NO_TIMING))
        rows = [handles[0].rows(p) for p in params]
        eager = []
        for f, p in enumerate(params):
            o = lib.DeviceScene(g.flat).render(p)
            eager.append((o["hit_id"], o["rgb8"], o["rgb_linear"]))
        def buffers():
            return ([torch.full((rows[f], W), -5, dtype=torch.int32, device=dev) for f in range(B)],
                    [torch.zeros((rows[f], W, 3), dtype=torch.uint8, device=dev) for f in range(B)],
                    [torch.zeros((rows[f], W, 3), dtype=torch.float32, device=dev) for f in range(B)])
        def check(hit, rgb8, lin, what):
            torch.cuda.synchronize()
            for f in range(B):
                assert np.array_equal(hit[f].cpu().numpy(), eager[f][0]), (what, L, f, "hit ids")
                assert np.array_equal(rgb8[f].cpu().numpy(), eager[f][1]), (what, L, f, "rgb8")
                assert np.array_equal(lin[f].cpu().numpy().view(np.uint32), eager[f][2].view(np.uint32)), (what, L, f, "linear")
                hit[f].fill_(-5); rgb8[f].zero_(); lin[f].zero_()
        # same-size frames only can share launches: frames with block_first 0 / 1 / 2 differ in rows -> three groups per batch
        hit, rgb8, lin = buffers()
        fb = lib.FrameBatch(handles, params, [x.data_ptr() for x in hit], None, [x.data_ptr() for x in lin], [x.data_ptr() for x in rgb8])
        cur = torch.cuda.current_stream()
        fb.render(cur.cuda_stream); fb.render(cur.cuda_stream)          # both counter sets: the tables exist
        check(hit, rgb8, lin, "eager batch")
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, capture_error_mode="thread_local"):
            fb.render(torch.cuda.current_stream().cuda_stream)
        for rep in range(3):
            gph.replay(); check(hit, rgb8, lin, f"captured batch, replay {rep}")
        # a batch first seen during a capture
        hit2, rgb82, lin2 = buffers()
        fb2 = lib.FrameBatch(handles, params, [x.data_ptr() for x in hit2], None, [x.data_ptr() for x in lin2], [x.data_ptr() for x in rgb82])
        gph2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph2, capture_error_mode="thread_local"):
            fb2.render(torch.cuda.current_stream().cuda_stream)
        for rep in range(2):
            gph2.replay(); check(hit2, rgb82, lin2, f"batch first seen while capturing, replay {rep}")
        # frame by frame on two forked streams inside one graph
        side = [torch.cuda.Stream(device=dev) for _ in range(2)]
        hit3, rgb83, lin3 = buffers()
        def frames():
            c = torch.cuda.current_stream()
            for st in side:
                st.wait_stream(c)
            for f in range(B):
                st = side[f % 2]
                handles[f].render_device(params[f], stream=st.cuda_stream, hit_id=hit3[f].data_ptr(), rgb_linear=lin3[f].data_ptr(), rgb8=rgb83[f].data_ptr())
            for st in side:
                c.wait_stream(st)
        frames(); torch.cuda.synchronize()
        gph3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph3, capture_error_mode="thread_local"):
            frames()
        for rep in range(2):
            gph3.replay(); check(hit3, rgb83, lin3, f"frames on forked streams, replay {rep}")
    print("graph capture case: ok")


if __name__ == "__main__":
    main()
