#!/usr/bin/env python3
"""bench.py -- Mrays/s (primary + shadow) of the HIP ray-trace path at 1920x1080 on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched by torch.distributed.run,
one rank per GPU over RCCL).  One "step" = one pass of the hot path over a batch of 36 frames (the reference's
main() renders a 36-frame orbit per run): closest-hit (+ shadow rays) and shading kernels over this rank's part
of every frame, then (N > 1) ONE gather of the framebuffer tiles to rank 0.  Prints ONE JSON line on rank 0.

The line carries, beside the throughput: `parity` (the last rendered frame against the committed reference golden /
the CPU oracle: BASELINE.json's metric names both halves), `roofline` (the bound is chosen from counters collected IN
THIS RUN: three short rocprofv3 --pmc passes of this same script as child processes, --no-pmc skips them) and
`cpu_baseline`.

Workload (config.workload): BASELINE.json configs[2], the configuration the north_star quotes its
target on: stanford-bunny over a ground slab, 1920x1080, 1 light sample; scene from
the committed fixture tests/golden/scene_ground_bunny.npz (flat scene exported from the compiled
reference), already resident in HBM when the timed region starts.

The oracle (oracle/) is used here ONLY for the cpu_baseline leg and is never on the measured GPU path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured streaming)
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles (32 lanes per cycle), so the chip's peak is
# 1024 SIMDs x 32 lane-operations per cycle (x 2 flop x 2.4 GHz = the guide's 157.3 TFLOP/s FP32 vector peak); measured on the GPU
# by tests/test_gpu_parity.py::test_valu_issue_rate_is_the_guides (independent v_fma_f32 streams, 8 waves per SIMD)
N_SIMD, SIMD_LANES, VALU_CYCLES_PER_WAVE_INST = 1024.0, 32.0, 2.0
NODE_BYTES, TRI_BYTES = 32, 36  # algorithmic bytes per slab test / Moller-Trumbore test
BLOCK_ROWS = 8                  # scanline block size for the multi-GPU block-cyclic split
BLOCK_COLS = 0                  # > 0: tiles of 8 x BLOCK_COLS pixels dealt in two dimensions (srt_params.block_cols); measured: no gain over whole-width blocks


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=36, help="frames rendered per step (the reference renders a 36-frame orbit per run)")
    ap.add_argument("--workload", default="ground_bunny", choices=["ground_bunny", "cube_ground", "main_nocats", "k4", "soup"])
    ap.add_argument("--tris", type=int, default=1000000, help="triangle count of the synthetic soup workload (BASELINE.json configs[4])")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--lights", type=int, default=1)
    ap.add_argument("--spp", type=int, default=1, help="EXTENSION (not in the reference): n^2 sub-pixel samples per pixel, one launch pair each "
                                                       "(BASELINE.json configs[4] names 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the frames of a step are spread over (each with its own scene handle and buffers, one copy of the scene's "
                         "records).  Default 4: one frame's shading overlaps the next frame's tracing.  The per-kernel durations of the "
                         "roofline block are measured on eager launches on ONE stream either way; profile with --streams 1 to see kernels alone")
    ap.add_argument("--block-rows", type=int, default=BLOCK_ROWS, help="rows per scanline block of the N-way split (a multiple of 8)")
    ap.add_argument("--batch-groups", type=int, default=1, help="with --batch: the batches of a step go to this many streams in turn")
    ap.add_argument("--batch-frames", type=int, default=0, help="with --batch: frames per srt_render_device_batch call (0 = as many as keep the intermediates in cache)")
    ap.add_argument("--batch", choices=["auto", "on", "off"], default="auto",
                    help="the frames of a step through srt_render_device_batch (one pair of launches); auto = when this rank owns a share of each frame")
    ap.add_argument("--no-graph", action="store_true", help="launch every frame eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured path); gloo = rehearsal of the N > 1 logic on a box with fewer "
                         "GPUs than ranks (ranks share devices, tiles are staged through host memory)")
    ap.add_argument("--emulate-split", default="", help="R/N: time rank R's share of an N-way scanline split on ONE GPU (no collective); "
                                                         "diagnostic for the strong-scaling ceiling, not a bench line")
    ap.add_argument("--frame-groups", type=int, default=1,
                    help="N > 1: 1 (default, what north_star asks for) = every frame is tiled over ALL ranks; F > 1 (opt-in hybrid) = the "
                         "frames of a step are dealt to F groups of ranks and only inside a group is a frame tiled -- that is frame-level "
                         "replication, which flatters strong scaling and cannot apply to single long frames (K4, K5)")
    ap.add_argument("--block-cols", type=int, default=BLOCK_COLS,
                    help="N > 1: width of the tiles the scanline blocks are cut into (tile (bx, by) -> rank (bx + by) mod N); 0 = whole-width "
                         "scanline blocks")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic / valu come from profiles/traffic.json, labelled)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity block")
    ap.add_argument("--variant", type=int, default=0, help="experimental kernel selector (srt_params.flags bits 8-15)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearsal of the N-rank plumbing WITHOUT a GPU (CPU tests): self-launch, rendezvous (gloo), block-cyclic ownership, the "
                         "gather, rank 0's assembly and the per-phase times -- the tiles carry a rank pattern instead of rendered pixels; no "
                         "throughput is reported (`value` null)")
    ap.add_argument("--no-soup", action="store_true",
                    help="skip the second, short measurement the default workload adds: the 1 M-triangle soup at the same resolution and split "
                         "(north_star asks for the 1 / 2 / 4 / 8-GPU curve on the soup; it is reported as `soup` beside the headline value)")
    return ap.parse_args()


_setup = None


def setup(args):
    """Process-wide initialisation, once: device, process group, native build."""
    global _setup
    if _setup is not None:
        return _setup
    import torch
    import torch.distributed as dist
    from simple_raytracer_amd import build, lib

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    # only one process compiles (ranks share the source tree); the others wait for it.  Under rocprofv3 nothing is compiled: the
    # profiler's preloaded library has initialised the GPU, and starting hipcc / g++ / make from here would be the exec hop this pool
    # forbids -- the artefacts must be fresh already (profiles/pmc_run.sh and tools/collect_profiles.sh build first)
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_SDK_TOOL_LIBRARIES") or os.environ.get("ROCP_TOOL_LIBRARIES"):
        stale = build.stale_artefacts()
        if stale:
            raise SystemExit(f"bench.py under the profiler will not compile: stale or missing {stale}; run `python -m simple_raytracer_amd.build` first")
    elif rank == 0:
        build.build_all()
    if world > 1:
        dist.barrier()
    lib.load()
    _setup = (rank, local_rank, world)
    return _setup


def measure(args):
    """One workload, measured as the contract says.  Returns the JSON object on rank 0, None elsewhere."""
    global BLOCK_ROWS
    BLOCK_ROWS = args.block_rows
    import torch
    import torch.distributed as dist
    from simple_raytracer_amd import abi, lib
    import golden_util as gu
    rank, local_rank, world = setup(args)
    W, H, L, B = args.width, args.height, args.lights, args.frames
    if args.workload == "soup":
        g = soup_workload(args.tris)
    else:
        g = gu.GoldenScene(args.workload)
    # Frames of a step go to S streams in turn (one scene handle each: own workspace, ONE copy of the records, srt_scene_share): one
    # frame's shading -- a short, latency-bound launch -- overlaps the next frame's tracing.  K3, N = 1: 0.147 ms per frame on one
    # stream, 0.137 on two, 0.135 on four.
    S = max(1, min(args.streams if args.streams > 0 else 4, B))
    # A rank that owns a QUARTER of every frame (or less) issues the frames of a step through srt_render_device_batch: one pair of
    # launches for all of them, which fills the chip where such a share does not (round 3, frame tables by value: a quarter 1.187 ms
    # per 36-frame step against 1.256 frame by frame on 4 streams, an eighth 0.635; a half 2.281 against 2.282 and whole frames 4.33
    # against 4.28: bigger shares fill the chip well enough, and frame by frame their shading overlaps the next trace).
    # One handle per frame of a batch, else one per stream.
    split_n = int(args.emulate_split.split("/")[1]) if args.emulate_split else world // max(1, args.frame_groups)
    n_mine = B // max(1, args.frame_groups)
    # (second half of round 3, batch launches with the same tile row of all frames in flight together: the fused pipeline (1..7 samples) is
    # ahead in batches from a HALF on -- 2.18 against 2.25 ms per step; whole frames 4.33 against 4.30 --, the 16+-sample pipeline from a quarter)
    batch_from = 2 if 1 <= L <= 7 else 4
    batch = (args.batch == "on" or (args.batch == "auto" and split_n >= batch_from)) and L >= 1 and args.variant == 0 and args.spp == 1 and n_mine > 1
    scenes_ = [lib.DeviceScene(g.flat, device=local_rank)]
    scenes_ += [scenes_[0].share() for _ in range((max(S, n_mine) if batch else S) - 1)]      # the frames of a step render ONE scene: one copy of its records
    scene = scenes_[0]
    lights = abi.light_staircase(g.light, L)
    from simple_raytracer_amd import tiling
    emu = [int(x) for x in args.emulate_split.split("/")] if args.emulate_split else None
    # frames of a step -> FG groups of ranks; scanline blocks of a frame -> the world / FG ranks of a group
    FG = max(1, args.frame_groups)
    if world % FG or B % FG:
        raise SystemExit(f"--frame-groups {FG} must divide --gpus {world} and --frames {B}")
    per_group = world // FG
    B_total, B = B, B // FG                      # B: frames THIS rank renders per step
    split_rank, split_world = (emu if emu else (rank % per_group, per_group))
    BC = args.block_cols if split_world > 1 else 0
    p = tiling.split_params(W, H, lights, split_rank, split_world, BLOCK_ROWS, BC, flags=(args.variant << 8) | (abi.SRT_FLAG_FRAMES_IN_FLIGHT if S > 1 else 0), spp=args.spp)
    rows, Wl = scene.rows(p), scene.cols(p)
    dev = torch.device("cuda", local_rank)
    NB = len(scenes_)
    hit = torch.empty((NB, rows, Wl), dtype=torch.int32, device=dev)
    tbuf = torch.empty((NB, rows, Wl), dtype=torch.float32, device=dev)
    lin = torch.empty((NB, rows, Wl, 3), dtype=torch.float32, device=dev)
    side = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else []
    # the 8-bit framebuffer tiles of the B frames of a step live in the gather object (padded to equal rows
    # on every rank) so that the kernels write straight into the buffer the collective sends
    SLOTS = 2 if world > 1 else 1        # double-buffered tiles: the gather of step s overlaps the rendering of step s+1
    gather = tiling.FrameGather(W, H, BLOCK_ROWS if per_group > 1 else H, rank, world, dev, frames=B_total,
                                stage_through_host=(args.backend == "gloo"), slots=SLOTS, frame_groups=FG, block_cols=BC)
    if emu:
        assert world == 1
        gather.tiles = [torch.zeros((B, rows, Wl, 3), dtype=torch.uint8, device=dev)]
        gather.tile = gather.tiles[0]
    stream = torch.cuda.current_stream().cuda_stream
    frame_bytes = gather.tile[0].numel()

    batches = None
    if batch:
        # Frames per batch: as many as keep the batch's intermediates (hit ids, t, linear colour: 20 bytes a pixel, reused by the
        # next batch) inside the Infinity Cache -- a batch traces all its frames, then shades them.  The batches of a step go
        # to --batch-groups streams in turn (own handles and buffers per stream), so one batch's shading overlaps the next's tracing.
        G = max(1, min(args.batch_groups, S, B))
        nb = args.batch_frames if args.batch_frames > 0 else max(1, min(B, (512 << 20) // max(1, rows * Wl * 23)))
        nb = min(nb, NB // G)
        starts = list(range(0, B, nb))
        def make(slot, i, a0):
            a1, h0 = min(B, a0 + nb), (i % G) * nb
            n = a1 - a0
            return lib.FrameBatch(scenes_[h0:h0 + n], [p] * n, [hit[h0 + j].data_ptr() for j in range(n)], [tbuf[h0 + j].data_ptr() for j in range(n)],
                                  [lin[h0 + j].data_ptr() for j in range(n)], [gather.tiles[slot].data_ptr() + f * frame_bytes for f in range(a0, a1)])
        batches = [[make(slot, i, a0) for i, a0 in enumerate(starts)] for slot in range(SLOTS)]

    def render_batches(slot):
        cur = torch.cuda.current_stream()
        bs = batches[slot]
        if G == 1:
            for b_ in bs:
                b_.render(cur.cuda_stream)
            return
        for k in range(min(G, len(bs))):
            side[k].wait_stream(cur)
        for i, b_ in enumerate(bs):
            b_.render(side[i % G].cuda_stream)
        for k in range(min(G, len(bs))):
            cur.wait_stream(side[k])

    if batch:
        for slot in range(SLOTS):                             # every slot twice: a handle's counter sets alternate per call, and the
            render_batches(slot); render_batches(slot)        # argument tables of either parity must exist before a capture
        torch.cuda.synchronize()
        if not scene.pipeline.endswith("(batched)"):          # the scene prefers another pipeline (a soup): frame by frame then
            batches, batch = None, False

    def render_frames(pp, slot=0, streams=True):
        """The B frames of a step.  With S > 1 frame f goes to stream f % S (fork from / join into the current stream,
        which is what a capturing graph records as parallel branches)."""
        cur = torch.cuda.current_stream()
        if batches is not None and streams:
            render_batches(slot)
            return
        use = side if (streams and S > 1) else []
        for st in use:
            st.wait_stream(cur)
        for f in range(B):
            k = f % S if use else 0
            st = use[k] if use else cur
            scenes_[k].render_device(pp, stream=st.cuda_stream, hit_id=hit[k].data_ptr(), t=tbuf[k].data_ptr(),
                                     rgb_linear=lin[k].data_ptr(), rgb8=gather.tiles[slot].data_ptr() + f * frame_bytes)
        for st in use:
            cur.wait_stream(st)

    # The B renders of a step are launch-bound when a rank owns 1/8 of a frame: capture them once into a hipGraph
    # (torch.cuda.CUDAGraph = HIP stream capture; the launches go through the C ABI on the capturing stream).
    graphs = None
    if not args.no_graph:        # (an odd number of renders per handle leaves the hit counters of replayed frames un-zeroed: only statistics nobody reads)
        p_quiet = tiling.split_params(W, H, lights, split_rank, split_world, BLOCK_ROWS, BC, spp=args.spp,
                                      flags=(args.variant << 8) | abi.SRT_FLAG_NO_TIMING | (abi.SRT_FLAG_FRAMES_IN_FLIGHT if S > 1 else 0))      # the frames of a step are in flight on S streams
        try:
            render_frames(p_quiet); torch.cuda.synchronize()          # allocate every workspace before capturing
            graphs = []
            for slot in range(SLOTS):
                gph = torch.cuda.CUDAGraph()
                # thread_local: RCCL's watchdog thread may query events while this thread captures
                with torch.cuda.graph(gph, capture_error_mode="thread_local"):
                    render_frames(p_quiet, slot)
                graphs.append(gph)
        except Exception as e:                                         # capture is an optimisation, not a requirement
            if rank == 0:
                print(f"bench: hipGraph capture unavailable ({e}); eager launches", file=sys.stderr)
            graphs = None
            torch.cuda.synchronize()
    graph = graphs

    def step(i):
        # one step = B frames (the reference's main() renders a 36-frame orbit per run, simple_raytracer.cpp:534):
        # every rank renders its scanline blocks of each frame, then ONE gather moves all B tiles to rank 0.  With
        # two tile slots the gather of step i runs on RCCL's stream while step i+1 renders into the other slot.
        slot = i % SLOTS
        gather.finish(slot)              # the gather that used this slot two steps ago (rank 0 de-interleaves it)
        if graphs is not None:
            graphs[slot].replay()
        else:
            render_frames(p, slot)
        gather.start(slot)

    def fence():
        gather.finish_all()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # what the TIMED region wrote (replayed hipGraph, frames overlapped on S streams or batched): frame 0's buffers -- handle 0 renders
    # frames 0, S, 2S, ... of a step, all the same picture -- copied out before anything else renders into them
    timed = None
    last_slot = (args.warmup + args.steps - 1) % SLOTS if (args.warmup + args.steps) else 0
    if rank == 0 and world == 1 and not emu and not args.no_parity:
        timed = (hit[0].cpu().numpy().copy(), tbuf[0].cpu().numpy().copy(), lin[0].cpu().numpy().copy(), gather.tiles[last_slot][0].cpu().numpy().copy())
    # the phases of a step one by one (N > 1: the SCALE records need them to be read)
    phases = phase_times(gather, (lambda slot: graphs[slot].replay()) if graphs is not None else (lambda slot: render_frames(p, slot)),
                         world, dev, args.backend) if world > 1 else None
    # N > 1: what rank 0 holds after the last gather must be the frame one GPU renders (every step renders the same frames)
    gathered_ok = None
    gathered_parity = None
    if world > 1 and rank == 0 and FG == 1 and gather.frame is not None:
        whole = lib.DeviceScene(g.flat, device=local_rank).render(abi.make_params(W, H, lights, spp=args.spp), want=("rgb8",))["rgb8"]
        got = gather.frame.cpu().numpy()
        gathered_ok = bool(all(np.array_equal(got[f], whole) for f in (0, got.shape[0] - 1)))
        # and against the compiled reference's own frame where the golden holds this size: only the 8-bit framebuffer travels, so that
        # is what can be compared at N > 1 (hit ids / t / linear colour of the same kernels are in the N = 1 line's parity block)
        ref8 = g.out(W, H, L, "rgb8") if hasattr(g, "out") and args.spp == 1 and g.out(W, H, L, "hit_id") is not None else None
        if ref8 is not None:
            d8 = np.abs(got[0].astype(np.int32) - ref8.astype(np.int32))
            gathered_parity = {"against": f"tests/golden/scene_{g.name}.npz (rgb8 of the compiled reference, {W}x{H}, {L} light sample(s)); frame 0 as rank 0 assembled it",
                               "rgb8_pixels_differing": int((d8.max(-1) > 0).sum()), "rgb8_max_LSB": int(d8.max()), "pixels": int(W * H)}
    # per-kernel durations: HIP events on the launch stream over B eager renders of the same frames (the events of
    # a captured graph cannot be read back), averaged by srt_sync
    for sc_ in scenes_:
        sc_.sync()
    render_frames(p, streams=False)
    torch.cuda.synchronize()
    st = scene.sync()
    pipeline = scene.pipeline                  # what the library launched for this frame (the counting launch below may take another route)
    rgb8 = gather.tiles[0][0]

    # the eager single-stream re-render of the same frame must be bitwise what the timed region (graph replay) wrote
    eager_equal = None
    if timed is not None:
        eager = (hit[0].cpu().numpy(), tbuf[0].cpu().numpy(), lin[0].cpu().numpy(), rgb8.cpu().numpy())
        eager_equal = bool(all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(timed, eager)))
    # ---- ray and work accounting (one extra untimed launch of the counting build) -----------------
    pc = abi.make_params(W, H, lights, block_rows=p.block_rows, block_first=p.block_first, block_stride=p.block_stride, block_cols=p.block_cols,
                         flags=abi.SRT_FLAG_COUNT_WORK | (args.variant << 8), spp=args.spp)
    scene.render_device(pc, stream=stream, hit_id=hit[0].data_ptr(), t=tbuf[0].data_ptr(), rgb_linear=lin[0].data_ptr(), rgb8=rgb8.data_ptr())
    torch.cuda.synchronize()
    sc = scene.sync()
    rays_rank = sc["primary_rays"] + sc["shadow_rays"]        # of one of this rank's frames (its scanline blocks)
    # ---- parity of the frame the timed region rendered last (rank 0, whole frames only) -------------------------------
    parity = None
    if rank == 0 and world == 1 and not emu and not args.no_parity:
        parity = parity_block(g, args, W, H, L, lights, *timed)
        parity["frame_compared"] = ("frame 0 of the last step of the TIMED region (" + ("hipGraph replay" if graph is not None else "eager launches") +
                                    f", {S} stream(s)), copied out before any other render")
        parity["eager_rerender_bitwise_equal"] = eager_equal
    if world > 1:
        rr = torch.tensor([rays_rank / FG, sc["primary_rays"] / FG, sc["shadow_rays"] / FG], dtype=torch.float64,        # summed over ranks: one whole frame
                          device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(rr)
        rays_total, prim_total, shad_total = [float(x) for x in rr.tolist()]
    else:
        rays_total, prim_total, shad_total = float(rays_rank), float(sc["primary_rays"]), float(sc["shadow_rays"])
    ms_step = dt / args.steps * 1e3
    value = rays_total * B_total / (dt / args.steps) / 1e6

    if rank == 0:
        # with --spp n^2 a frame is n^2 launch pairs: counts are per LAUNCH (averaged over the sub-frames), like the kernel times
        sc = dict(sc)
        for k in ("hit_rays", "node_tests_primary", "tri_tests_primary", "node_tests_shadow", "tri_tests_shadow"):
            sc[k] = sc[k] // args.spp
        if args.spp > 1:        # the event pair around the traversal spans all sub-frames: per launch = / spp (includes the small shade / accumulate launches in between)
            st = dict(st); st["ms_primary"] = st["ms_primary"] / args.spp
        pixels = int(sc["primary_rays"]) // args.spp      # pixels of the image this rank renders
        hits, miss = sc["hit_rays"], pixels - sc["hit_rays"]
        items = hits * L
        # algorithmic bytes per launch: 32 B per slab test + 36 B per Moller-Trumbore test of the kernel's own
        # traversal + the per-pixel / per-item records each kernel must read and write
        b_closest = NODE_BYTES * sc["node_tests_primary"] + TRI_BYTES * sc["tri_tests_primary"] + 8 * pixels + 15 * miss
        b_shadow = NODE_BYTES * sc["node_tests_shadow"] + TRI_BYTES * sc["tri_tests_shadow"] + 4 * pixels + 8 * hits + items // 8
        b_shade = 4 * pixels + (4 + 12 + 4 + 15) * hits + items // 8
        names = pipeline.split("+")                      # what the library launched, e.g. k_trace_nq+k_shade_tile
        if args.variant == 1:
            kern = {"k_closest_hit": dict(ms=st["ms_primary"], bytes=NODE_BYTES * sc["node_tests_primary"] + TRI_BYTES * sc["tri_tests_primary"] + 8 * pixels),
                    "k_shade": dict(ms=st["ms_shade"], bytes=NODE_BYTES * sc["node_tests_shadow"] + TRI_BYTES * sc["tri_tests_shadow"] + 12 * hits + 23 * pixels)}
        elif names[0] == "k_trace_nq":                   # closest hit + shadow rays in one launch
            kern = {"k_trace_nq": dict(ms=st["ms_primary"] + st["ms_shadow"], bytes=b_closest + b_shadow), "k_shade_tile": dict(ms=st["ms_shade"], bytes=b_shade)}
        else:
            kern = {names[0]: dict(ms=st["ms_primary"], bytes=b_closest)}
            if len(names) == 3:
                kern[names[1]] = dict(ms=st["ms_shadow"], bytes=b_shadow)
            kern["k_shade_tile"] = dict(ms=st["ms_shade"], bytes=b_shade)
        dom = max(kern, key=lambda k: kern[k]["ms"])
        dom_ms = kern[dom]["ms"]
        effective = kern[dom]["bytes"] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # ---- counters of THIS run: three rocprofv3 --pmc passes of this script as child processes -------------------
        pmc = None
        if world == 1 and not args.no_pmc and not args.emulate_split:
            pmc = collect_pmc(sys.argv[1:], dom)
        if pmc is None and world == 1:
            pmc = committed_pmc(args.workload, dom, W, H, L)
        roof = roofline_block(dom, dom_ms, kern[dom]["bytes"], effective, pmc, g)
        pk = (pmc or {}).get("per_kernel") or {}
        if roof.get("valu") and pk and args.spp == 1 and all("SQ_INSTS_VALU" in v for v in pk.values()):
            # the same counters against the TIMED region: the frame's VALU wave-instructions (all kernels of the pipeline, per launch =
            # per frame) over the cycles a frame takes there, frames overlapped on S streams
            insts_f = sum(v["SQ_INSTS_VALU"] for v in pk.values())
            lanes_f = sum(v.get("SQ_THREAD_CYCLES_VALU", 0.0) for v in pk.values()) / insts_f if insts_f else 0.0
            cyc_f = (ms_step / B_total) * 1e-3 * roof["valu"]["clock_GHz"] * 1e9
            issue_f = insts_f * VALU_CYCLES_PER_WAVE_INST / (N_SIMD * cyc_f) if cyc_f else 0.0
            roof["timed_region"] = {"valu_issue_frac": round(issue_f, 4), "lanes_active_of_64": round(lanes_f, 1),
                                    "frac_of_lane_peak": round(issue_f * lanes_f / 64.0, 4), "kernels": sorted(pk),
                                    "note": "VALU wave-instructions of ALL kernels of a frame (counters of the kernels running alone) x 2 / (1024 SIMDs x the cycles a frame "
                                            "takes in the timed region): what the overlap of frames on streams adds to the single kernel's `frac`"}
        roof["kernel_ms_note"] = ("HIP events around eager launches on ONE stream (the kernel running alone, as in profiles/*_kernel_stats.csv and the --pmc "
                                  f"passes); the timed region overlaps the frames of a step on {S} stream(s), where a per-kernel duration is not defined")
        out = {
            "metric": "Mrays/sec (primary+shadow) at 1920x1080; max per-pixel |dRGB| vs CPU ref", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            **({"backend": "gloo (rehearsal, not a measurement of the RCCL path)"} if args.backend == "gloo" and world > 1 else {}),
            **({"gathered_frames_equal_one_gpu_render": gathered_ok} if gathered_ok is not None else {}),
            **({"parity": gathered_parity} if gathered_parity is not None else {}),
            **(phases if phases is not None else {}),
            "config": {"workload": f"{args.workload}: stanford-bunny (69,451 tris) over a ground slab, BVH + slab-AABB, "
                                   f"{W}x{H}, {L} light sample(s) [BASELINE.json configs[2]]" if args.workload == "ground_bunny"
                       else (f"{args.workload} {W}x{H} {L} light(s) [BASELINE.json configs[1]]" if args.workload == "cube_ground"
                             else f"main_nocats: the scene of the reference's main() (ground cube, bunny, 3 textured trees; the cats are a missing blob), "
                                  f"{W}x{H}, {L} light sample(s) [BASELINE.json configs[3] shape]" if args.workload == "main_nocats"
                             else f"k4: composite scene (ground cube, bunny, 3 textured trees, horse, house without its 'Plane'; 223,855 triangles, 8 textures), "
                                  f"{W}x{H}, {L} light sample(s) [BASELINE.json configs[3]]" if args.workload == "k4"
                             else f"soup: {args.tris} random triangles, {W}x{H}, {L} light sample(s), spp {args.spp} [BASELINE.json configs[4]]"),
                       "scene": "static, resident; no per-frame update (the frames of a step are renders of ONE scene already in HBM: a hot-path rate, not an orbit's frame rate)",
                       "scene_source": f"tests/golden/scene_{args.workload}.npz" if args.workload != "soup" else
                                       f"SplitMix64(0x5eed) soup, {args.tris} triangles in 4 objects, built by the host mirror (SURVEY.md s8d K5)",
                       "nodes": g.flat.n_nodes, "tris": g.flat.n_tris,
                       "parallelism": "1 GPU" if world == 1 else
                                      (f"the {B_total} frames of a step dealt to {FG} group(s) of {per_group} GPU(s); inside a group " +
                                       ((f"tiles of {BLOCK_ROWS} x {BC} pixels, tile (bx, by) -> rank (bx + by) mod {per_group}" if BC else
                                         f"scanline blocks of {BLOCK_ROWS} rows, block-cyclic") if per_group > 1 else "whole frames") +
                                       "; one RCCL gather of all tiles per step, overlapped with the next step's rendering"),
                       "frames_per_step": B_total, "ms_per_frame": round(ms_step / B_total, 5), "launch": ("hipGraph replay" if graph is not None else "eager") + (f", srt_render_device_batch (frames share launches), {len(batches[0])} batch(es) of {nb} on {G} stream(s)" if batches is not None else f", {S} stream(s)"),
                       "primary_rays_per_frame": prim_total, "shadow_rays_per_frame": shad_total},
            "roofline": roof,
            "kernels": {k: {"ms": round(v["ms"], 5), "algorithmic_bytes": v["bytes"]} for k, v in kern.items()},
        }
        out["config"]["pipeline"] = pipeline
        if parity is not None:
            out["parity"] = parity
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(g, W, H, L, lights)
            if L == 1:
                out["cpu_reference"] = cpu_reference(g, W, H)
        return out
    return None


def launch_ranks(args):
    """`python bench.py --gpus N` from a plain shell (no torchrun around it, WORLD_SIZE unset): this process -- which has not touched
    HIP or torch.cuda -- starts N fresh child processes of the same command line, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, rendezvous on 127.0.0.1), lets rank 0's JSON line through on stdout and returns the worst exit
    code.  A rank that dies takes the others with it (they would wait in a collective for ever)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst, alive = 0, list(procs)
    while alive:
        time.sleep(0.05)
        for pr in list(alive):
            rc = pr.poll()
            if rc is None:
                continue
            alive.remove(pr)
            if rc != 0:
                worst = worst or rc
                for other in alive:                              # exact children of this process
                    other.terminate()
    return worst


def dry_run(args):
    """--dry-run: the N-rank plumbing of measure() without a GPU -- process group (gloo), ownership tables, one equal-size gather per
    step into rank 0 (double-buffered as in the measured path), assembly, the per-phase clocks and the JSON line's N > 1 fields.
    The tiles carry a pattern (image row, image column, frame) that rank 0 checks after assembly.  Nothing is rendered and no
    throughput is claimed."""
    import torch
    import torch.distributed as dist
    from simple_raytracer_amd import tiling
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        dist.init_process_group("gloo")
    W, H, B = args.width, args.height, args.frames
    dev = torch.device("cpu")
    BC = args.block_cols if world > 1 else 0
    fg = tiling.FrameGather(W, H, args.block_rows if world > 1 else H, rank, world, dev, frames=B, slots=2 if world > 1 else 1, block_cols=BC)
    pm = torch.as_tensor(fg.pix_of[rank])                          # image pixel of every local pixel (-1 = padding)
    def fill(slot, step):
        y, x = (pm // W).clamp(min=0), (pm % W).clamp(min=0)
        for f in range(B):
            t = fg.tiles[slot][f, : fg.rows, : fg.cols]
            t[..., 0] = (y % 251).to(torch.uint8); t[..., 1] = (x % 241).to(torch.uint8); t[..., 2] = (f * 7 + step) % 256
    t_render = t_gather = t_asm = 0.0
    for step in range(args.warmup + args.steps):
        slot = step % len(fg.tiles)
        fg.finish(slot)
        t0 = time.perf_counter(); fill(slot, step); t_render += time.perf_counter() - t0
        fg.start(slot)
    frame = fg.finish_all()
    ph = phase_times(fg, lambda slot: fill(slot, 99), world, dev, "gloo", reps=2)
    ok = None
    if rank == 0:
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        got = fg.frame if world > 1 else fg.tiles[0][:, :H, :W]      # (one rank: its tile IS the frame)
        ok = bool(torch.equal(got[0, ..., 0], (yy % 251).to(torch.uint8)) and torch.equal(got[0, ..., 1], (xx % 241).to(torch.uint8))
                  and int(got[B - 1, 0, 0, 2]) == ((B - 1) * 7 + 99) % 256)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "Mrays/sec (primary+shadow) at 1920x1080; max per-pixel |dRGB| vs CPU ref", "value": None, "unit": "Mrays/s",
                          "dry_run": "no GPU: N-rank plumbing only (self-launch, gloo rendezvous, ownership, gather, assembly); tiles carry a pattern",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "assembled_frames_ok": ok, **ph}), flush=True)
    return 0 if (ok or rank != 0) else 1


def phase_times(gather, render_slot, world, dev, backend, reps=3):
    """What a step is made of, phase by phase and NOT overlapped (the timed region overlaps them: the gather of step s runs beside the
    rendering of step s + 1, and rank 0 assembles on a stream of its own): render (max over ranks), the collective, rank 0's assembly.
    Measured after the timed region with a barrier before every phase.  `gather_bytes` is what rank 0 receives per step."""
    import torch
    import torch.distributed as dist
    cuda = torch.device(dev).type == "cuda"
    def sync():
        if cuda:
            torch.cuda.synchronize()
    def barrier():
        sync()
        if world > 1:
            dist.barrier()
    acc = [0.0, 0.0, 0.0]
    for _ in range(reps):
        barrier(); t0 = time.perf_counter(); render_slot(0); sync(); acc[0] += time.perf_counter() - t0
        if world > 1:
            barrier(); t0 = time.perf_counter(); gather.start(0); gather.wait_collective(0); sync(); acc[1] += time.perf_counter() - t0
            barrier(); t0 = time.perf_counter(); gather.finish(0); gather.wait_assembly(); sync(); acc[2] += time.perf_counter() - t0
    vals = [a / reps * 1e3 for a in acc]
    if world > 1:
        tt = torch.tensor(vals, dtype=torch.float64, device=dev if (cuda and backend == "nccl") else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        vals = [float(x) for x in tt.tolist()]
    tile_bytes = gather.tiles[0].numel() * gather.tiles[0].element_size()
    return {"render_ms_max": round(vals[0], 4), "gather_ms": round(vals[1], 4) if world > 1 else 0.0, "assemble_ms": round(vals[2], 4) if world > 1 else 0.0,
            "gather_bytes": int(tile_bytes * (world - 1)) if world > 1 else 0,
            "phase_note": "per step, each phase alone behind a barrier (max over ranks); the timed region overlaps the gather and rank 0's assembly with the next step's rendering"}


def main():
    import copy
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.dry_run:
        sys.exit(dry_run(args))
    out = measure(args)
    rank, _, world = setup(args)
    if args.workload == "ground_bunny" and not args.no_soup and not args.emulate_split and args.spp == 1 and args.variant == 0:
        # north_star's scaling curve is quoted on the synthetic soup: same resolution, same split, a short run
        a2 = copy.copy(args)
        a2.workload, a2.frames, a2.steps, a2.warmup, a2.lights = "soup", 4, 3, 1, 1      # 4 frames: one per stream
        a2.no_cpu_baseline = a2.no_pmc = a2.no_parity = True
        o2 = measure(a2)
        if rank == 0:
            out["soup"] = {k: o2[k] for k in ("value", "unit", "n_gpus", "steps", "ms_per_step")}
            out["soup"].update({"workload": o2["config"]["workload"], "ms_per_frame": o2["config"]["ms_per_frame"], "pipeline": o2["config"]["pipeline"],
                                "kernels": o2["kernels"], "note": "secondary measurement for the N-GPU curve north_star quotes on the soup; `value` above stays the headline workload"})
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


class soup_workload:
    """BASELINE.json configs[4]: synthetic N-triangle soup, generator fully specified in tests/scenes.py, hierarchy built
    by the host-side C++ mirror of the reference's builder."""
    def __init__(self, n_tris):
        import scenes
        from simple_raytracer_amd import host
        self.recipe, meshes = scenes.soup(n_tris)
        self.flat = host.build_flat_scene(self.recipe, meshes)
        self.light = np.array(self.recipe.light, np.float32)
        self.recipe = None          # no reference replay for this workload in cpu_reference()


PMC_GROUPS = [["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], ["FETCH_SIZE"], ["WRITE_SIZE"],
              # what the kernel waits for (roofline.limiter): wave-cycle split, then the vector-memory path
              ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"],
              ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TA_BUSY_avr"],
              ["TCP_TCP_LATENCY_sum", "TCP_TA_TCP_STATE_READ_sum", "TD_TD_BUSY_sum", "TD_TC_STALL_sum"]]
PMC_REQUIRED = 3        # the first three groups carry the roofline; a failed limiter pass only drops `limiter`


def collect_pmc(argv, kernel):
    """Counters of the dominant kernel, collected NOW: rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py <same workload>,
    one child process per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass; no other trace domain is combined with
    --pmc).  Returns per-launch means, or None when rocprofv3 is not usable here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None
    keep, skip = [], False
    for a in argv:                                  # the child renders the same workload, briefly
        if skip:
            skip = False; continue
        if a in ("--steps", "--warmup", "--frames", "--streams"):
            skip = True; continue
        if a.startswith(("--steps=", "--warmup=", "--frames=", "--streams=")) or a in ("--no-cpu-baseline", "--no-pmc", "--no-parity", "--no-soup"):
            continue
        keep.append(a)
    child = ["python3", os.path.join(ROOT, "bench.py")] + keep + ["--steps", "2", "--warmup", "1", "--frames", "4", "--streams", "1",       # one stream: the counters of a kernel that runs alone
                                                              "--no-cpu-baseline", "--no-pmc", "--no-parity", "--no-soup"]
    vals, durs, per_kernel = {}, [], {}
    tmp = tempfile.mkdtemp(prefix="srt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    try:
        for i, grp in enumerate(PMC_GROUPS):
            d = os.path.join(tmp, f"pass{i}")
            r = subprocess.run([exe, "--kernel-trace", "--pmc", *grp, "--output-format", "csv", "-d", d, "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            if r.returncode:
                if i >= PMC_REQUIRED:
                    print(f"bench: rocprofv3 pass {grp} failed (rc {r.returncode}); roofline.limiter will lack these counters", file=sys.stderr)
                    continue
                print(f"bench: rocprofv3 pass {grp} failed (rc {r.returncode}); using profiles/traffic.json", file=sys.stderr)
                return None
            acc, acc_all = {}, {}
            for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
                for row in csv.DictReader(open(f)):
                    nm = row["Kernel_Name"].replace("void ", "")
                    if "<true" in nm:                                        # the counting build
                        continue
                    if nm.startswith(kernel):
                        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    if nm.startswith("k_") and row["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU"):
                        acc_all.setdefault((nm.split("<")[0].split("(")[0], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            for c, v in acc.items():
                vals[c] = sum(v) / len(v)
            for (kn, c), v in acc_all.items():
                per_kernel.setdefault(kn, {})[c] = sum(v) / len(v)
            if i == 0:
                for f in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")):
                    for row in csv.DictReader(open(f)):
                        nm = row["Kernel_Name"].replace("void ", "")
                        if nm.startswith(kernel) and "<true" not in nm:
                            durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    except Exception as e:          # profiling is evidence, not a requirement of the measurement
        print(f"bench: PMC collection failed ({e}); using profiles/traffic.json", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    need = ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE")
    if any(k not in vals for k in need):
        return None
    return {"source": f"rocprofv3 --pmc child passes of this run ({time.perf_counter() - t0:.0f} s, {len(durs)} launches)", "counters": vals,
            "kernel_ms_under_profiler": sum(durs) / len(durs) if durs else None, "per_kernel": per_kernel}


def committed_pmc(workload, kernel, W, H, L):
    """Fallback: the counters of the committed profiling passes (profiles/traffic.json), when this run cannot profile itself."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f).get(f"{workload}_{W}x{H}_L{L}", {})
    v = t.get("_valu", {}).get(kernel)
    if kernel not in t or not v:
        return None
    return {"source": "profiles/traffic.json (committed rocprofv3 passes of an earlier run, NOT this run)",
            "counters": {"SQ_INSTS_VALU": v["insts"], "GRBM_GUI_ACTIVE": v["cycles"] * 8, "SQ_THREAD_CYCLES_VALU": (v.get("lanes_active") or 0) * v["insts"]},
            "traffic_bytes": t[kernel], "kernel_ms_under_profiler": None}


def roofline_block(kernel, kernel_ms, algorithmic_bytes, effective_gbs, pmc, g):
    """The bound is chosen from counters, not assumed.  HBM: (2 x FETCH_SIZE + WRITE_SIZE) per launch (KiB; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for 16-B-per-lane reads on gfx950 -- an upper bound for the narrower ones) over the launch time
    against 8 TB/s.  VALU: CDNA4 has SIMD-32 -- a wave64 VALU instruction issues over 2 cycles -- so wave-instructions x 2 / (1024
    SIMDs x launch cycles) is the issue utilisation; times the active lanes per instruction / 64 it is the fraction of the
    lane-operation peak (1024 SIMDs x 32 lanes x clock = the guide's 157.3 TFLOP/s FP32 vector peak / 2 flop).  bound = "hbm" when the
    HBM fraction is at least 25 % and the larger of the two, else "valu".  `limiter` names what the dominant kernel actually waits for,
    from the vector-memory counters of the same passes (a kernel far below both peaks is bound by latency, not by a throughput).
    The algorithmic-bytes rate of SURVEY.md s8(d) is kept as a labelled EFFECTIVE figure: records come from L2 / Infinity Cache."""
    scene_mb = (g.flat.n_nodes * 32 + g.flat.n_tris * 96) / 1e6
    eff = {"effective_algorithmic_GBps": round(effective_gbs, 2), "algorithmic_bytes_per_launch": int(algorithmic_bytes),
           "effective_note": "32 B x slab tests + 36 B x triangle tests (+ per-pixel bytes) of the kernel's own traversal per launch time: NOT HBM traffic -- "
                             f"the {scene_mb:.1f} MB of records are served by L2 / Infinity Cache, so this rate can exceed the HBM peak"}
    if pmc is None or kernel_ms <= 0:
        return {"bound": "unmeasured", "kernel": kernel, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                "kernel_ms": round(kernel_ms, 5), **eff}
    c = pmc["counters"]
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0                         # summed over the 8 XCDs
    prof_ms = pmc.get("kernel_ms_under_profiler") or kernel_ms
    clock_ghz = cycles / (prof_ms * 1e6)
    insts = c["SQ_INSTS_VALU"]
    lanes = (c.get("SQ_THREAD_CYCLES_VALU", 0.0) / insts) if insts else 0.0
    issue = insts * VALU_CYCLES_PER_WAVE_INST / (N_SIMD * cycles) if cycles else 0.0
    valu_frac = issue * lanes / 64.0
    traffic = pmc.get("traffic_bytes")
    if traffic is None:
        traffic = 2 * 1024.0 * c["FETCH_SIZE"] + 1024.0 * c["WRITE_SIZE"]
    hbm_gbs = traffic / (prof_ms * 1e-3) / 1e9
    hbm_frac = hbm_gbs / HBM_PEAK_GBS
    common = {"kernel": kernel, "kernel_ms": round(kernel_ms, 5), "kernel_ms_under_profiler": round(prof_ms, 5), "traffic": int(traffic),
              "hbm_GBps": round(hbm_gbs, 1), "hbm_frac": round(hbm_frac, 4),
              "valu": {"wave_insts_per_launch": int(insts), "launch_cycles": int(cycles), "clock_GHz": round(clock_ghz, 3), "issue_frac": round(issue, 4),
                       "lanes_active_of_64": round(lanes, 1), "frac_of_lane_peak": round(valu_frac, 4),
                       "yardstick": "SIMD-32: wave-instructions x 2 cycles / (1024 SIMDs x launch cycles) x active lanes / 64"},
              "limiter": limiter_block(c, cycles), "source": pmc["source"], **eff}
    if hbm_frac >= 0.25 and hbm_frac >= valu_frac:
        return {"bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4), **common}
    peak = N_SIMD * SIMD_LANES * clock_ghz / 1e3                # T lane-operations / s at the clock the launch ran at
    return {"bound": "valu", "achieved": round(insts * lanes / (prof_ms * 1e-3) / 1e12, 3), "peak": round(peak, 3), "unit": "Tlaneop/s",
            "frac": round(valu_frac, 4), **common}


def limiter_block(c, cycles):
    """What the dominant kernel waits for, from the SQ / TA / TCP counters of the same child passes (per launch means).  SQ_WAVE_CYCLES,
    SQ_WAIT_ANY, SQ_ACTIVE_INST_* count quad-cycles summed over waves; TA / TCP counters are summed over the 256 CUs' instances."""
    if not c.get("SQ_WAVE_CYCLES"):
        return None
    wc = c["SQ_WAVE_CYCLES"]
    out = {"wave_cycles_waiting_frac": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4),
           "wave_cycles_issue_stalled_frac": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
           "wave_cycles_valu_frac": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4),
           "wave_cycles_scalar_frac": round(c.get("SQ_ACTIVE_INST_SCA", 0.0) / wc, 4),
           "wave_cycles_lds_frac": round(c.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, 4)}
    if c.get("SQ_INSTS_VMEM_RD"):
        out["vmem_read_wave_insts"] = int(c["SQ_INSTS_VMEM_RD"])
    cu_cycles = 256.0 * cycles
    if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") is not None and cu_cycles:
        out["l1_tag_lookups_per_cu_cycle"] = round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / cu_cycles, 4)
    if c.get("TA_BUSY_avr") is not None and cycles:
        out["ta_busy_frac"] = round(c["TA_BUSY_avr"] / cycles, 4)
    if c.get("TCP_PENDING_STALL_CYCLES_sum") is not None and cu_cycles:
        out["l1_pending_on_l2_stall_frac"] = round(c["TCP_PENDING_STALL_CYCLES_sum"] / cu_cycles, 4)
    if c.get("TCP_TCP_TA_DATA_STALL_CYCLES_sum") is not None and cu_cycles:
        out["l1_data_return_stall_frac"] = round(c["TCP_TCP_TA_DATA_STALL_CYCLES_sum"] / cu_cycles, 4)
    if c.get("TCP_TCP_LATENCY_sum") and c.get("TCP_TA_TCP_STATE_READ_sum"):
        out["l1_latency_cycles_per_wave_inst"] = round(c["TCP_TCP_LATENCY_sum"] / c["TCP_TA_TCP_STATE_READ_sum"], 1)
    if c.get("TD_TC_STALL_sum") is not None and cu_cycles:
        out["td_waiting_for_l1_data_frac"] = round(c["TD_TC_STALL_sum"] / cu_cycles, 4)
    w = out["wave_cycles_waiting_frac"]
    if w >= 0.5:
        out["name"] = "latency: waves parked behind s_waitcnt (vector-memory gathers of node / triangle records through the L1) most of their life"
    elif out["wave_cycles_valu_frac"] >= 0.5:
        out["name"] = "VALU issue"
    else:
        out["name"] = "mixed: issue and memory waits"
    return out


def parity_block(g, args, W, H, L, lights, hit, t, lin, rgb8):
    """BASELINE.json's metric names both halves: the frame the timed region rendered last, against (i) the committed reference
    golden of this workload at this size when there is one (hit ids and rgb8 in full, pre-tone-map floats on the stored
    subsample), else (ii) the CPU oracle on a band of scanlines of the same frame."""
    import golden_util as gu
    out = {"tolerance_max_abs_dRGB_linear": 1e-4}
    gold = g.out(W, H, L, "hit_id") if hasattr(g, "out") and args.spp == 1 else None
    if gold is not None:
        ref8 = g.out(W, H, L, "rgb8")
        d8 = np.abs(rgb8.astype(np.int32) - ref8.astype(np.int32))
        st = int(g.out(W, H, L, "sub_stride")) if g.out(W, H, L, "sub_stride") is not None else 1
        ref_lin = g.out(W, H, L, "sub_lin") if st > 1 else g.out(W, H, L, "lin").reshape(-1, 3)
        got_lin = lin.reshape(-1, 3)[::st]
        out.update({"against": f"tests/golden/scene_{g.name}.npz (outputs of the compiled reference, {W}x{H}, {L} light sample(s))",
                    "hit_id_mismatches": int((hit != gold).sum()), "t_bitwise_equal": gu.sha(t) == str(g.out(W, H, L, "sha_t")),
                    "max_abs_dRGB_linear": float(np.abs(got_lin - ref_lin).max()), "linear_samples_compared": int(ref_lin.shape[0]),
                    "rgb8_pixels_differing": int((d8.max(-1) > 0).sum()), "rgb8_max_LSB": int(d8.max()), "pixels": int(W * H)})
        return out
    bands = [b for b in getattr(g, "bands", []) if b[:3] == (W, H, L)] if args.spp == 1 else []
    if bands:           # a band of scanlines of this very frame that the compiled reference rendered (K4: a whole frame is hours of reference time)
        _, _, _, y0, y1 = bands[-1]
        sl = slice(y0, y1)
        ref8 = g.band_out(W, H, L, y0, y1, "rgb8")
        d8 = np.abs(rgb8[sl].astype(np.int32) - ref8.astype(np.int32))
        st = int(g.band_out(W, H, L, y0, y1, "sub_stride"))
        out.update({"against": f"tests/golden/scene_{g.name}.npz (the compiled reference's render of scanlines {y0}..{y1 - 1} of this {W}x{H} frame, {L} light samples)",
                    "hit_id_mismatches": int((hit[sl] != g.band_out(W, H, L, y0, y1, "hit_id")).sum()),
                    "t_bitwise_equal": gu.sha(np.ascontiguousarray(t[sl])) == str(g.band_out(W, H, L, y0, y1, "sha_t")),
                    "max_abs_dRGB_linear": float(np.abs(np.ascontiguousarray(lin[sl]).reshape(-1, 3)[::st] - g.band_out(W, H, L, y0, y1, "sub_lin")).max()),
                    "rgb8_pixels_differing": int((d8.max(-1) > 0).sum()), "rgb8_max_LSB": int(d8.max()), "pixels": int(W * (y1 - y0))})
        return out
    from oracle import pyoracle as po
    from simple_raytracer_amd import abi
    band = 8
    y0 = (H // 2) // band * band
    p = abi.make_params(W, H, lights, block_rows=band, block_first=y0 // band, block_stride=10 ** 6, spp=args.spp)
    c = po.render(g.flat, p)
    sl = slice(y0, y0 + band)
    d8 = np.abs(rgb8[sl].astype(np.int32) - c["rgb8"].astype(np.int32))
    out.update({"against": f"CPU oracle (oracle/srt_oracle.c) on scanlines {y0}..{y0 + band - 1} of the same frame",
                "hit_id_mismatches": int((hit[sl] != c["hit_id"]).sum()), "t_bitwise_equal": bool(np.array_equal(t[sl].view(np.uint32), c["t"].view(np.uint32))),
                "max_abs_dRGB_linear": float(np.abs(lin[sl] - c["rgb_linear"]).max()), "rgb8_pixels_differing": int((d8.max(-1) > 0).sum()),
                "rgb8_max_LSB": int(d8.max()), "pixels": int(W * band)})
    return out


def cpu_reference(g, W, H):
    """The reference's OWN hot path (oracle/_ref, compiled from its sources in the build container), one
    thread as the reference is, one frame of the same workload.  Extra context beside cpu_baseline."""
    from oracle import pyoracle as po
    if g.recipe is None:
        return None
    if not po.ref_available():          # said loudly: the compiled reference is git-ignored and has to travel with the snapshot
        print("bench: oracle/_ref/libsrt_ref.so did not travel to this box: no cpu_reference leg", file=sys.stderr)
        return {"absent": "oracle/_ref/libsrt_ref.so is not on this box (built from /root/reference by oracle/Makefile in the build container; git-ignored, shipped by gpurun)"}
    import golden_util as gu
    try:
        s = po.RefScene()
        g.recipe.replay(s, {k: gu.load_mesh(k) for k in g.recipe.meshes})
        flat = s.export()
        t0 = time.perf_counter()
        img, n = s.render(W, H, list(g.light) + [1.0])
        el = time.perf_counter() - t0
        hits = int((img.sum(-1) > 0).sum())
        return {"value": round((W * H + hits) / el / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "reference",
                "sample": f"1 frame {W}x{H} through the compiled reference's sendRaysAndIntersectPointsColors in {el:.2f} s"}
    except Exception as e:      # the prebuilt checker is optional on the GPU box
        return {"error": str(e)}


def cpu_baseline(g, W, H, L, lights):
    """The CPU restatement (oracle/, 'port') on the same workload, all host cores (OpenMP over rows), on a
    bounded sample: whole frames of the same 1920x1080 workload for about 10 s."""
    from oracle import pyoracle as po
    from simple_raytracer_amd import abi
    cores = po.oracle_lib().oracle_num_threads()
    # bounded sample: whole frames when a frame is cheap, else a band of scanline blocks in the middle of the frame
    band = max(1, H // 64)
    probe = abi.make_params(W, H, lights, block_rows=band, block_first=(H // band) // 2, block_stride=10 ** 6)
    t0 = time.perf_counter(); o = po.render(g.flat, probe, n_threads=cores); tp = time.perf_counter() - t0
    whole = tp * (H / band) < 1.0
    p = abi.make_params(W, H, lights) if whole else probe
    what = f"whole frames of the same {W}x{H} workload" if whole else f"passes over the middle {band} scanlines of the same {W}x{H} workload"
    o = po.render(g.flat, p, n_threads=cores)       # warm
    rays = o["stats"]["primary_rays"] + o["stats"]["shadow_rays"]
    n, t0 = 0, time.perf_counter()
    while True:
        po.render(g.flat, p, n_threads=cores); n += 1
        el = time.perf_counter() - t0
        if el > 8.0 or n >= 200:
            break
    t1 = time.perf_counter()
    o1 = po.render(g.flat, probe, n_threads=1)
    el1 = time.perf_counter() - t1
    rays1 = o1["stats"]["primary_rays"] + o1["stats"]["shadow_rays"]
    return {"value": round(rays * n / el / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{n} {what} in {el:.1f} s (OpenMP over rows)",
            "value_1_thread": round(rays1 / el1 / 1e6, 3)}


if __name__ == "__main__":
    main()
