#!/usr/bin/env python3
"""bench.py -- Mrays/s (primary + shadow) of the HIP ray-trace path at 1920x1080 on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched by torch.distributed.run,
one rank per GPU over RCCL).  One "step" = one pass of the hot path over one frame: closest-hit kernel
+ shadow/shade kernel over this rank's scanline blocks, then (N > 1) the framebuffer gather to rank 0.
Prints ONE JSON line on rank 0.

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
NODE_BYTES, TRI_BYTES = 32, 36  # algorithmic bytes per slab test / Moller-Trumbore test
BLOCK_ROWS = 8                  # scanline block size for the multi-GPU block-cyclic split


def main():
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
                    help="HIP streams the frames of a step are spread over (each with its own scene handle and buffers).  Default: 1 "
                         "at N = 1 (kernels run alone, so their durations are the ones rocprofv3 reports), 4 at N > 1, where a rank's "
                         "share of a frame is too small to fill the chip and independent frames overlap their ramp-up and tails")
    ap.add_argument("--no-graph", action="store_true", help="launch every frame eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured path); gloo = rehearsal of the N > 1 logic on a box with fewer "
                         "GPUs than ranks (ranks share devices, tiles are staged through host memory)")
    ap.add_argument("--emulate-split", default="", help="R/N: time rank R's share of an N-way scanline split on ONE GPU (no collective); "
                                                         "diagnostic for the strong-scaling ceiling, not a bench line")
    ap.add_argument("--frame-groups", type=int, default=0,
                    help="N > 1: the frames of a step are dealt to this many groups of ranks, and inside a group every frame is split "
                         "by scanline blocks (1 = every frame split over all ranks).  Default: the largest divisor of N that divides "
                         "--frames (36 frames: 2, 4, 4 groups at N = 2, 4, 8), because a whole or half frame fills the chip better "
                         "than an eighth of one")
    ap.add_argument("--variant", type=int, default=0, help="experimental kernel selector (srt_params.flags bits 8-15)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from simple_raytracer_amd import abi, build, lib
    import golden_util as gu

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
    # only one process compiles (ranks share the source tree); the others wait for it
    if rank == 0:
        build.build_all()
    if world > 1:
        dist.barrier()
    lib.load()

    W, H, L, B = args.width, args.height, args.lights, args.frames
    if args.workload == "soup":
        g = soup_workload(args.tris)
    else:
        g = gu.GoldenScene(args.workload)
    S = max(1, min(args.streams if args.streams > 0 else (1 if (world == 1 and not args.emulate_split) else 4), B))
    scenes_ = [lib.DeviceScene(g.flat, device=local_rank) for _ in range(S)]     # one handle (workspace, counters) per stream
    scene = scenes_[0]
    lights = abi.light_staircase(g.light, L)
    from simple_raytracer_amd import tiling
    emu = [int(x) for x in args.emulate_split.split("/")] if args.emulate_split else None
    # frames of a step -> FG groups of ranks; scanline blocks of a frame -> the world / FG ranks of a group
    FG = args.frame_groups if args.frame_groups > 0 else max(f for f in range(1, world + 1) if world % f == 0 and B % f == 0)
    if world % FG or B % FG:
        raise SystemExit(f"--frame-groups {FG} must divide --gpus {world} and --frames {B}")
    per_group = world // FG
    B_total, B = B, B // FG                      # B: frames THIS rank renders per step
    split_rank, split_world = (emu if emu else (rank % per_group, per_group))
    p = tiling.split_params(W, H, lights, split_rank, split_world, BLOCK_ROWS, flags=args.variant << 8, spp=args.spp)
    rows = scene.rows(p)
    dev = torch.device("cuda", local_rank)
    hit = torch.empty((S, rows, W), dtype=torch.int32, device=dev)
    tbuf = torch.empty((S, rows, W), dtype=torch.float32, device=dev)
    lin = torch.empty((S, rows, W, 3), dtype=torch.float32, device=dev)
    side = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else []
    # the 8-bit framebuffer tiles of the B frames of a step live in the gather object (padded to equal rows
    # on every rank) so that the kernels write straight into the buffer the collective sends
    SLOTS = 2 if world > 1 else 1        # double-buffered tiles: the gather of step s overlaps the rendering of step s+1
    gather = tiling.FrameGather(W, H, BLOCK_ROWS if per_group > 1 else H, rank, world, dev, frames=B_total,
                                stage_through_host=(args.backend == "gloo"), slots=SLOTS, frame_groups=FG)
    if emu:
        assert world == 1
        gather.tiles = [torch.zeros((B, rows, W, 3), dtype=torch.uint8, device=dev)]
        gather.tile = gather.tiles[0]
    stream = torch.cuda.current_stream().cuda_stream
    frame_bytes = gather.tile[0].numel()

    def render_frames(pp, slot=0, streams=True):
        """The B frames of a step.  With S > 1 frame f goes to stream f % S (fork from / join into the current stream,
        which is what a capturing graph records as parallel branches)."""
        cur = torch.cuda.current_stream()
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
        p_quiet = tiling.split_params(W, H, lights, split_rank, split_world, BLOCK_ROWS, flags=(args.variant << 8) | abi.SRT_FLAG_NO_TIMING, spp=args.spp)
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
    # per-kernel durations: HIP events on the launch stream over B eager renders of the same frames (the events of
    # a captured graph cannot be read back), averaged by srt_sync
    for sc_ in scenes_:
        sc_.sync()
    render_frames(p, streams=False)
    torch.cuda.synchronize()
    st = scene.sync()
    rgb8 = gather.tiles[0][0]

    # ---- ray and work accounting (one extra untimed launch of the counting build) -----------------
    pc = abi.make_params(W, H, lights, block_rows=p.block_rows, block_first=p.block_first, block_stride=p.block_stride,
                         flags=abi.SRT_FLAG_COUNT_WORK | (args.variant << 8), spp=args.spp)
    scene.render_device(pc, stream=stream, hit_id=hit[0].data_ptr(), t=tbuf[0].data_ptr(), rgb_linear=lin[0].data_ptr(), rgb8=rgb8.data_ptr())
    torch.cuda.synchronize()
    sc = scene.sync()
    rays_rank = sc["primary_rays"] + sc["shadow_rays"]        # of one of this rank's frames (its scanline blocks)
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
        pixels = W * rows
        hits, miss = sc["hit_rays"], pixels - sc["hit_rays"]
        items = hits * L
        # algorithmic bytes per launch: 32 B per slab test + 36 B per Moller-Trumbore test of the kernel's own
        # traversal + the per-pixel / per-item records each kernel must read and write
        kern = {
            "k_closest_hit_nq": dict(ms=st["ms_primary"],
                                    bytes=NODE_BYTES * sc["node_tests_primary"] + TRI_BYTES * sc["tri_tests_primary"]
                                    + 8 * pixels + 15 * miss),
            "k_shadow_nq": dict(ms=st["ms_shadow"],
                             bytes=NODE_BYTES * sc["node_tests_shadow"] + TRI_BYTES * sc["tri_tests_shadow"] + 4 * pixels + 8 * hits + items // 8),
            "k_shade_tile": dict(ms=st["ms_shade"], bytes=4 * pixels + (4 + 12 + 4 + 15) * hits + items // 8),
        }
        if (args.variant == 0 and L < 8) or args.variant in (11, 17):      # closest hit + shadow rays in one launch (shipped below 8 light samples)
            a, b = kern.pop("k_closest_hit_nq"), kern.pop("k_shadow_nq")
            kern = {"k_trace_nq": dict(ms=st["ms_primary"] + st["ms_shadow"], bytes=a["bytes"] + b["bytes"]), **kern}
        if args.variant == 1:
            kern = {"k_closest_hit": dict(ms=st["ms_primary"], bytes=NODE_BYTES * sc["node_tests_primary"] + TRI_BYTES * sc["tri_tests_primary"] + 8 * pixels),
                    "k_shade": dict(ms=st["ms_shade"], bytes=NODE_BYTES * sc["node_tests_shadow"] + TRI_BYTES * sc["tri_tests_shadow"] + 12 * hits + 23 * pixels)}
        dom = max(kern, key=lambda k: kern[k]["ms"])
        achieved = kern[dom]["bytes"] / (kern[dom]["ms"] * 1e-3) / 1e9 if kern[dom]["ms"] > 0 else 0.0
        out = {
            "metric": "Mrays/sec (primary+shadow) at 1920x1080", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            **({"backend": "gloo (rehearsal, not a measurement of the RCCL path)"} if args.backend == "gloo" and world > 1 else {}),
            "config": {"workload": f"{args.workload}: stanford-bunny (69,451 tris) over a ground slab, BVH + slab-AABB, "
                                   f"{W}x{H}, {L} light sample(s) [BASELINE.json configs[2]]" if args.workload == "ground_bunny"
                       else (f"{args.workload} {W}x{H} {L} light(s) [BASELINE.json configs[1]]" if args.workload == "cube_ground"
                             else f"main_nocats: the scene of the reference's main() (ground cube, bunny, 3 textured trees; the cats are a missing blob), "
                                  f"{W}x{H}, {L} light sample(s) [BASELINE.json configs[3] shape]" if args.workload == "main_nocats"
                             else f"k4: composite scene (ground cube, bunny, 3 textured trees, horse, house without its 'Plane'; 223,855 triangles, 8 textures), "
                                  f"{W}x{H}, {L} light sample(s) [BASELINE.json configs[3]]" if args.workload == "k4"
                             else f"soup: {args.tris} random triangles, {W}x{H}, {L} light sample(s), spp {args.spp} [BASELINE.json configs[4]]"),
                       "scene": f"tests/golden/scene_{args.workload}.npz" if args.workload != "soup" else
                                f"SplitMix64(0x5eed) soup, {args.tris} triangles in 4 objects, built by the host mirror (SURVEY.md s8d K5)",
                       "nodes": g.flat.n_nodes, "tris": g.flat.n_tris,
                       "parallelism": "1 GPU" if world == 1 else
                                      (f"the {B_total} frames of a step dealt to {FG} group(s) of {per_group} GPU(s); inside a group " +
                                       (f"scanline blocks of {BLOCK_ROWS} rows, block-cyclic" if per_group > 1 else "whole frames") +
                                       "; one RCCL gather of all tiles per step, overlapped with the next step's rendering"),
                       "frames_per_step": B_total, "ms_per_frame": round(ms_step / B_total, 5), "launch": ("hipGraph replay" if graph is not None else "eager") + f", {S} stream(s)",
                       "primary_rays_per_frame": prim_total, "shadow_rays_per_frame": shad_total},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": measured_traffic(args.workload, dom, W, H, L) if world == 1 else None,
                         "algorithmic_bytes_per_launch": kern[dom]["bytes"], "kernel_ms": round(kern[dom]["ms"], 5),
                         "valu": measured_valu(args.workload, dom, W, H, L) if world == 1 else None,
                         "note": "algorithmic bytes = 32 B x slab tests + 36 B x triangle tests (+ per-pixel output bytes) of the "
                                 f"kernel's own traversal; the scene ({(g.flat.n_nodes * 32 + g.flat.n_tris * 96) / 1e6:.1f} MB of node and triangle records) is "
                                 "L2 / Infinity-Cache resident, so this is an effective rate that can exceed the HBM peak"},
            "kernels": {k: {"ms": round(v["ms"], 5), "algorithmic_bytes": v["bytes"]} for k, v in kern.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(g, W, H, L, lights)
            if L == 1:
                out["cpu_reference"] = cpu_reference(g, W, H)
        print(json.dumps(out), flush=True)
    if world > 1:
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


def measured_traffic(workload, kernel, W, H, L):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json, written
    by profiles/pmc_summary.py from FETCH_SIZE / WRITE_SIZE collected in separate passes), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    return t.get(f"{workload}_{W}x{H}_L{L}", {}).get(kernel)


def measured_valu(workload, kernel, W, H, L):
    """VALU issue utilisation of `kernel` from the committed PMC passes: a wave64 VALU instruction holds its SIMD for 4
    cycles, the chip has 256 CUs x 4 SIMDs.  None when this workload was not profiled."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        v = json.load(f).get(f"{workload}_{W}x{H}_L{L}", {}).get("_valu", {}).get(kernel)
    if not v or not v.get("cycles"):
        return None
    return {"wave_insts_per_launch": v["insts"], "launch_cycles": v["cycles"], "lanes_active_of_64": v.get("lanes_active"),
            "issue_frac": round(v["insts"] * 4 / (1024 * v["cycles"]), 4), "source": "profiles/traffic.json (rocprofv3 --pmc SQ_INSTS_VALU, GRBM_GUI_ACTIVE)"}


def cpu_reference(g, W, H):
    """The reference's OWN hot path (oracle/_ref, compiled from its sources in the build container), one
    thread as the reference is, one frame of the same workload.  Extra context beside cpu_baseline."""
    from oracle import pyoracle as po
    if not po.ref_available() or g.recipe is None:
        return None
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
