#!/usr/bin/env python3
"""Occupancy timeline of the fused trace kernel (diagnostic build, -DSRT_DIAG): per wave and 8x8-tile quadrant the
absolute start / end stamps and the hardware slot it ran on.  Prints how many wave slots are busy over the kernel's
duration and the sum of wave-cycles.  Usage (GPU box): python profiles/diag_timeline.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simple_raytracer_amd import lib, build
lib.LIB_PATH = build.build_diag()
import golden_util as gu
g = gu.GoldenScene("ground_bunny")
ds = lib.DeviceScene(g.flat)
W, H = 1920, 1080
n_waves = ((W + 7) // 8) * ((H + 7) // 8) * 4
for variant in (0,):
    p = g.params(W, H, 1, flags=variant << 8)
    for _ in range(3):
        o = ds.render(p)
    d = o["rgb_linear"].reshape(-1).view(np.uint64)[: n_waves * 8].reshape(n_waves, 8)
    ok = (d[:, 6] > 0) & (d[:, 7] > d[:, 6])          # waves of background tiles leave no record (they end at the launch-time barrier)
    d = d[ok]
    closest = d[:, 0].astype(np.float64)
    hw = d[:, 5]
    k0 = d[:, 6].astype(np.int64); k1 = d[:, 7].astype(np.int64)
    cu_ = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64); se_ = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
    xcc_ = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    slot_ = (xcc_ * 8 + se_) * 16 + cu_
    # clocks are only comparable inside one CU: align every CU to its own first stamp, then look at the average CU
    span = 0
    for c in np.unique(slot_):
        m = slot_ == c
        base = k0[m].min(); k0[m] -= base; k1[m] -= base
        span = max(span, int(k1[m].max()))
    n_cu = len(np.unique(slot_))
    dur = (k1 - k0).astype(np.float64)
    print(f"variant {variant}: kernel {o['stats']['ms_primary']*1e3 + o['stats']['ms_shadow']*1e3:.1f} us (stamped build); longest CU span {span} ticks; "
          f"sum of wave durations {dur.sum()/1e6:.1f} M ticks; closest-hit part {closest.sum()/1e6:.1f} M")
    print(f"   duration percentiles (ticks): 50% {np.percentile(dur,50):.0f}  90% {np.percentile(dur,90):.0f}  99% {np.percentile(dur,99):.0f}  max {dur.max():.0f}")
    edges = np.linspace(0, span, 21)
    busy = []
    for a, b in zip(edges[:-1], edges[1:]):
        ov = np.clip(np.minimum(k1, b) - np.maximum(k0, a), 0, None)
        busy.append(ov.sum() / (b - a) / n_cu)
    print("   busy waves per CU (of 24 slots) in each 5% slice of the span: " + " ".join(f"{x:.1f}" for x in busy))
    cu = (hw >> np.uint64(8)) & np.uint64(0xf); se = (hw >> np.uint64(13)) & np.uint64(0x7); xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
    slot = (xcc.astype(np.int64) * 8 + se.astype(np.int64)) * 16 + cu.astype(np.int64)
    per_cu = np.bincount(slot, weights=dur)
    per_cu = per_cu[per_cu > 0]
    print(f"   CUs seen {len(per_cu)}; wave-ticks per CU: min {per_cu.min()/1e6:.2f} M  mean {per_cu.mean()/1e6:.2f} M  max {per_cu.max()/1e6:.2f} M")
