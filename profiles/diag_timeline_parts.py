#!/usr/bin/env python3
"""diag_timeline.py for the parts of the K3 frame (tools/k3_parts_probe.py): how long do the waves of slab tiles live, and how many
wave slots per CU are busy while the launch lasts?  Diagnostic build (-DSRT_DIAG).  Usage (GPU box): python profiles/diag_timeline_parts.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from simple_raytracer_amd import abi, host, lib, build
lib.LIB_PATH = build.build_diag()
import golden_util as gu
import k3_parts_probe as parts
W, H = 1920, 1080
n_waves = ((W + 7) // 8) * ((H + 7) // 8) * 4
meshes = {"bunny": gu.load_mesh("bunny"), "cube": gu.load_mesh("cube")}
for which in (("bunny", "slab"), ("slab",), ("bunny",)):
    r = parts.recipe(which)
    flat = host.build_flat_scene(r, {k: meshes[k] for k in r.meshes})
    ds = lib.DeviceScene(flat)
    p = abi.make_params(W, H, abi.light_staircase(np.array(r.light, np.float32), 1))
    for _ in range(3):
        o = ds.render(p)
    d = o["rgb_linear"].reshape(-1).view(np.uint64)[: n_waves * 8].reshape(n_waves, 8)
    ok = (d[:, 6] > 0) & (d[:, 7] > d[:, 6])          # waves of background tiles leave no record
    d = d[ok]
    hw = d[:, 5]
    k0 = d[:, 6].astype(np.int64); k1 = d[:, 7].astype(np.int64)
    cu_ = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64); se_ = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
    xcc_ = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    slot_ = (xcc_ * 8 + se_) * 16 + cu_
    span = 0
    for c in np.unique(slot_):
        m = slot_ == c
        base = k0[m].min(); k0[m] -= base; k1[m] -= base
        span = max(span, int(k1[m].max()))
    n_cu = len(np.unique(slot_))
    dur = (k1 - k0).astype(np.float64)
    closest = d[:, 0].astype(np.float64)
    pre = d[:, 3].astype(np.float64); ch = d[:, 4].astype(np.float64); sh = dur - pre - ch
    print(f"   median ticks: launch .. barrier {np.percentile(pre, 50):.0f}   closest-hit phase {np.percentile(ch, 50):.0f} (its own record: {np.percentile(closest, 50):.0f})   shadow phase + exit {np.percentile(sh, 50):.0f}")
    print(f"{'+'.join(which)}: kernel {(o['stats']['ms_primary'] + o['stats']['ms_shadow']) * 1e3:.1f} us (stamped build); waves with a record {len(d)}; longest CU span {span} ticks; "
          f"sum of wave durations {dur.sum() / 1e6:.2f} M ticks (closest-hit part {closest.sum() / 1e6:.2f} M)")
    print(f"   wave duration percentiles (ticks): 10% {np.percentile(dur, 10):.0f}  50% {np.percentile(dur, 50):.0f}  90% {np.percentile(dur, 90):.0f}  99% {np.percentile(dur, 99):.0f}  max {dur.max():.0f};"
          f"  closest-hit part 50% {np.percentile(closest, 50):.0f}")
    edges = np.linspace(0, span, 21)
    busy = []
    for a, b in zip(edges[:-1], edges[1:]):
        ov = np.clip(np.minimum(k1, b) - np.maximum(k0, a), 0, None)
        busy.append(ov.sum() / (b - a) / n_cu)
    print("   busy recorded waves per CU (of 24 slots) in each 5% slice of the span: " + " ".join(f"{x:.1f}" for x in busy))
