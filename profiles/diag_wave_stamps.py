#!/usr/bin/env python3
"""Per-wave cycle stamps of the closest-hit phase (diagnostic build, -DSRT_DIAG): where do the heaviest waves
of the K3 frame spend their cycles?  Usage (GPU box): python profiles/diag_wave_stamps.py"""
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
p = g.params(W, H, 1, flags=10 << 8)          # unfused closest-hit kernel
for _ in range(2):
    o = ds.render(p)
n_waves = ((W + 7) // 8) * ((H + 7) // 8) * 4
d = o["rgb_linear"].reshape(-1).view(np.uint64)[: n_waves * 8].reshape(n_waves, 8).astype(np.float64)
tot, steps, batches, test, commit, tri, pop, load = d.T
items = titems = np.zeros_like(tot)
print(f"waves {n_waves}; kernel {o['stats']['ms_primary']*1e3:.1f} us (stamped build)")
order = np.argsort(-tot)
def row(sel, name):
    print(f"{name:14s} n={len(sel):6d} total {tot[sel].mean():9.0f} cyc  steps {steps[sel].mean():6.1f} ({items[sel].mean():7.0f} pairs)  batches {batches[sel].mean():6.1f} ({titems[sel].mean():7.0f} pairs)"
          f"  test {test[sel].mean():8.0f}  commit+push {commit[sel].mean():8.0f}  tri {tri[sel].mean():8.0f}  other {(tot-test-commit-tri)[sel].mean():8.0f}")
row(order[:100], "top 100")
row(order[:1000], "top 1000")
row(order[:10000], "top 10000")
row(order, "all")
light = np.where(steps <= 1)[0]
row(light, "<=1 step")
row(np.where(steps == 2)[0], "2 steps")
print("per node step (top 1000):  test %.0f cyc, commit+push %.0f cyc;  per tri batch %.0f cyc" % (
    test[order[:1000]].sum() / steps[order[:1000]].sum(), commit[order[:1000]].sum() / steps[order[:1000]].sum(), tri[order[:1000]].sum() / batches[order[:1000]].sum()))
top = order[:1000]
print("per node step (top 1000): LDS pop %.0f cyc, node loads %.0f cyc, slab test (rest of 'test') %.0f cyc" % (
    pop[top].sum() / steps[top].sum(), load[top].sum() / steps[top].sum(), test[top].sum() / steps[top].sum()))
print("sum of wave cycles / 1e6: %.1f" % (tot.sum() / 1e6))
