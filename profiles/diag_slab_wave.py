#!/usr/bin/env python3
"""Where a wave of a ground-slab quadrant spends its closest-hit phase (diagnostic build, -DSRT_DIAG, unfused closest-hit kernel):
medians of the phase's own stamps over the waves that hit the slab.  Usage (GPU box): python profiles/diag_slab_wave.py"""
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
r = parts.recipe(("slab",))
flat = host.build_flat_scene(r, {k: meshes[k] for k in r.meshes})
ds = lib.DeviceScene(flat)
p = abi.make_params(W, H, abi.light_staircase(np.array(r.light, np.float32), 1), flags=10 << 8)      # unfused: the phase's full record survives
for _ in range(3):
    o = ds.render(p)
d = o["rgb_linear"].reshape(-1).view(np.uint64)[: n_waves * 8].reshape(n_waves, 8).astype(np.float64)
tot, steps, batches, test, commit, tri, pop, load = d.T
sel = steps >= 1
print(f"slab only, unfused closest-hit kernel {o['stats']['ms_primary'] * 1e3:.1f} us (stamped build); waves with node steps: {int(sel.sum())}")
m = lambda a: np.percentile(a[sel], 50)
print(f"  median ticks: phase {m(tot):.0f} = node steps {m(steps):.0f} x (pop {m(pop / np.maximum(steps, 1)):.0f} + node loads {m(load / np.maximum(steps, 1)):.0f} + slab test {m(test / np.maximum(steps, 1)):.0f} + commit/push {m(commit / np.maximum(steps, 1)):.0f})"
      f" + triangle batches {m(batches):.0f} x {m(tri / np.maximum(batches, 1)):.0f} + rest {m(tot - test - commit - tri):.0f}")
