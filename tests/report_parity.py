#!/usr/bin/env python3
"""Prints the measured parity of the HIP path against the reference goldens at 1920x1080 (run on the
GPU box): hit-id mismatches, t bits, max |dRGB| before tone mapping (vs the oracle), rgb8 LSB differences."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from simple_raytracer_amd import lib
from oracle import pyoracle as po

for name in ("cube_ground", "ground_bunny"):
    g = gu.GoldenScene(name)
    W, H, L = 1920, 1080, 1
    ds = lib.DeviceScene(g.flat)
    o = ds.render(g.params(W, H, L))
    c = po.render(g.flat, g.params(W, H, L))
    ref8 = g.out(W, H, L, "rgb8")
    d8 = np.abs(o["rgb8"].astype(int) - ref8.astype(int))
    print(f"{name} {W}x{H}: hit-id mismatches vs reference {int((o['hit_id'] != g.out(W, H, L, 'hit_id')).sum())}, "
          f"t bitwise equal {gu.sha(o['t']) == str(g.out(W, H, L, 'sha_t'))}, "
          f"max|dRGB_linear| vs oracle {np.abs(o['rgb_linear'] - c['rgb_linear']).max():.3e}, "
          f"linear floats differing {int((o['rgb_linear'].view(np.uint32) != c['rgb_linear'].view(np.uint32)).sum())} of {o['rgb_linear'].size}, "
          f"rgb8 pixels off by 1 LSB {int((d8.max(-1) > 0).sum())} (max {d8.max()}) of {int((o['hit_id'] >= 0).sum())} hit px")
