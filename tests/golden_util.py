"""Helpers to read the committed fixtures under tests/golden/ (see tests/golden/make_golden.py)."""
import hashlib
import os
import re

import numpy as np

from simple_raytracer_amd import abi
import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SCENES = ["cube", "sphere", "cubes4_a0", "cubes4_a40", "spheres6", "cube_ground", "ground_bunny", "texquad"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_kat():
    return np.load(os.path.join(GOLDEN, "kat.npz"))


def load_mesh(key):
    return scenes.mesh_points(np.load(os.path.join(GOLDEN, "meshes", key + ".npz")))


class GoldenScene:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, f"scene_{name}.npz"))
        self.flat = abi.FlatScene.from_npz_dict(self.z)
        self.light = self.z["light"]
        rj = str(self.z["recipe"])
        self.recipe = scenes.Recipe.from_json(rj) if rj else None
        self.renders = sorted({(int(m.group(1)), int(m.group(2)), int(m.group(3)))
                               for m in (re.match(r"(\d+)x(\d+)_L(\d+)_hit_id$", k) for k in self.z.files) if m})

    def out(self, W, H, L, key):
        k = f"{W}x{H}_L{L}_{key}"
        return self.z[k] if k in self.z.files else None

    def params(self, W, H, L, **kw):
        return abi.make_params(W, H, abi.light_staircase(self.light, L), **kw)


def all_renders(max_pixels=None):
    out = []
    for s in SCENES:
        g = GoldenScene(s)
        for (W, H, L) in g.renders:
            if max_pixels is None or W * H <= max_pixels:
                out.append((s, W, H, L))
    return out
