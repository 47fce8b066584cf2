"""Helpers to read the committed fixtures under tests/golden/ (see tests/golden/make_golden.py)."""
import hashlib
import os
import re

import numpy as np

from simple_raytracer_amd import abi
import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SCENES = ["cube", "sphere", "cubes4_a0", "cubes4_a40", "spheres6", "cube_ground", "ground_bunny", "texquad", "main_nocats", "k4"]


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
        self.light = self.z["light"]
        rj = str(self.z["recipe"])
        self.recipe = scenes.Recipe.from_json(rj) if rj else None
        self.compact = "scene_node_min" not in self.z.files
        self._flat = None if self.compact else abi.FlatScene.from_npz_dict(self.z)
        self.renders = sorted({(int(m.group(1)), int(m.group(2)), int(m.group(3)))
                               for m in (re.match(r"(\d+)x(\d+)_L(\d+)_hit_id$", k) for k in self.z.files) if m})

    @property
    def flat(self):
        """The reference's flat scene.  Big scenes are stored as recipe + sha256 of every array of the reference's export:
        they are rebuilt with the host-side mirror (C++, CPU) and must hash to the reference's arrays."""
        if self._flat is None:
            from simple_raytracer_amd import build, host
            build.build_host()
            f = host.build_flat_scene(self.recipe, {k: load_mesh(k) for k in self.recipe.meshes})
            assert f.names == [str(x) for x in self.z["scene_names"]], "object order differs from the reference's"
            for k in abi.FlatScene.ARRAYS:
                key = "sha_scene_" + k
                if key in self.z.files:
                    assert sha(getattr(f, k)) == str(self.z[key]), f"host mirror's {k} differs from the reference's export"
            self._flat = f
        return self._flat

    @property
    def bands(self):
        """(W, H, L, y0, y1) of the scanline-band outputs (frames too big for a whole reference render)."""
        return sorted({tuple(int(x) for x in m.groups())
                       for m in (re.match(r"band_(\d+)x(\d+)_L(\d+)_y(\d+)_(\d+)_hit_id$", k) for k in self.z.files) if m})

    def band_out(self, W, H, L, y0, y1, key):
        return self.z[f"band_{W}x{H}_L{L}_y{y0}_{y1}_{key}"]

    def band_params(self, W, H, L, y0, y1, **kw):
        """srt_params that render exactly rows [y0, y1): one scanline block (y0 must be a multiple of the band height)."""
        rows = y1 - y0
        assert y0 % rows == 0
        return abi.make_params(W, H, abi.light_staircase(self.light, L), block_rows=rows, block_first=y0 // rows, block_stride=10 ** 6, **kw)

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
