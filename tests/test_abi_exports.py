"""CPU-only: the C-ABI library builds, loads, exports every symbol include/srt.h declares, and its
host-side helpers (no device work) agree with the oracle.  No compute calls without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from simple_raytracer_amd import abi, build, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    build.build_all()
    return lib.load()


def test_header_symbols_are_exported(L):
    hdr = open(os.path.join(ROOT, "include", "srt.h")).read()
    declared = set(re.findall(r"\b(srt_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(lib.ABI_SYMBOLS), declared ^ set(lib.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.srt_abi_version() == 3


def test_struct_sizes_match_header(L):
    # the C compiler's layout of the PODs == the ctypes mirror (checked through srt_params_default)
    p = abi.Params()
    L.srt_params_default(C.byref(p), 600, 400)
    assert (p.width, p.height, p.block_rows, p.block_first, p.block_stride) == (600, 400, 400, 0, 1)
    assert p.focal == 400.0 and p.n_lights == 1 and p.shadow_div == 5.0 and p.reinhard == 0.5
    assert abs(p.gamma - 1.1) < 1e-7 and tuple(p.background)[:3] == (173, 216, 230) and p.spp == 1 and p.flags == 0


def test_flag_and_error_constants_match_the_header():
    """abi.py restates the header's enums by hand: every SRT_FLAG_* / SRT_ERR_* it names must have the header's value."""
    import re
    text = open(os.path.join(ROOT, "include", "srt.h")).read()
    found = {}
    for name, expr in re.findall(r"\b(SRT_(?:FLAG|ERR)_[A-Z_]+|SRT_OK)\s*=\s*([^,/\n]+)", text):
        e = expr.strip().replace("u", "")
        found[name] = eval(e, {"__builtins__": {}})
    assert {"SRT_FLAG_SMOOTH_NORMALS", "SRT_FLAG_COUNT_WORK", "SRT_FLAG_NO_TIMING", "SRT_FLAG_FRAMES_IN_FLIGHT"} <= set(found)
    for name, value in found.items():
        if hasattr(abi, name):
            assert getattr(abi, name) == value, name


def test_light_staircase_and_rows(L, oracle):
    base = np.array([500.0, -300.0, -200.0], np.float32)
    out = np.empty((64, 3), np.float32)
    L.srt_light_staircase(base.ctypes.data_as(C.POINTER(C.c_float)), 64, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(out.view(np.uint32), oracle.light_staircase(base, 64).view(np.uint32))
    for (H, rows, first, stride) in [(1080, 1080, 0, 1), (1080, 16, 3, 8), (37, 5, 1, 2), (10, 16, 0, 1), (10, 3, 5, 8)]:
        p = abi.make_params(64, H, [base], block_rows=rows, block_first=first, block_stride=stride)
        assert L.srt_rows_owned(C.byref(p)) == len(abi.rows_owned(H, rows, first, stride))
        assert L.srt_rows_owned(C.byref(p)) == oracle.oracle_lib().oracle_rows_owned(C.byref(p))


def test_argument_errors_without_device_work(L):
    h = C.c_void_p()
    assert L.srt_scene_create(0, None, C.byref(h)) == 1            # SRT_ERR_ARG
    d = abi.SceneDesc()
    assert L.srt_scene_create(0, C.byref(d), C.byref(h)) == 1
    assert b"argument" in L.srt_strerror(1) and b"fallback" in L.srt_strerror(4)


def test_no_exception_crosses_the_c_abi(L):
    """the ABI never throws.  A host allocation failure inside srt_scene_create (forced by the test hook)
    comes back as SRT_ERR_OOM; the same call without the hook gets as far as the device check."""
    import golden_util as gu
    g = gu.GoldenScene("cube")
    d = g.flat.desc()
    h = C.c_void_p()
    L.srt_debug_fail_host_allocs(1)
    assert L.srt_scene_create(0, C.byref(d), C.byref(h)) == abi.SRT_ERR_OOM and not h.value
    assert b"memory" in L.srt_strerror(abi.SRT_ERR_OOM)
    L.srt_debug_fail_host_allocs(0)
    rc = L.srt_scene_create(0, C.byref(d), C.byref(h))
    assert rc in (abi.SRT_OK, abi.SRT_ERR_NO_GPU)
    if rc == abi.SRT_OK:
        L.srt_scene_destroy(h)
    # limits are checked before anything is allocated: 2^26 nodes do not fit a node-queue entry
    d2 = g.flat.desc(); d2.n_nodes = 1 << 26
    assert L.srt_scene_create(0, C.byref(d2), C.byref(h)) == abi.SRT_ERR_LIMIT


def test_product_does_not_link_the_oracle():
    """The product library must not route through oracle/ (parity claims depend on it)."""
    import subprocess
    out = subprocess.run(["ldd", lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "srt_ref" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "simple_raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "pyoracle" not in src and "liboracle" not in src and "oracle_render" not in src, f
