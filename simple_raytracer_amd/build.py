"""Builds the product's native code in-tree with hipcc for gfx950 (no JIT cache: the .so files travel
with the repo snapshot to the GPU box).  `python -m simple_raytracer_amd.build` or build_all()."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_HIP = os.path.join(HERE, "libsrt_hip.so")

# -ffp-contract=off is part of the numerical contract: FMA contraction changes hit results
# (SURVEY.md H1).  Correctly rounded f32 divide / sqrt are hipcc defaults and must stay on.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, "srt_hip.hip")]
    deps = srcs + [os.path.join(CSRC, "srt_device.h"), os.path.join(CSRC, "srt_kernels.h"), os.path.join(CSRC, "srt_packet.h"), os.path.join(HERE, "..", "include", "srt.h")]
    if not force and not _stale(LIB_HIP, deps):
        return LIB_HIP
    cmd = [hipcc()] + HIPCC_FLAGS + ["-o", LIB_HIP] + srcs
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed for libsrt_hip.so")
    if verbose and r.stderr:
        print(r.stderr)
    return LIB_HIP


LIB_HOST = os.path.join(HERE, "libsrt_host.so")
HOST_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-pthread"]


def build_host(force=False, verbose=False):
    """Host-side C++ mirror of the reference's scene interface (g++); links the HIP library for the
    drop-in sendRaysAndIntersectPointsColors."""
    hdir = os.path.join(CSRC, "host")
    srcs = [os.path.join(hdir, "srt_host.cpp"), os.path.join(hdir, "srt_jpeg.cpp"), os.path.join(hdir, "srt_host_c.cpp")]
    deps = srcs + [os.path.join(hdir, "srt_host.h"), os.path.join(HERE, "..", "include", "srt.h"), LIB_HIP]
    if not force and not _stale(LIB_HOST, deps):
        return LIB_HOST
    cmd = ["g++"] + HOST_FLAGS + ["-o", LIB_HOST] + srcs + ["-L" + HERE, "-lsrt_hip", "-Wl,-rpath,$ORIGIN", "-lz"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("g++ failed for libsrt_host.so")
    if verbose and r.stderr:
        print(r.stderr)
    return LIB_HOST


EXAMPLE_ORBIT = os.path.join(HERE, "..", "examples", "orbit")


def build_examples(force=False, verbose=False):
    """examples/orbit: a scene script in the shape of the reference's main(), on the host mirror."""
    src = os.path.join(HERE, "..", "examples", "orbit.cpp")
    if not force and not _stale(EXAMPLE_ORBIT, [src, LIB_HOST]):
        return EXAMPLE_ORBIT
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-o", EXAMPLE_ORBIT, src,
           "-L" + HERE, "-lsrt_host", "-lsrt_hip", "-Wl,-rpath,$ORIGIN/../simple_raytracer_amd"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("g++ failed for examples/orbit")
    return EXAMPLE_ORBIT


EXAMPLE_C = os.path.join(HERE, "..", "examples", "c_abi_minimal")


def build_c_example(force=False, verbose=False):
    """examples/c_abi_minimal: include/srt.h from plain C (gcc), no host mirror."""
    src = os.path.join(HERE, "..", "examples", "c_abi_minimal.c")
    if not force and not _stale(EXAMPLE_C, [src, LIB_HIP, os.path.join(HERE, "..", "include", "srt.h")]):
        return EXAMPLE_C
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-Wextra", "-o", EXAMPLE_C, src, "-L" + HERE, "-lsrt_hip", "-Wl,-rpath,$ORIGIN/../simple_raytracer_amd"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("gcc failed for examples/c_abi_minimal")
    return EXAMPLE_C


def stale_artefacts():
    """Names of the product's native artefacts that are missing or older than their sources (nothing is built)."""
    hdir = os.path.join(CSRC, "host")
    hip_deps = [os.path.join(CSRC, f) for f in ("srt_hip.hip", "srt_device.h", "srt_kernels.h", "srt_packet.h")] + [os.path.join(HERE, "..", "include", "srt.h")]
    host_deps = [os.path.join(hdir, f) for f in ("srt_host.cpp", "srt_jpeg.cpp", "srt_host_c.cpp", "srt_host.h")]
    out = []
    if _stale(LIB_HIP, hip_deps):
        out.append("libsrt_hip.so")
    if _stale(LIB_HOST, host_deps):
        out.append("libsrt_host.so")
    return out


def build_all(force=False, verbose=False):
    return [build_hip(force, verbose), build_host(force, verbose), build_examples(force, verbose), build_c_example(force, verbose)]


def build_diag(verbose=False):
    """Diagnostic build with in-kernel cycle stamps (-DSRT_DIAG): libsrt_hip_diag.so, never loaded by the product."""
    out = os.path.join(HERE, "libsrt_hip_diag.so")
    cmd = [hipcc()] + HIPCC_FLAGS + ["-DSRT_DIAG", "-o", out, os.path.join(CSRC, "srt_hip.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed for the diagnostic build")
    return out


if __name__ == "__main__":
    if "--diag" in sys.argv:
        print(build_diag(verbose=True))
    else:
        build_all(force="--force" in sys.argv, verbose=True)
