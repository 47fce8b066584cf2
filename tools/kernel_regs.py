#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in libsrt_hip.so (from the code object's metadata notes).
Usage: python tools/kernel_regs.py [substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "simple_raytracer_amd", "libsrt_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
def main():
    pats = sys.argv[1:]
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", f"--input={LIB}", f"--output={d}/k.co",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=False, capture_output=True)
        co = f"{d}/k.co"
        if not os.path.exists(co) or os.path.getsize(co) == 0:
            # fat binary section: extract with objcopy
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={d}/fat.bin", LIB], check=True)
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", f"--input={d}/fat.bin", f"--output={co}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    cur = {}
    rows = []
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "name" and v.startswith(("_Z", "k_")) and "kd" not in v:
            cur["name"] = v
        if k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count", "vgpr_spill_count"):
            cur[k] = v
        if k == "wavefront_size":
            if "name" in cur:
                rows.append(cur)
            cur = {}
    dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, dem):
        n = n.replace("void ", "").split("(")[0]
        if pats and not any(p in n for p in pats):
            continue
        print(f"{n:90s} vgpr {r.get('vgpr_count','?'):>4} sgpr {r.get('sgpr_count','?'):>4} scratch {r.get('private_segment_fixed_size','?'):>5} lds {r.get('group_segment_fixed_size','?'):>6} spill {r.get('vgpr_spill_count','0')}")
if __name__ == "__main__":
    main()
