"""Builds libvtmc.so (the C-ABI library of include/vtmc.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container as well as on the GPU
box.  -ffp-contract=off keeps every FP32 expression a single IEEE operation sequence (the CPU
oracle is compiled the same way), so positions and normals normally agree bit for bit and the
1e-5 bar of the north-star is met with a wide margin.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvtmc.so")
SOURCES = ["vtmc_api.hip", "classify_kernels.hip", "emit_kernels.hip", "sweep_kernels.hip", "terrain.hip", "density.hip"]
HEADERS = ["vtmc_internal.h", "mc_device.h", "emit_device.h", "mc_tables_packed.h", os.path.join("..", "..", "include", "vtmc.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libvtmc.so (there is no CPU fallback)")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
