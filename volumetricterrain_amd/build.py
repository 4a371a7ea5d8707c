"""Builds libvtmc.so (the C-ABI library of include/vtmc.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container as well as on the GPU
box.  -ffp-contract=off keeps every FP32 expression a single IEEE operation sequence (the CPU
oracle is compiled the same way), so positions and normals normally agree bit for bit and the
1e-5 bar of the north-star is met with a wide margin.
"""
import fcntl
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvtmc.so")
SOURCES = ["vtmc_api.hip", "classify_kernels.hip", "emit_kernels.hip", "terrain.hip", "density.hip",
           "chunk_io.hip", "comm.hip"]
HEADERS = ["vtmc_internal.h", "vtmc_ctx.h", "mc_device.h", "emit_device.h", "mc_tables_packed.h", os.path.join("..", "..", "include", "vtmc.h")]
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent FP32 operations into v_pk_fma_f32 / v_pk_add_f32 (+ moves to
# pair the operands); on gfx950 a packed FP32 op costs more than the two scalar ones it replaces
# (MI355X_MICROARCH.md, "packed f32 VALU ... an anti-lever"): the sampler runs 30 % faster without it.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function", "-fno-slp-vectorize"]


# the device code of the extract path and the tuning defaults that select its variants (host-side files do not change a kernel's traffic)
EXTRACT_KERNEL_FILES = ["classify_kernels.hip", "emit_kernels.hip", "emit_device.h", "mc_device.h", "mc_tables_packed.h", "vtmc_internal.h"]


def kernel_source_hash():
    """SHA-256 over the extract path's kernel sources: profiles/pmc_traffic.json records it next to the counter values it was
    measured with, and bench.py reports `traffic` only while the sources still hash to the same value."""
    import hashlib
    h = hashlib.sha256()
    for name in EXTRACT_KERNEL_FILES:
        h.update(name.encode())
        h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compiles to a unique temporary next to the target and renames it into place under a file lock:
    every rank of a torchrun job calls this, and a rank must never dlopen a half-written library."""
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libvtmc.so (there is no CPU fallback)")
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():   # another process built it while this one waited
                return LIB
            tmpdir = tempfile.mkdtemp(prefix=".build-", dir=HERE)   # hipcc's offload-bundle temporaries stay in here
            try:
                tmp = os.path.join(tmpdir, "libvtmc.so")
                cmd = [hipcc] + FLAGS + ["-ldl", "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
                if verbose:
                    print(" ".join(cmd))
                subprocess.run(cmd, check=True)
                os.replace(tmp, LIB)
            finally:
                shutil.rmtree(tmpdir, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


def build_variant(out, extra_flags):
    """A diagnostic build next to the product library (e.g. tools/_ab/libvtmc_timeline.so with -DVTMC_TIMELINE); loaded through VTMC_LIB."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc] + FLAGS + list(extra_flags) + ["-ldl", "-o", out] + [os.path.join(CSRC, s) for s in SOURCES], check=True)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
