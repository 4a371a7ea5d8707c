#!/usr/bin/env python3
"""Where a wave of the vertex-once emit kernel spends its cycles, phase by phase (diagnostic build -DVTMC_EMIT_TIMING: s_memtime deltas
summed over all waves; the marks cost ~10 % of the kernel, the shares are what counts).
    python -c "from volumetricterrain_amd import build; build.build_variant('tools/_ab/libvtmc_phases.so', ['-DVTMC_EMIT_TIMING'])"
    VTMC_LIB=tools/_ab/libvtmc_phases.so python tools/emit_phases.py [tuning, e.g. emit_ablate=1]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import _lib, sharding  # noqa: E402

NAMES = ["wait for the tile (vmcnt)", "tile -> LDS, next tile's loads, block descriptor", "pass 1: active cells", "N: vertex numbering + triangle slots",
         "V: vertex evaluation", "T: vertex gather into the record", "T: staging + record stores", "loop tail"]
n, c, dim = 1024, 128, 130
L = _lib.load()
L.vtmc_debug_emit_phases.argtypes = [ctypes.c_void_p]
L.vtmc_debug_emit_phases.restype = ctypes.c_int32
ex = vt.Extractor(0)
org = sharding.chunk_origins(n, c)
d = torch.empty(len(org) * dim ** 3, dtype=torch.float32, device="cuda")
ex.density_fill_device(vt.density_params("perlin3d", n), org, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())
for spec in sys.argv[1:] or ["base"]:
    kv = {} if spec == "base" else {k: int(v) for k, v in (it.split("=") for it in spec.split(","))}
    indexed = bool(kv.pop("indexed", 0))   # indexed=1: the welded output's kernel (phases 3-6: numbering | vertices | slots | index triples)
    ex.set_output_mode(indexed)
    ex.set_tuning(emit_ablate=0)
    ex.set_tuning(**kv)
    for _ in range(3):
        ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3)
    buf = torch.zeros(16, dtype=torch.int64, device="cuda")
    assert L.vtmc_debug_emit_phases(buf.data_ptr()) == 0
    K = 5
    ms = 0.0
    for _ in range(K):
        ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3)
        ms += ex.last_stage_ms()["emit"] / K
    torch.cuda.synchronize()
    assert L.vtmc_debug_emit_phases(None) == 0
    t = buf.cpu().numpy()
    tot, blocks = float(t[:8].sum()), int(t[8]) // K
    print("== %s: emit %.3f ms (instrumented), %d blocks per launch, %.0f cycles per block and wave" % (spec, ms, blocks, tot / max(t[8], 1)))
    names = NAMES if not indexed else NAMES[:3] + ["N: vertex numbering", "V: vertex evaluation + vertex stores", "pass 2: triangle slots", "T: index triples + stores", NAMES[7]]
    for i in range(8):
        print("   %-52s %5.1f %%   %7.0f cycles per block" % (names[i], 100.0 * t[i] / tot, t[i] / max(t[8], 1)))
