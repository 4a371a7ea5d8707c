#!/usr/bin/env python3
"""Does the emit kernel's tile fetch get cheaper when the classify pass over the SAME samples ran microseconds earlier -- i.e. when a
batch of chunks (samples + its output) fits the 256 MiB Infinity Cache?  One resident 1024^3 field (512 chunks of 130^3 samples); the
pass is cut into batches of b chunks, two contexts taking turns on one stream; per batch size the sums of the per-stage HIP-event times
and the wall time of the whole pass are printed.  emit_ablate=1 (stores off) isolates the read side.
    python tools/batch_residency_probe.py [b ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import sharding  # noqa: E402

n, c, dim = 1024, 128, 130
exs = [vt.Extractor(0), vt.Extractor(0)]
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
org = sharding.chunk_origins(n, c)
d = torch.empty(len(org) * dim ** 3, dtype=torch.float32, device="cuda")
exs[0].density_fill_device(vt.density_params("perlin3d", n), org, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr(), stream.cuda_stream)
torch.cuda.synchronize()
sizes = [int(a) for a in sys.argv[1:]] or [512, 128, 64, 32, 16, 8]
vs = dim ** 3 * 4


def one_pass(b, acc):
    nb = (512 + b - 1) // b
    T = 0
    for k in range(nb + 1):
        if k < nb:
            e = exs[k % 2]
            cnt = min(b, 512 - k * b)
            e.extract_volumes_device_async(d.data_ptr() + k * b * vs, (c, c, c), (1, dim, dim * dim), cnt, dim ** 3, stream.cuda_stream, 0)
        if k >= 1:
            e = exs[(k - 1) % 2]
            T += e.extract_finish()
            if acc is not None:
                for kk, v in e.last_stage_ms().items():
                    acc[kk] += v
    return T


for ablate in (0, 1):
    for e in exs:
        e.set_tuning(emit_ablate=ablate)
    print("emit_ablate=%d (%s)" % (ablate, "stores off: the read side alone" if ablate else "the shipped kernel"))
    for b in sizes:
        for _ in range(3):
            one_pass(b, None)
        acc = {"classify": 0.0, "scan": 0.0, "emit": 0.0, "total": 0.0}
        K = 10
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            T = one_pass(b, acc)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / K * 1e3
        print("  batch %4d chunks (%6.1f MB of samples): wall %.3f ms/pass, classify %.3f scan %.3f emit %.3f sum %.3f  T=%d" % (
            b, b * vs / 1e6, wall, acc["classify"] / K, acc["scan"] / K, acc["emit"] / K, acc["total"] / K, T))
