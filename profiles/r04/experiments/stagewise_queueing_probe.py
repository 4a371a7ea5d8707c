#!/usr/bin/env python3
"""Stage-wise queueing (vtmc_extract_continue): the bench workload as
  plain   : two whole steps in flight (classify, scan, emit of step k, then of step k + 1 ...)            = bench.py --pipeline 2
  inter   : main stream C1 C2 E1 C3 E2 C4 E3 ..., the scan of step k on a second stream beside the classify kernel of step k + 1
  inter0  : the same order with the scan on the main stream (C1 C2 S1 E1 C3 S2 E2 ...): what the order alone does
Three contexts take turns; every step's T is checked.  Wall time per step, alternating rounds in one process.
    python tools/interleave_probe.py [steps]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import sharding  # noqa: E402

n, c, dim = 1024, 128, 130
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 512
exs = [vt.Extractor(0) for _ in range(3)]
stream = torch.cuda.Stream()
aux = torch.cuda.Stream()
torch.cuda.set_stream(stream)
org = sharding.chunk_origins(n, c)[:n_chunks]
d = torch.empty(len(org) * dim ** 3, dtype=torch.float32, device="cuda")
exs[0].density_fill_device(vt.density_params("perlin3d", n), org, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr(), stream.cuda_stream)
torch.cuda.synchronize()
args = (d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(org), dim ** 3, stream.cuda_stream)
for e in exs:
    e.set_tuning(stage_events=0)
T0 = exs[0].extract_volumes_device(*args[:5], args[5])
for e in exs[1:]:
    assert e.extract_volumes_device(*args[:5], args[5]) == T0


def plain(k_steps):
    for i in range(k_steps + 1):
        if i < k_steps:
            exs[i % 2].extract_volumes_device_async(*args, 0)
        if i >= 1:
            assert exs[(i - 1) % 2].extract_finish() == T0


def inter(k_steps, aux_ptr):
    for i in range(k_steps + 2):
        if i < k_steps:
            exs[i % 3].extract_volumes_device_async(*args, 4)      # classify of step i
        if 1 <= i <= k_steps:
            exs[(i - 1) % 3].extract_continue(aux_ptr)              # scan + emit of step i - 1
        if i >= 2:
            assert exs[(i - 2) % 3].extract_finish() == T0


modes = {"plain": lambda: plain(K), "inter": lambda: inter(K, aux.cuda_stream), "inter0": lambda: inter(K, None)}
for f in modes.values():
    f()
res = {m: [] for m in modes}
for _ in range(5):
    for m, f in modes.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / K * 1e3)
for m, v in res.items():
    print("%-7s  ms per step: median %.4f  min %.4f   (%s)" % (m, statistics.median(v), min(v), " ".join("%.4f" % x for x in v)))
