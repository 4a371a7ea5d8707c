#!/usr/bin/env python3
"""Runs the density sampler alone on one 64-chunk batch of the fbm8 2048^3 world (profiling aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import sharding  # noqa: E402

n, c, dim = 2048, 128, 130
org = sharding.chunk_origins(n, c)[:64]
ex = vt.Extractor(0)
d = torch.empty(64 * dim ** 3, dtype=torch.float32, device="cuda")
prm = vt.density_params(sys.argv[1] if len(sys.argv) > 1 else "fbm8", n)
if len(sys.argv) > 2:
    ex.set_tuning(density_ablate=int(sys.argv[2]))
if os.environ.get("VTMC_FILL_SIGNS") == "1":
    ex.set_tuning(fill_keeps_signs=1)
ms = []
for _ in range(6):
    ex.density_fill_device(prm, org, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())
    ms.append(ex.last_fill_ms())
print("fill ms:", " ".join("%.4f" % m for m in ms))
