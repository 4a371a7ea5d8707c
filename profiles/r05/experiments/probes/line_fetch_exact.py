#!/usr/bin/env python3
"""How many distinct 128-byte lines lie under the tile rows the emit kernel asks for -- exactly, with the kernel's own row masks
(outer product of the y and z layers that hold a cell with triangles, each widened by the two rows above), on the bench field
(1024^3 perlin3d from the device sampler, 512 chunks of 130^3 samples).  Compare with TCC_EA0_RDREQ_128B of the emit kernel
(profiles/*/pmc_requests_by_size.json): the difference is what the L2s fetch more than once.
    python tools/_ab/line_fetch_exact.py [n_chunks]      (on the GPU box; ~2 minutes of numpy for 512 chunks)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import volumetricterrain_amd as vt  # noqa: E402
from volumetricterrain_amd import sharding  # noqa: E402

n, c, dim = 1024, 128, 130
n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ex = vt.Extractor(0)
org = sharding.chunk_origins(n, c)[:n_chunks]
d = torch.empty(len(org) * dim ** 3, dtype=torch.float32, device="cuda")
ex.density_fill_device(vt.density_params("perlin3d", n), org, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())
base_mod = d.data_ptr() % 128
print("device base %% 128 = %d" % base_mod)
T = ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), n_chunks, dim ** 3)
_, off_ptr, _ = ex.device_results()
offs = ex.copy_u32(off_ptr, n_chunks * 4096 + 1).astype(np.int64)
active_gpu = np.diff(offs) > 0
print("T = %d, active blocks (device) %d" % (T, active_gpu.sum()))

t0 = time.time()
all_lines, masked_lines = [], []
n_rows_all = n_rows_masked = 0
sum_all = sum_masked = 0
n_active = 0
ar10 = np.arange(10)
for v in range(n_chunks):
    f = d[v * dim ** 3:(v + 1) * dim ** 3].cpu().numpy().reshape(dim, dim, dim)   # [z, y, x]
    s = f > 0
    a = s[:129, :129, :129]
    any8 = np.zeros((128, 128, 128), bool)
    all8 = np.ones((128, 128, 128), bool)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                w = a[dz:dz + 128, dy:dy + 128, dx:dx + 128]
                any8 |= w
                all8 &= w
    cell = (any8 & ~all8).reshape(16, 8, 16, 8, 16, 8)            # bz, z, by, y, bx, x
    ym = cell.any(axis=(1, 5)).transpose(0, 1, 3, 2)              # bz, by, bx, y
    zm = cell.any(axis=(3, 5)).transpose(0, 2, 3, 1)              # bz, by, bx, z
    act = ym.any(-1)
    assert np.array_equal(act.ravel(), active_gpu[v * 4096:(v + 1) * 4096]), "CPU and device disagree on the active blocks of chunk %d" % v
    bz, by, bx = np.nonzero(act)
    n_active += len(bz)
    ymb, zmb = ym[bz, by, bx], zm[bz, by, bx]                     # [n, 8]
    ny = np.zeros((len(bz), 10), bool)
    nz = np.zeros((len(bz), 10), bool)
    for k in (0, 1, 2):
        ny[:, k:k + 8] |= ymb
        nz[:, k:k + 8] |= zmb
    need = nz[:, :, None] & ny[:, None, :]                        # [n, z, y]
    z = (8 * bz)[:, None, None] + ar10[None, :, None]
    y = (8 * by)[:, None, None] + ar10[None, None, :]
    start = base_mod + (np.int64(v) * dim ** 3 + (z * dim + y) * dim + (8 * bx)[:, None, None]) * 4
    l0, l1 = start // 128, (start + 39) // 128
    two = l1 != l0
    sum_all += int((1 + two).sum())
    sum_masked += int((1 + two)[need].sum())
    n_rows_all += need.size
    n_rows_masked += int(need.sum())
    all_lines.append(np.unique(np.concatenate([l0.ravel(), l1.ravel()])))
    masked_lines.append(np.unique(np.concatenate([l0[need], l1[need]])))
    if v % 64 == 63:
        print("  chunk %d  (%.0f s)" % (v + 1, time.time() - t0), flush=True)
da = len(np.unique(np.concatenate(all_lines)))
dm = len(np.unique(np.concatenate(masked_lines)))
print("active blocks %d" % n_active)
print("rows asked for with the masks: %.1f %% of all (%d of %d)" % (100.0 * n_rows_masked / n_rows_all, n_rows_masked, n_rows_all))
print("all 100 rows   : %.2f M line touches, %.2f M distinct lines = %.3f GB" % (sum_all / 1e6, da / 1e6, da * 128 / 1e9))
print("with row masks : %.2f M line touches, %.2f M distinct lines = %.3f GB" % (sum_masked / 1e6, dm / 1e6, dm * 128 / 1e9))
