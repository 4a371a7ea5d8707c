import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import volumetricterrain_amd as vt
from volumetricterrain_amd import sharding
n, c = 1024, 128
dim = c + 2
origins = sharding.chunk_origins(n, c)
ex = vt.Extractor(0)
d = torch.empty(len(origins) * dim ** 3, dtype=torch.float32, device="cuda")
ex.density_fill_device(vt.density_params("perlin3d", n), origins, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())
T = ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(origins), dim ** 3)
_, d_off, _ = ex.device_results()
B = len(origins) * 16 ** 3
off = ex.copy_u32(d_off, B + 1).astype(np.int64)
cnt = np.diff(off)
print("T", T, off[-1], "active blocks", (cnt > 0).sum(), "of", B)
br = cnt.reshape(-1, 8)          # bricks: 8 consecutive blocks along x (nbx = 16)
act = (br > 0).sum(1)
print("bricks", len(act), "active", (act > 0).sum(), "hist of active blocks per brick", np.bincount(act, minlength=9))
tri = br.sum(1)
print("tris per active brick: mean %.1f max %d" % (tri[act > 0].mean(), tri.max()))
g = (act > 0).reshape(-1, 64).sum(1)
print("active bricks per group of 64: mean %.1f max %d min %d" % (g.mean(), g.max(), g.min()))
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/active_blocks.npy", np.packbits(cnt > 0))
# row masks: the count words carry them (counts buffer is internal) -- approximated by all rows in the simulation
