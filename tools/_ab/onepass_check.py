import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import volumetricterrain_amd as vt
from volumetricterrain_amd import sharding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
c = int(sys.argv[2]) if len(sys.argv) > 2 else 128
kind = sys.argv[3] if len(sys.argv) > 3 else "perlin3d"
dim = c + 2
origins = sharding.chunk_origins(n, c)
ex = vt.Extractor(0)
d = torch.empty(len(origins) * dim ** 3, dtype=torch.float32, device="cuda")
ex.density_fill_device(vt.density_params(kind, n), origins, (dim, dim, dim), (1, dim, dim * dim), dim ** 3, d.data_ptr())
B = len(origins) * (c // 8) ** 3
res = {}
for op in (0, 1):
    for fm in (1, 0):
        ex.set_tuning(one_pass=op, emit_fast_math=fm)
        T = ex.extract_volumes_device(d.data_ptr(), (c, c, c), (1, dim, dim * dim), len(origins), dim ** 3)
        tri, off, vc = ex.device_results()
        res[(op, fm)] = (T, ex.copy_to_host(tri, 76 * T).copy(), ex.copy_u32(off, B + 1).copy(), ex.copy_u32(vc, 2 * len(origins)).copy(), ex.last_stage_ms())
        print("one_pass", op, "fast", fm, "T", T, ex.last_stage_ms(), flush=True)
ok = True
for fm in (1, 0):
    a, b = res[(0, fm)], res[(1, fm)]
    same = a[0] == b[0] and np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    print("fast", fm, "identical:", same, "offsets", np.array_equal(a[2], b[2]), "volcounts", np.array_equal(a[3], b[3]))
    if not same and a[0] == b[0]:
        ra, rb = a[1].reshape(-1, 76), b[1].reshape(-1, 76)
        bad = np.nonzero((ra != rb).any(1))[0]
        print("  differing records", len(bad), bad[:10])
    ok &= same
print("OK" if ok else "MISMATCH")
