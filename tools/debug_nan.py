import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, oracle, volumetricterrain_amd as vt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = oracle.density_volume("perlin3d", n)
want, want_offs, _ = oracle.extract_grid(g, threads=16)
blocks = oracle.all_blocks(n, n, n)
ex = vt.Extractor(0)
g64 = oracle.density_volume("perlin3d", 64)
for ver, fm in ((2,0),(1,0),(1,1),(2,1)):
    ex.set_tuning(emit_fast_math=fm, emit_version=ver)
    ex.extract_grid(g64)
for rep in range(3):
    ex.extract_grid(g)
    got, offs = ex.read_triangles()
    print("rep", rep, "offs equal", np.array_equal(offs, want_offs), "block equal", np.array_equal(got["block"], want["block"]))
    for f in ("p0","n0","n1","n2"):
        ng, nw = np.isnan(got[f]), np.isnan(want[f])
        d = np.abs(np.where(nw|ng, 0, got[f]) - np.where(nw|ng, 0, want[f]))
        badrows = np.unique(np.argwhere((ng != nw) | (d > 1e-5))[:, 0])
        print("  ", f, "nan got/want", ng.sum(), nw.sum(), "bad rows", len(badrows), "max dev", d.max())
        for i in badrows[:8]:
            print("      tri", i, "blk", want["block"][i], "local", i - want_offs[want["block"][i]], "of", want_offs[want["block"][i]+1]-want_offs[want["block"][i]], "got", got[f][i], "want", want[f][i])
