import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, oracle, volumetricterrain_amd as vt
g = oracle.density_volume("perlin3d", 64)
want, want_offs, cases = oracle.extract_grid(g, threads=8, want_cases=True)
ex = vt.Extractor(0)
for version in (1, 2):
    ex.set_tuning(emit_fast_math=0, emit_version=version)
    ex.extract_grid(g)
    got = ex.read_triangles(False)
    nbad = 0
    for f in ("p0","p1","p2","n0","n1","n2"):
        d = np.abs(got[f]-want[f]); bad = np.argwhere(d > 0)
        print(version, f, "mismatches", len(bad), "max", d.max())
        for (i,k) in bad[:3]:
            b = want["block"][i]; print("   tri", i, "comp", k, got[f][i], want[f][i], "pos", want["p"+f[1]][i])
