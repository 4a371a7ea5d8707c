import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, oracle, volumetricterrain_amd as vt
n = 256
g = oracle.density_volume("perlin3d", n)
want, want_offs, _ = oracle.extract_grid(g, threads=16)
ex = vt.Extractor(0)
g64 = oracle.density_volume("perlin3d", 64)
def check(tag):
    ex.extract_grid(g)
    got, offs = ex.read_triangles()
    bad = 0
    for f in ("p0","p1","p2","n0","n1","n2"):
        ng, nw = np.isnan(got[f]), np.isnan(want[f])
        d = np.abs(np.where(nw|ng, 0, got[f]) - np.where(nw|ng, 0, want[f]))
        rows = np.unique(np.argwhere((ng != nw) | (d > 1e-5))[:, 0])
        bad += len(rows)
    print(tag, "bad rows", bad)
mode = sys.argv[1]
if mode == "a":      # variants first, but capacity preallocated
    ex.reserve_triangles(3_000_000)
    for ver, fm in ((2,0),(1,0),(1,1),(2,1)):
        ex.set_tuning(emit_fast_math=fm, emit_version=ver); ex.extract_grid(g64)
    check("prealloc after variants")
    check("again")
elif mode == "b":    # no variants; force regrow repeatedly
    for i in range(4):
        ex.reserve_triangles(1000)
        check("regrow %d" % i)
elif mode == "c":    # variants, regrow, repeatedly
    for i in range(3):
        for ver, fm in ((2,0),(1,0),(1,1),(2,1)):
            ex.set_tuning(emit_fast_math=fm, emit_version=ver); ex.extract_grid(g64)
        ex.reserve_triangles(1000)
        check("variants+regrow %d" % i)
        check("  steady")
elif mode == "d":    # only exact v2 on 64 then 256 fast
    ex.set_tuning(emit_fast_math=0, emit_version=2); ex.extract_grid(g64)
    ex.set_tuning(emit_fast_math=1, emit_version=2)
    check("after exact v2")
    check("  steady")
