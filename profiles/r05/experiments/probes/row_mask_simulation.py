"""How many tile rows would a per-(y, z) row mask save over the outer product of a y mask and a z mask?  (CPU, oracle sampler, a few chunks
of the bench field: the 1024^3 perlin3d grid as 128^3 chunks)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle as om
om.build()

rng = np.random.default_rng(1)
tot = dict(blocks=0, outer=0, plus=0, full=0)
for _ in range(12):
    o = tuple(int(v) * 128 for v in rng.integers(0, 8, 3))
    g = om.density_volume("perlin3d", 1024, origin=o, dims=(130, 130, 130))   # [x, y, z]
    s = g > 0
    c = [s[i:i + 128, j:j + 128, k:k + 128] for i in (0, 1) for j in (0, 1) for k in (0, 1)]
    allin = np.logical_and.reduce(c)
    anyin = np.logical_or.reduce(c)
    act = anyin & ~allin                                   # cells with triangles [x, y, z]
    a = act.reshape(16, 8, 16, 8, 16, 8).transpose(0, 2, 4, 1, 3, 5)   # [bx, by, bz, cx, cy, cz]
    rowact = a.any(3)                                      # [bx, by, bz, cy, cz]: a cell row along x holds triangles
    blk = rowact.any((3, 4))
    ra = rowact[blk]                                       # [n, cy, cz]
    n = len(ra)
    ym, zm = ra.any(2), ra.any(1)                          # [n, 8]
    def dil(m):                                            # cell layers -> the 10 sample rows they touch (c, c + 1, c + 2)
        out = np.zeros((n, 10), bool)
        for d in range(3):
            out[:, d:d + 8] |= m
        return out
    outer = dil(ym)[:, :, None] & dil(zm)[:, None, :]
    full = np.zeros((n, 10, 10), bool)
    plus = np.zeros((n, 10, 10), bool)
    for dy in range(3):
        for dz in range(3):
            full[:, dy:dy + 8, dz:dz + 8] |= ra
            if not (dy == 2 and dz == 2):
                plus[:, dy:dy + 8, dz:dz + 8] |= ra
    tot["blocks"] += n
    tot["outer"] += int(outer.sum())
    tot["full"] += int(full.sum())
    tot["plus"] += int(plus.sum())
b = tot["blocks"]
print("blocks with triangles: %d;  tile rows fetched per block of 100: outer product of y and z masks (now) %.1f, per-(y, z) mask with the 3 x 3 dilation %.1f, without the far corner %.1f"
      % (b, tot["outer"] / b, tot["full"] / b, tot["plus"] / b))
