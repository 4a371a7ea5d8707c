import numpy as np
act = np.unpackbits(np.load("gpurun_out/active_blocks.npy")).astype(bool)[:512*4096]
act = act.reshape(512, 16, 16, 16)   # v, bz, by, bx
dim = 130
def lines_of(v, bz, by, bx, width):
    # rows y in 0..9, z in 0..9 of the tile starting at (8bx, 8by, 8bz); width floats along x
    y = (8*by)[:,None,None] + np.arange(10)[None,:,None]
    z = (8*bz)[:,None,None] + np.arange(10)[None,None,:]
    start = (v[:,None,None].astype(np.int64)*dim**3 + z*dim*dim + y*dim + (8*bx)[:,None,None]) * 4
    end = start + width*4 - 1
    l0, l1 = start // 128, end // 128
    return l0.reshape(len(v), -1), l1.reshape(len(v), -1)
v, bz, by, bx = np.nonzero(act)
l0, l1 = lines_of(v, bz, by, bx, 10)
per_block = (1 + (l1 != l0)).sum()
allv = np.concatenate([l0.ravel(), l1.ravel()])
print("active blocks", len(v))
print("sum over blocks of lines (no sharing): %.2f M  = %.2f GB" % (per_block/1e6, per_block*128/1e9))
print("distinct lines: %.2f M = %.2f GB" % (len(np.unique(allv))/1e6, len(np.unique(allv))*128/1e9))
# pair merging: units = x-pairs (bx even, bx+1); a unit with both active loads 18-wide rows once, else the single block's rows
pair = act.reshape(512,16,16,8,2)
both = pair.all(-1); one = pair.any(-1) & ~both
vb, zb, yb, xb = np.nonzero(both)
a0, a1 = lines_of(vb, zb, yb, 2*xb, 18)
n_both = ((a1 - a0) + 1).sum()
vo, zo, yo, xo = np.nonzero(one)
which = pair[vo, zo, yo, xo, 1].astype(int)
b0, b1 = lines_of(vo, zo, yo, 2*xo + which, 10)
n_one = (1 + (b1 != b0)).sum()
print("pairs both active %d, single %d" % (len(vb), len(vo)))
print("with pair loads: %.2f M = %.2f GB" % ((n_both+n_one)/1e6, (n_both+n_one)*128/1e9))
# quads
quad = act.reshape(512,16,16,4,4)
