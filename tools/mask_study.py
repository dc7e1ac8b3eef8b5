"""CPU study: executed/useful MFMA rows for different output orderings and group sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from segmentation_amd import scenes

def morton(idx):
    def spread(v):
        v = v.astype(np.uint64) & 0xFFFF
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    return (spread(idx[:,1]) << 2) | (spread(idx[:,2]) << 1) | spread(idx[:,3])

def waste(valid, order, g):
    v = valid[:, order]
    n = v.shape[1]
    npad = (n + g - 1)//g*g
    vp = np.zeros((27, npad), bool); vp[:, :n] = v
    act = vp.reshape(27, -1, g).any(2)
    return act.sum()*g / v.sum()

tv = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
sc = scenes.make_scene(0, tv); n = len(sc['coord'])
sp = oracle.sparsify(sc['coord'], sc['feat'], np.array([0, n]), 0.02)
idx = sp['voxel_coords']
pair = oracle.subm_rulebook(idx, sp['spatial_shape'])
valid = pair >= 0
N = idx.shape[0]
print("N", N, "avg nbrs", valid.sum()/N)
mk = np.zeros(N, np.uint32)
for k in range(27): mk |= valid[k].astype(np.uint32) << k
mo = np.argsort(morton(idx), kind='stable')
for g in (16, 32):
    print("group", g, "morton:", round(waste(valid, mo, g), 3), " global-mask-sort:", round(waste(valid, np.argsort(mk, kind='stable'), g), 3))
    for B in (128, 256, 512, 1024, 4096):
        # morton blocks of B voxels, sorted by mask inside the block
        blk = np.arange(N)//B
        key = blk.astype(np.uint64) << np.uint64(32) | mk[mo].astype(np.uint64)
        o2 = mo[np.argsort(key, kind='stable')]
        print("   block", B, "mask-sorted:", round(waste(valid, o2, g), 3))
# per-offset compaction per block of BM outputs (approach b): padding to g per (block,k)
for BM in (128, 256, 512):
    for g in (16, 32):
        v = valid[:, mo]
        npad = (N + BM - 1)//BM*BM
        vp = np.zeros((27, npad), bool); vp[:, :N] = v
        c = vp.reshape(27, -1, BM).sum(2)
        ex = ((c + g - 1)//g*g).sum()
        print("compaction BM", BM, "g", g, ":", round(ex / v.sum(), 3))
