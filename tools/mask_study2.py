"""CPU study: active (tile, offset) stages for mask-sorted orderings at block (fetch) and wave (MFMA) granularity."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from segmentation_amd import scenes

def spread(v):
    v = v.astype(np.uint64) & 0xFFFF
    v = (v | (v << 32)) & 0x1F00000000FFFF
    v = (v | (v << 16)) & 0x1F0000FF0000FF
    v = (v | (v << 8)) & 0x100F00F00F00F00F
    v = (v | (v << 4)) & 0x10C30C30C30C30C3
    v = (v | (v << 2)) & 0x1249249249249249
    return v
def morton(idx): return (spread(idx[:,1]) << 2) | (spread(idx[:,2]) << 1) | spread(idx[:,3])
def waste(valid, order, g):
    v = valid[:, order]; n = v.shape[1]; npad = (n + g - 1)//g*g
    vp = np.zeros((27, npad), bool); vp[:, :n] = v
    return vp.reshape(27, -1, g).any(2).sum()*g / v.sum()

tv = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
sc = scenes.make_scene(0, tv); n = len(sc['coord'])
sp = oracle.sparsify(sc['coord'], sc['feat'], np.array([0, n]), 0.02)
idx = sp['voxel_coords']; pair = oracle.subm_rulebook(idx, sp['spatial_shape'])
valid = pair >= 0; N = idx.shape[0]
mo = np.argsort(morton(idx), kind='stable')
def maskkey(bits):
    mk = np.zeros(N, np.uint64)
    for i, k in enumerate(bits): mk |= valid[k].astype(np.uint64) << np.uint64(len(bits) - 1 - i)
    return mk
freq = valid.mean(1)
orders = {"natural": list(range(27)), "balanced-first": list(np.argsort(np.abs(freq - 0.5))),
          "rare-first": list(np.argsort(freq)), "common-first": list(np.argsort(-freq))}
print("N", N, "freq", np.round(freq, 2))
print("morton:", [round(waste(valid, mo, g), 3) for g in (32, 128, 256)])
for name, bits in orders.items():
    mk = maskkey(bits)
    for B in (1024, 4096, 16384, 65536, 1 << 30):
        blk = (np.arange(N)//B).astype(np.uint64)
        o2 = mo[np.argsort((blk << np.uint64(32)) | mk[mo], kind='stable')]
        print(f"{name:15s} B={B:10d}", [round(waste(valid, o2, g), 3) for g in (32, 128, 256)])
# fixed class order (corners, edges, faces), with and without the face bits
def nzc(k): return (k // 9 != 1) + ((k // 3) % 3 != 1) + (k % 3 != 1)
ce = [k for k in range(27) if nzc(k) == 3] + [k for k in range(27) if nzc(k) == 2]
fa = [k for k in range(27) if nzc(k) == 1]
for name, bits in (("corner,edge,face", ce + fa), ("corner,edge", ce), ("corner only", ce[:8])):
    mk = maskkey(bits)
    for B in (4096, 16384):
        blk = (np.arange(N)//B).astype(np.uint64)
        o2 = mo[np.argsort((blk << np.uint64(32)) | mk[mo], kind='stable')]
        print(f"{name:18s} B={B:8d}", [round(waste(valid, o2, g), 3) for g in (32, 128, 256)])
# spatial cells of 2^cs voxels per side instead of blocks of consecutive Morton rows; base order inside
print("cells:")
co = [k for k in range(27) if nzc(k) == 3]; ed = [k for k in range(27) if nzc(k) == 2]
for cs in (5, 6, 7):
    cell = morton(np.concatenate([idx[:, :1], idx[:, 1:] >> cs], 1))
    for name, bits in (("corner,edge", co + ed), ("edge", ed), ("edge,corner", ed + co), ("edge+face", ed + fa)):
        mk = maskkey(bits)
        o2 = np.argsort((cell << np.uint64(32)) | mk, kind='stable')
        print(f"cs={cs} {name:14s}", [round(waste(valid, o2, g), 3) for g in (32, 128, 256)])
