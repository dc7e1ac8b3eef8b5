import numpy as np


def morton_key(idx):
    """Interleave the low 16 bits of (x, y, z): x is the most significant of each bit triple."""
    def spread(v):
        v = v.astype(np.uint64) & np.uint64(0xFFFF)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    idx = np.asarray(idx)
    return (spread(idx[:, 1]) << np.uint64(2)) | (spread(idx[:, 2]) << np.uint64(1)) | spread(idx[:, 3])
