"""CPU restatement of the reference voxelizer.  Test infrastructure only.

Follows /root/reference/mosaic3d/src/models/utils/structure.py:21-66 (fnv_hash_vec,
mean_pooling), :197-251 (Point.sparsify) and /root/reference/mosaic3d/src/models/utils/misc.py:8-68.
"""
import numpy as np

FNV_OFFSET = np.uint64(14695981039346656037)
FNV_PRIME = np.uint64(1099511628211)


def fnv_hash_vec(arr):
    """FNV64-1A over the integer columns of ``arr`` (structure.py:21-31).

    Negative values wrap through astype(uint64) exactly like the reference."""
    arr = np.asarray(arr)
    assert arr.ndim == 2
    a = arr.astype(np.int64).astype(np.uint64)
    h = np.full(a.shape[0], FNV_OFFSET, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(a.shape[1]):
            h = h * FNV_PRIME
            h = np.bitwise_xor(h, a[:, j])
    return h


def ravel_hash(coords):
    """ravel_multi_index_auto_shape (misc.py:36-68): row-major linear index of
    (coords - min) over shape (max - min + 1), as int64."""
    c = np.asarray(coords).astype(np.int64)
    c = c - c.min(0)
    shape = c.max(0) + 1
    strides = np.array([np.prod(shape[i + 1:]) for i in range(len(shape))], dtype=np.int64)
    return (c * strides).sum(-1)


def offset2bincount(offset):
    """misc.py:8-13: diff(offset), prepending 0 only when offset[0] != 0."""
    offset = np.asarray(offset, dtype=np.int64)
    if offset[0] != 0:
        offset = np.concatenate([[0], offset])
    return np.diff(offset)


def offset2batch(offset):
    """misc.py:16-21."""
    bc = offset2bincount(offset)
    return np.repeat(np.arange(len(bc), dtype=np.int64), bc)


def batch2offset(batch):
    """misc.py:24-26."""
    return np.cumsum(np.bincount(np.asarray(batch, dtype=np.int64))).astype(np.int64)


def grid_coords(coord, grid_size, mode="div"):
    """structure.py:219-220: (coord / grid_size).int() - min over the whole batch.

    coord is float32; the division is an IEEE float32 division by float32(grid_size)
    (what torch's CPU kernel computes for tensor / python-float); ``mode="recip"`` restates
    torch's CUDA kernel, which multiplies by the float32 reciprocal instead.  ``.int()``
    truncates toward zero."""
    coord = np.asarray(coord, dtype=np.float32)
    if mode == "div":
        q = coord / np.float32(grid_size)
    else:
        q = coord * (np.float32(1.0) / np.float32(grid_size))
    g = np.trunc(q).astype(np.int32)
    return g - g.min(0)


def mean_pooling(coordinates, features, hash_method="fnv"):
    """structure.py:34-66.  Returns voxel_coords, voxel_feats, v2p_map, key_unique."""
    coordinates = np.asarray(coordinates)
    if hash_method == "fnv":
        key = fnv_hash_vec(coordinates)
    elif hash_method == "ravel":
        key = ravel_hash(coordinates)
    else:
        raise ValueError(f"Unknown hash method: {hash_method}")
    idx_sort = np.argsort(key, kind="stable")
    key_sort = key[idx_sort]
    ukey, index, inverse, count = np.unique(key_sort, return_index=True, return_inverse=True,
                                            return_counts=True)
    voxel_coords = coordinates[idx_sort[index]]
    f = np.asarray(features)
    sums = np.zeros((len(ukey), f.shape[1]), dtype=np.float64)
    np.add.at(sums, inverse, f[idx_sort].astype(np.float64))
    voxel_feats = (sums / count[:, None]).astype(f.dtype)
    v2p_map = np.zeros_like(inverse, dtype=np.int64)
    v2p_map[idx_sort] = inverse
    return voxel_coords, voxel_feats, v2p_map, ukey


def sparsify(coord, feat, offset, grid_size, pad=128, hash_method="fnv", grid_mode="div"):
    """Point.__init__ + Point.sparsify (structure.py:137-143,197-251).

    Returns dict(voxel_coords [N0,4] int32 (b,x,y,z), voxel_feats [N0,C], v2p_map [N] int64,
    spatial_shape list[3], batch_size int, grid_coord [N,3] int32, key [N0] uint64/int64)."""
    batch = offset2batch(offset)
    g = grid_coords(coord, grid_size, grid_mode)
    spatial_shape = (g.max(0).astype(np.int64) + pad).tolist()
    bc = np.concatenate([batch[:, None].astype(np.int32), g.astype(np.int32)], 1)
    vc, vf, v2p, ukey = mean_pooling(bc, feat, hash_method)
    return dict(voxel_coords=vc.astype(np.int32), voxel_feats=vf, v2p_map=v2p,
                spatial_shape=spatial_shape, batch_size=int(batch[-1]) + 1, grid_coord=g,
                key=ukey)
