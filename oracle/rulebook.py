"""CPU restatement of the sparse-convolution kernel maps ("rulebooks").  Test infrastructure only.

The reference obtains these from the third-party spconv wheel at the call sites
/root/reference/mosaic3d/src/models/networks/spunet/spconv_unet_v1m3_pdnorm.py:98-117,205-212
(SubMConv3d k3), :154-161 (SparseConv3d k2 s2) and :184-190 (SparseInverseConv3d).  spconv is not
vendored, so the definitions below restate spconv 2.x's published semantics: parity unpinned by
reference tests; anchored by dense-conv3d known-answer tests.

Conventions
-----------
indices  : [N,4] int32 rows (b, d0, d1, d2), all rows distinct.
SubM k3  : pair[k, j] = i  with  indices[i] == indices[j] + (0, k0-1, k1-1, k2-1),
           k = (k0*3 + k1)*3 + k2, else -1.  Output order == input order.
k2 s2 p0 : output voxel set = unique(b, d >> 1) restricted to d>>1 < out_shape, sorted ascending by
           the row-major linear index over (batch, out_shape); pair_fwd[k, o] = i or -1,
           pair_bwd[k, i] = o or -1, k = ((d0&1)*2 + (d1&1))*2 + (d2&1).
"""
import numpy as np


def kernel_offsets(ksize=3):
    r = np.arange(ksize) - (ksize // 2)
    return np.array([(a, b, c) for a in r for b in r for c in r], dtype=np.int64)


def _lin(b, xyz, shape):
    s = np.asarray(shape, dtype=np.int64)
    return ((b.astype(np.int64) * s[0] + xyz[:, 0]) * s[1] + xyz[:, 1]) * s[2] + xyz[:, 2]


def subm_rulebook(indices, spatial_shape, ksize=3):
    idx = np.asarray(indices).astype(np.int64)
    n = idx.shape[0]
    shape = np.asarray(spatial_shape, dtype=np.int64)
    key = _lin(idx[:, 0], idx[:, 1:], shape)
    order = np.argsort(key, kind="stable")
    skey = key[order]
    offs = kernel_offsets(ksize)
    pair = np.full((offs.shape[0], n), -1, dtype=np.int32)
    for k, off in enumerate(offs):
        q = idx[:, 1:] + off
        ok = np.all((q >= 0) & (q < shape), axis=1)
        qk = _lin(idx[:, 0], q, shape)
        pos = np.searchsorted(skey, qk)
        pos = np.clip(pos, 0, max(n - 1, 0))
        hit = ok & (skey[pos] == qk) if n else ok
        pair[k, hit] = order[pos[hit]].astype(np.int32)
    return pair


def strided_rulebook(indices, spatial_shape):
    """kernel 2, stride 2, padding 0, dilation 1."""
    idx = np.asarray(indices).astype(np.int64)
    n = idx.shape[0]
    shape = np.asarray(spatial_shape, dtype=np.int64)
    out_shape = shape // 2  # floor((s - 2)/2) + 1
    parent = idx[:, 1:] >> 1
    ok = np.all(parent < out_shape, axis=1)
    k = ((idx[:, 1] & 1) * 2 + (idx[:, 2] & 1)) * 2 + (idx[:, 3] & 1)
    pkey = _lin(idx[:, 0], parent, out_shape)
    ukey, inv = np.unique(pkey[ok], return_inverse=True)
    m = ukey.shape[0]
    out_idx = np.zeros((m, 4), dtype=np.int32)
    r = ukey.copy()
    out_idx[:, 3] = r % out_shape[2]; r //= out_shape[2]
    out_idx[:, 2] = r % out_shape[1]; r //= out_shape[1]
    out_idx[:, 1] = r % out_shape[0]; r //= out_shape[0]
    out_idx[:, 0] = r
    pair_fwd = np.full((8, m), -1, dtype=np.int32)
    pair_bwd = np.full((8, n), -1, dtype=np.int32)
    src = np.nonzero(ok)[0]
    pair_fwd[k[src], inv] = src.astype(np.int32)
    pair_bwd[k[src], src] = inv.astype(np.int32)
    return out_idx, out_shape.tolist(), pair_fwd, pair_bwd
