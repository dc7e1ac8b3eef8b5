"""CPU restatement of spconv's Native (gather -> mm -> scatter-add per kernel offset) convolution.
Test infrastructure only.

Call sites in the reference: spconv_unet_v1m3_pdnorm.py:98-117 (SubMConv3d), :154-161
(SparseConv3d k2 s2), :184-190 (SparseInverseConv3d); structure.py:336-347 (Custom1x1Subm3d).
Weight layout [Cout, k0, k1, k2, Cin]; correlation orientation (same as nn.Conv3d with the
weight permuted to [Cout, Cin, k0, k1, k2]).  Parity unpinned by reference tests; anchored by
the dense conv3d known-answer tests.
"""
import numpy as np
import torch


def _native(feats, weight, pair, n_out):
    """out[o] = sum_k W[:, k, :] @ feats[pair[k, o]]  for pair[k, o] >= 0."""
    cout = weight.shape[0]
    cin = weight.shape[-1]
    w = weight.reshape(cout, -1, cin)
    out = torch.zeros(n_out, cout, dtype=feats.dtype)
    pair_t = torch.as_tensor(np.asarray(pair), dtype=torch.int64)
    for k in range(w.shape[1]):
        p = pair_t[k]
        o = torch.nonzero(p >= 0).squeeze(1)
        if o.numel() == 0:
            continue
        out.index_add_(0, o, feats.index_select(0, p[o]) @ w[:, k, :].T)
    return out


def subm_conv(feats, weight, pair, bias=None):
    out = _native(feats, weight, pair, feats.shape[0])
    return out if bias is None else out + bias


def strided_conv(feats, weight, pair_fwd, bias=None):
    out = _native(feats, weight, pair_fwd, pair_fwd.shape[1])
    return out if bias is None else out + bias


def inverse_conv(feats, weight, pair_bwd, bias=None):
    """out[i] = W[:, k, :] @ feats[pair_bwd[k, i]]; output rows = inputs of the matching strided conv."""
    out = _native(feats, weight, pair_bwd, pair_bwd.shape[1])
    return out if bias is None else out + bias


def conv1x1(feats, weight, bias=None):
    """structure.py:337-341."""
    out = feats @ weight.reshape(weight.shape[0], weight.shape[-1]).T
    return out if bias is None else out + bias


def dense_equivalent_subm(indices, feats, weight, spatial_shape, batch_size):
    """Known-answer: conv3d(pad=1) on the densified tensor, sampled at the active sites."""
    idx = torch.as_tensor(np.asarray(indices), dtype=torch.int64)
    c = feats.shape[1]
    dense = torch.zeros(batch_size, c, *spatial_shape, dtype=torch.float64)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feats.double()
    w = weight.double().permute(0, 4, 1, 2, 3)
    y = torch.nn.functional.conv3d(dense, w, padding=weight.shape[1] // 2)
    return y[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]


def dense_equivalent_strided(indices, feats, weight, spatial_shape, batch_size, out_indices):
    idx = torch.as_tensor(np.asarray(indices), dtype=torch.int64)
    oidx = torch.as_tensor(np.asarray(out_indices), dtype=torch.int64)
    c = feats.shape[1]
    dense = torch.zeros(batch_size, c, *spatial_shape, dtype=torch.float64)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feats.double()
    w = weight.double().permute(0, 4, 1, 2, 3)
    y = torch.nn.functional.conv3d(dense, w, stride=2)
    return y[oidx[:, 0], :, oidx[:, 1], oidx[:, 2], oidx[:, 3]]


def avgpool(feats, pair, n_out=None):
    """indice_avgpool_implicit_gemm: mean of feats[pair[k, o]] over the valid k (zero for rows without pairs).
    spconv itself is absent: parity unpinned, this is the definition the GPU op is held to."""
    pair_t = torch.as_tensor(np.asarray(pair), dtype=torch.int64)
    n_out = pair_t.shape[1] if n_out is None else n_out
    out = torch.zeros(n_out, feats.shape[1], dtype=torch.float64)
    cnt = torch.zeros(n_out, dtype=torch.float64)
    for k in range(pair_t.shape[0]):
        p = pair_t[k, :n_out]
        o = torch.nonzero(p >= 0).squeeze(1)
        out.index_add_(0, o, feats.double().index_select(0, p[o]))
        cnt.index_add_(0, o, torch.ones(o.numel(), dtype=torch.float64))
    return out / cnt.clamp(min=1).unsqueeze(1), cnt.long()
