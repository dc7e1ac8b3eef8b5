"""Known-answer tests for the oracle's kernel maps and sparse convolutions: no spconv exists here,
so the restatement is anchored on torch.nn.functional.conv3d over the densified tensor."""
import numpy as np
import pytest
import torch

import oracle


def _random_sparse(seed, n=300, shape=(12, 10, 9), batch=2, c=5):
    rng = np.random.default_rng(seed)
    cells = rng.choice(batch * shape[0] * shape[1] * shape[2], size=n, replace=False)
    b, r = np.divmod(cells, shape[0] * shape[1] * shape[2])
    x, r = np.divmod(r, shape[1] * shape[2])
    y, z = np.divmod(r, shape[2])
    idx = np.stack([b, x, y, z], 1).astype(np.int32)
    idx = idx[rng.permutation(n)]
    feats = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32))
    return idx, feats


def test_subm_rulebook_properties():
    idx, _ = _random_sparse(0)
    pair = oracle.subm_rulebook(idx, (12, 10, 9))
    n = idx.shape[0]
    assert pair.shape == (27, n)
    assert np.array_equal(pair[13], np.arange(n))
    offs = oracle.kernel_offsets(3)
    for k in range(27):
        j = np.nonzero(pair[k] >= 0)[0]
        i = pair[k, j]
        assert np.array_equal(idx[i, 1:], idx[j, 1:] + offs[k])
        assert np.array_equal(idx[i, 0], idx[j, 0])
        assert np.array_equal(pair[26 - k, i], j)  # centre symmetry


def test_subm_conv_equals_dense_conv3d():
    idx, feats = _random_sparse(1, c=6)
    w = torch.randn(7, 3, 3, 3, 6, generator=torch.Generator().manual_seed(0))
    pair = oracle.subm_rulebook(idx, (12, 10, 9))
    out = oracle.subm_conv(feats, w, pair)
    ref = oracle.dense_equivalent_subm(idx, feats, w, (12, 10, 9), 2)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(12, 10, 8), (11, 9, 7)])
def test_strided_conv_equals_dense_conv3d(shape):
    idx, feats = _random_sparse(2, n=200, shape=shape, c=4)
    w = torch.randn(5, 2, 2, 2, 4, generator=torch.Generator().manual_seed(1))
    oi, oshape, pf, pb = oracle.strided_rulebook(idx, shape)
    assert oshape == [s // 2 for s in shape]
    # canonical order: ascending row-major linear index
    lin = ((oi[:, 0].astype(np.int64) * oshape[0] + oi[:, 1]) * oshape[1] + oi[:, 2]) * oshape[2] + oi[:, 3]
    assert (np.diff(lin) > 0).all()
    out = oracle.strided_conv(feats, w, pf)
    ref = oracle.dense_equivalent_strided(idx, feats, w, shape, 2, oi)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)
    # every kept input has exactly one parent
    kept = (pb >= 0).sum(0)
    dropped = np.any((idx[:, 1:] >> 1) >= np.array(oshape), axis=1)
    assert np.array_equal(kept, (~dropped).astype(kept.dtype))


def test_inverse_conv_is_transpose_of_strided_structure():
    idx, _ = _random_sparse(3, n=150, shape=(8, 8, 8), c=3)
    oi, oshape, pf, pb = oracle.strided_rulebook(idx, (8, 8, 8))
    m = oi.shape[0]
    x = torch.randn(m, 4, generator=torch.Generator().manual_seed(2))
    w = torch.randn(6, 2, 2, 2, 4, generator=torch.Generator().manual_seed(3))
    out = oracle.inverse_conv(x, w, pb)
    # out[i] = W[:, k(i), :] @ x[parent(i)]
    for i in (0, 17, 149):
        k = int(np.nonzero(pb[:, i] >= 0)[0][0])
        ref = w.reshape(6, 8, 4)[:, k, :] @ x[pb[k, i]]
        torch.testing.assert_close(out[i], ref, rtol=1e-5, atol=1e-6)
        # and the child's parity selects k
        assert k == ((idx[i, 1] & 1) * 2 + (idx[i, 2] & 1)) * 2 + (idx[i, 3] & 1)


def test_pdnorm_fold_matches_unfused():
    from oracle import spunet as S
    cfg = dict(S.SPUNET34C, layers=(1,) * 8)
    P = S.make_spunet_params(cfg, seed=3, trained_like=True)
    ctx = P["embedding_table.weight"][1:2]
    x = torch.randn(50, 64, generator=torch.Generator().manual_seed(4))
    ref = S.pdnorm(P, "enc.1.block0.bn1", x, 1, ctx)
    a, b = S.fold_pdnorm(P, "enc.1.block0.bn1", 1, ctx)
    torch.testing.assert_close(x * a + b, ref, rtol=1e-5, atol=1e-6)


def test_argmax_commutes_with_gather():
    g = torch.Generator().manual_seed(5)
    vf = torch.randn(40, 16, generator=g)
    emb = torch.nn.functional.normalize(torch.randn(7, 16, generator=g), dim=1)
    v2p = torch.randint(0, 40, (300,), generator=g)
    per_point = oracle.head_predict(vf, v2p, emb)
    per_voxel = (vf @ emb.t()).max(1)[1][v2p]
    assert torch.equal(per_point, per_voxel)
