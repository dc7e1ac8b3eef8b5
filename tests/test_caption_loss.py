"""caption_loss picks one log-probability per (point, caption) pair; the reference gathers scores[point_indices]
([L, U]), averages per caption with segment_csr and lets nll_loss take the caption's column
(mosaic3d/src/models/losses/caption_loss.py:212-235).  Both forms, in float64 on the CPU, must agree."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from segmentation_amd import train
from segmentation_amd.sparse import segment_csr


def _reference_form(feats, text, labels, point_indices, offsets, weights, normalize, logit_scale, reduction):
    if normalize:
        feats = F.normalize(feats, dim=-1)
    logits = feats @ text.T
    if logit_scale is not None:
        logits = logit_scale.exp() * logits
    scores = F.log_softmax(logits, dim=-1)
    reduced = segment_csr(scores[point_indices], offsets, reduce="mean")
    loss = F.nll_loss(reduced, labels, reduction="none")
    if reduction == "mean":
        return loss.mean()
    return (loss * weights).sum() / weights.sum()


@pytest.mark.parametrize("reduction", ["weighted_sum", "mean"])
@pytest.mark.parametrize("normalize,scaled", [(True, False), (False, True)])
def test_column_pick_equals_gather_segment_mean_nll(reduction, normalize, scaled):
    rng = np.random.default_rng(5)
    n, d, u = 300, 16, 7
    feats = torch.from_numpy(rng.normal(size=(n, d))).requires_grad_(True)
    text = F.normalize(torch.from_numpy(rng.normal(size=(u, d))), dim=1)
    counts = torch.tensor([40, 1, 0, 120, 33, 0, 64])  # captions without points have mean 0 in segment_csr
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)])
    point_indices = torch.from_numpy(rng.integers(0, n, size=int(counts.sum())))
    labels = torch.from_numpy(rng.integers(0, u, size=counts.numel()))
    weights = torch.tensor([40.0, 3.0, 2.0, 100.0, 33.0, 1.0, 64.0], dtype=torch.float64)  # not the segment lengths
    scale = torch.tensor(np.log(1 / 0.07), dtype=torch.float64, requires_grad=True) if scaled else None
    got = train.caption_loss(feats, text, labels, point_indices, offsets, weights, normalize=normalize,
                             logit_scale=scale, reduction=reduction)
    got.backward()
    g_feats, g_scale = feats.grad.clone(), None if scale is None else scale.grad.clone()
    feats.grad = None
    if scale is not None:
        scale.grad = None
    ref = _reference_form(feats, text, labels, point_indices, offsets, weights, normalize, scale, reduction)
    ref.backward()
    assert abs(got.item() - ref.item()) < 1e-12 * max(1.0, abs(ref.item()))
    assert torch.allclose(g_feats, feats.grad, rtol=1e-10, atol=1e-14)
    if scale is not None:
        assert torch.allclose(g_scale, scale.grad, rtol=1e-10, atol=1e-14)


def test_row_map_and_unknown_reduction():
    rng = np.random.default_rng(6)
    vox = torch.from_numpy(rng.normal(size=(50, 8)))
    v2p = torch.from_numpy(rng.integers(0, 50, size=200))
    text = F.normalize(torch.from_numpy(rng.normal(size=(4, 8))), dim=1)
    offsets = torch.tensor([0, 30, 90])
    pi = torch.from_numpy(rng.integers(0, 200, size=90))
    labels, w = torch.tensor([1, 3]), torch.tensor([30.0, 60.0], dtype=torch.float64)
    a = train.caption_loss(vox[v2p], text, labels, pi, offsets, w)
    b = train.caption_loss(vox, text, labels, pi, offsets, w, row_map=v2p)
    assert abs(a.item() - b.item()) < 1e-13
    with pytest.raises(ValueError):
        train.caption_loss(vox, text, labels, pi, offsets, w, row_map=v2p, reduction="sum")


@pytest.mark.parametrize("normalize,with_bias,scaled", [(True, True, False), (True, False, True), (False, True, True)])
def test_folded_last_layer_equals_loss_of_its_output(normalize, with_bias, scaled, monkeypatch):
    """caption_loss_folded(pre, W, b, ...) == caption_loss(pre @ W.T + b, ...) with all gradients, in float64."""
    rng = np.random.default_rng(11)
    n, cin, d, u = 400, 12, 40, 9
    monkeypatch.setattr(train._TallMatmul, "RANGES", 3)  # 400 rows = 3 ranges of 133 + a tail of 1
    pre = torch.from_numpy(rng.normal(size=(n, cin))).requires_grad_(True)
    w = torch.from_numpy(rng.normal(size=(d, 1, 1, 1, cin)) * 0.3).requires_grad_(True)  # the 1x1x1 kernel's layout
    b = torch.from_numpy(rng.normal(size=(d,)) * 0.2).requires_grad_(True) if with_bias else None
    scale = torch.tensor(np.log(1 / 0.07), dtype=torch.float64, requires_grad=True) if scaled else None
    text = F.normalize(torch.from_numpy(rng.normal(size=(u, d))), dim=1)
    v2p = torch.from_numpy(rng.integers(0, n, size=1500))
    counts = torch.tensor([200, 0, 31, 77, 1, 400])
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)])
    pi = torch.from_numpy(rng.integers(0, 1500, size=int(counts.sum())))
    labels = torch.from_numpy(rng.integers(0, u, size=counts.numel()))
    weights = counts.double() + 1.0
    leaves = [t for t in (pre, w, b, scale) if t is not None]
    got = train.caption_loss_folded(pre, w, b, text, labels, pi, offsets, weights, normalize=normalize,
                                    logit_scale=scale, row_map=v2p)
    g_got = torch.autograd.grad(got, leaves)
    feats = pre @ w.reshape(d, cin).T
    if b is not None:
        feats = feats + b
    ref = train.caption_loss(feats, text, labels, pi, offsets, weights, normalize=normalize, logit_scale=scale,
                             row_map=v2p)
    g_ref = torch.autograd.grad(ref, leaves)
    assert abs(got.item() - ref.item()) < 1e-12 * max(1.0, abs(ref.item()))
    for a, r in zip(g_got, g_ref):
        assert torch.allclose(a, r, rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("tie_on_leading_columns", [False, True])
def test_unique_rows_equals_numpy_unique_axis0(tie_on_leading_columns):
    """The device routine of unique_text_features (hash grouping + leading-column order, verified, with a
    torch.unique(dim=0) fallback) is plain torch code: run it on the CPU against np.unique(axis=0)."""
    rng = np.random.default_rng(9)
    pool = torch.from_numpy(rng.normal(size=(40, 12)).astype(np.float32))
    pool[3, 0] = -0.0
    pool[4] = pool[3]
    pool[4, 0] = 0.0  # equal to row 3 for np.unique, different bits
    if tie_on_leading_columns:
        pool[6, :4] = pool[5, :4]
    text = pool[torch.from_numpy(rng.integers(0, 40, size=300))]
    first, inverse = train._unique_rows_on_device(text)
    uniq, f_ref, i_ref = np.unique(text.numpy(), return_index=True, return_inverse=True, axis=0)
    assert np.array_equal(text[first].numpy(), uniq)
    assert np.array_equal(inverse.numpy(), np.asarray(i_ref).reshape(-1))
    assert np.array_equal(first.numpy(), f_ref)
