"""The caption loss and the open-vocabulary head against vectors produced by the reference's OWN classes
(tests/golden/reference_loss.npz, written by tests/golden/make_golden_loss.py from
mosaic3d/src/models/losses/caption_loss.py:25-49,195-235 and clip_alignment_loss.py:126-147):

* the CPU oracle (oracle/caption_loss.py, oracle/head.py) reproduces them, which pins it;
* the product's loss (train.caption_loss with a row map, train.caption_loss_folded, train.unique_text_features,
  train.flatten_caption_indices: plain torch code, run here in float64 on the CPU) reproduces them, values and
  gradients; the GPU run of the same functions is tests/test_gpu_loss_golden.py."""
import numpy as np
import pytest
import torch

import _lossgolden as lg
from oracle import caption_loss as oloss
from segmentation_amd import train


@pytest.fixture(scope="module")
def golden():
    return lg.load()


def _grads(loss, leaves):
    return torch.autograd.grad(loss, [l for l in leaves if l is not None])


@pytest.mark.parametrize("name", list(lg.SETTINGS))
def test_oracle_loss_equals_the_reference(golden, name):
    g, b = golden, lg.batch(golden)
    normalize, scaled, reduction = lg.SETTINGS[name]
    pre, w, bias = b["pre"].requires_grad_(True), b["w"].requires_grad_(True), b["b"].requires_grad_(True)
    scale = lg.scale_of(g, name) if scaled else None
    feats = pre @ w.T + bias
    loss = oloss.training_loss(feats, b["v2p"], b["embeddings"], b["point_indices"], b["caption_offsets"], b["counts"],
                               normalize=normalize, logit_scale=scale, reduction=reduction)
    assert abs(loss.item() - g[f"{name}_loss"][0]) < 1e-13
    grads = _grads(loss, [pre, w, bias, scale])
    for got, key in zip(grads, ["dpre", "dw", "db", "dscale"]):
        assert np.allclose(got.detach().numpy().reshape(-1), g[f"{name}_{key}"].reshape(-1), rtol=1e-10, atol=1e-15), key


def test_oracle_text_bookkeeping_equals_the_reference(golden):
    g, b = golden, lg.batch(golden)
    text, per_seg, per_cap = oloss.extract_text_features(b["embeddings"])
    assert np.array_equal(text.numpy(), g["text_features"])
    assert np.array_equal(per_seg.numpy(), g["labels_per_segment"])
    assert np.array_equal(per_cap.numpy(), g["labels_per_caption"])
    # the product's version (numpy on the CPU, hash grouping on the GPU)
    t2, s2, c2 = train.unique_text_features(b["embeddings"])
    assert np.array_equal(t2.numpy(), g["text_features"])
    assert np.array_equal(s2.numpy(), g["labels_per_segment"])
    assert np.array_equal(c2.numpy(), g["labels_per_caption"])


def test_flatten_caption_indices_equals_the_collate_form(golden):
    b = lg.batch(golden)
    per_scene = [[l for l, s in zip(b["local_indices"], b["scene"]) if s == sc] for sc in range(2)]
    pi, offs, counts = train.flatten_caption_indices(per_scene, torch.from_numpy(b["offset"]))
    assert torch.equal(pi, b["point_indices"]) and torch.equal(offs, b["caption_offsets"])
    assert torch.equal(counts, b["counts"])


@pytest.mark.parametrize("name", list(lg.SETTINGS))
@pytest.mark.parametrize("folded", [False, True])
def test_product_loss_equals_the_reference(golden, name, folded):
    g, b = golden, lg.batch(golden)
    normalize, scaled, reduction = lg.SETTINGS[name]
    pre, w, bias = b["pre"].requires_grad_(True), b["w"].requires_grad_(True), b["b"].requires_grad_(True)
    scale = lg.scale_of(g, name) if scaled else None
    text, seg_labels, _ = train.unique_text_features(b["embeddings"])
    if folded:
        loss = train.caption_loss_folded(pre, w, bias, text, seg_labels, b["point_indices"], b["caption_offsets"],
                                         b["counts"], normalize=normalize, logit_scale=scale, reduction=reduction,
                                         row_map=b["v2p"])
    else:
        loss = train.caption_loss(pre @ w.T + bias, text, seg_labels, b["point_indices"], b["caption_offsets"],
                                  b["counts"], normalize=normalize, logit_scale=scale, reduction=reduction,
                                  row_map=b["v2p"])
    assert abs(loss.item() - g[f"{name}_loss"][0]) < 1e-12
    grads = _grads(loss, [pre, w, bias, scale])
    for got, key in zip(grads, ["dpre", "dw", "db", "dscale"]):
        assert np.allclose(got.detach().numpy().reshape(-1), g[f"{name}_{key}"].reshape(-1), rtol=1e-8, atol=1e-13), key
