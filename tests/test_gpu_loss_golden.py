"""The loss of the training step and the open-vocabulary head on the GPU against vectors recorded from the
reference's OWN classes (tests/golden/reference_loss.npz; mosaic3d/src/models/losses/caption_loss.py:25-49,195-235,
clip_alignment_loss.py:126-147 run untouched by tests/golden/make_golden_loss.py):

* train.caption_loss (per-voxel rows + row map), train.caption_loss_folded and the on-device unique-row routine:
  loss and every gradient, float64 on the device to 1e-9, float32 (what a training step runs) to 2e-4;
* SpUNetEngine.predict(return_logits=True): the per-voxel logits at the sampled points within 1e-4 of the logits the
  reference's CLIPAlignmentEval.predict forms from the reference's own forward pass, labels >= 99.9 % of its
  logits.max(1)[1] (language_module.py:328), and every disagreement inside a margin the tolerance explains."""
import numpy as np
import pytest
import torch

import _lossgolden as lg
from oracle import spunet as ospunet
from segmentation_amd import scenes, train
from segmentation_amd.engine import SpUNetEngine

pytestmark = pytest.mark.gpu
# the forward generator's constants (tests/golden/make_golden_forward.py)
SEEDS, TARGET, PARAM_SEED, CONDITION, CLASSES = [21, 22], 5000, 3, "Structured3D", 20


def _rel(a, ref):
    ref = torch.as_tensor(ref, dtype=torch.float64).reshape(-1)
    return ((a.detach().double().cpu().reshape(-1) - ref).norm() / ref.norm().clamp_min(1e-300)).item()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-4)])
@pytest.mark.parametrize("name", list(lg.SETTINGS))
@pytest.mark.parametrize("folded", [False, True])
def test_loss_on_the_device_equals_the_reference(device, name, folded, dtype, tol):
    g = lg.load()
    b = lg.batch(g, device=device, dtype=dtype)
    normalize, scaled, reduction = lg.SETTINGS[name]
    pre, w, bias = b["pre"].requires_grad_(True), b["w"].requires_grad_(True), b["b"].requires_grad_(True)
    scale = lg.scale_of(g, name, device, dtype) if scaled else None
    text, seg_labels, first = train.unique_text_features(b["embeddings"])   # the hash-grouping routine of the GPU
    assert np.array_equal(seg_labels.cpu().numpy(), g["labels_per_segment"])
    assert np.array_equal(first.cpu().numpy(), g["labels_per_caption"])
    assert np.array_equal(text.double().cpu().numpy(), g["text_features"].astype(np.float32 if dtype == torch.float32
                                                                                 else np.float64).astype(np.float64))
    if folded:
        loss = train.caption_loss_folded(pre, w, bias, text, seg_labels, b["point_indices"], b["caption_offsets"],
                                         b["counts"], normalize=normalize, logit_scale=scale, reduction=reduction,
                                         row_map=b["v2p"])
    else:
        loss = train.caption_loss(pre @ w.T + bias, text, seg_labels, b["point_indices"], b["caption_offsets"],
                                  b["counts"], normalize=normalize, logit_scale=scale, reduction=reduction,
                                  row_map=b["v2p"])
    leaves = [t for t in (pre, w, bias, scale) if t is not None]
    grads = torch.autograd.grad(loss, leaves)
    assert abs(loss.item() - g[f"{name}_loss"][0]) < tol * abs(g[f"{name}_loss"][0])
    for got, key in zip(grads, ["dpre", "dw", "db", "dscale"]):
        assert _rel(got, g[f"{name}_{key}"]) < tol, (key, _rel(got, g[f"{name}_{key}"]))


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("bf16x6", 1e-4), ("bf16x3", 1e-4)])
@pytest.mark.parametrize("fold_head", [False, True])
def test_engine_logits_and_labels_equal_the_reference_head(device, prec, tol, fold_head):
    g = lg.load()
    cfg = ospunet.SPUNET34C
    P = ospunet.make_spunet_params(cfg, seed=PARAM_SEED, trained_like=True)
    b = scenes.make_batch(SEEDS, target_voxels=TARGET)
    sd = {("backbone." + k if not k.startswith("embedding_table") else k): v for k, v in P.items()}
    eng = SpUNetEngine(sd, cfg, CONDITION, device, precision=prec)
    eng.set_text_embeddings(torch.from_numpy(scenes.text_embeddings(CLASSES, cfg["out_channels"])))
    out = eng.predict(torch.from_numpy(b["coord"]).to(device), torch.from_numpy(b["feat"]).to(device), b["offset"], 0.02,
                      return_logits=True, fold_head=fold_head)
    assert out["pred"].shape[0] == int(g["head_n_points"][0])
    rows = torch.from_numpy(g["head_rows"]).to(device)
    got = out["voxel_logits"][out["v2p_map"][rows]].double().cpu()
    ref = torch.from_numpy(g["head_raw_logits"])
    err = ((got - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
    assert err < tol, err
    pred = out["pred"].cpu().numpy()
    want = g["head_raw_max"].astype(np.int64)
    assert np.array_equal(want, g["head_raw_pred"].astype(np.int64))   # argmax(dim=1) and max(1)[1] of the reference
    differ = pred != want
    assert 1.0 - differ.mean() >= 0.999, differ.mean()
    # a label may differ only where the reference's two best logits are closer than the tolerance allows to tell
    scale = float(np.abs(g["head_raw_logits"]).max())
    assert (g["head_raw_margin"][differ] < 10 * tol * scale).all()
    # normalising the rows first (normalize_input=True) scales each row by a positive number: the same labels up to ties
    norm_differs = (g["head_norm_max"] != g["head_raw_max"]).mean()
    assert norm_differs < 1e-3, norm_differs
