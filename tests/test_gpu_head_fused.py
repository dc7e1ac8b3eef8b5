"""Fused head (last 1x1 layer + text logits + argmax) against a float64 restatement and the unfused kernels.
Bars: logits within 1e-4 relative per row with products from 2-term bf16 splits, within 2e-6 with 3-term splits
(f32 level) and within 2e-5 with f16 pairs (format 4: the features are exact to 2^-22, the logits product uses unscaled
second terms because it only feeds the argmax), labels identical to the first-index argmax of the kernel's own logits and >= 99.9 % equal to the
float64 argmax."""
import numpy as np
import pytest
import torch

from segmentation_amd import ops

pytestmark = pytest.mark.gpu


def _case(device, n, cin, cmid, ncls, seed, affine=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, generator=g)
    w1 = torch.randn(cmid, 1, cin, generator=g) / cin ** 0.5
    text = torch.nn.functional.normalize(torch.randn(ncls, cmid, generator=g), dim=1)
    scale = (1.0 + 0.1 * torch.randn(cmid, generator=g)) if affine else None
    shift = 0.1 * torch.randn(cmid, generator=g)
    feat = x.double() @ w1[:, 0].double().T
    if scale is not None:
        feat = feat * scale.double()
    feat = feat + shift.double()
    ref = feat @ text.double().T
    dev = lambda t: None if t is None else t.to(device)
    return dict(x=dev(x), w1=dev(w1), text=dev(text), scale=dev(scale), shift=dev(shift), ref=ref)


@pytest.mark.parametrize("n,cin,cmid,ncls", [(1000, 96, 768, 200), (129, 96, 768, 200), (4096, 64, 64, 20),
                                             (777, 128, 256, 128), (300, 32, 96, 129), (2000, 96, 768, 256),
                                             (513, 96, 512, 20), (700, 96, 768, 150), (640, 128, 1024, 256),
                                             (900, 64, 768, 200), (70000, 96, 768, 200),
                                             (40000, 32, 64, 100), (1, 96, 768, 200),
                                             (31, 96, 32, 1), (33, 64, 32, 2)])
@pytest.mark.parametrize("terms", [2, 3, 4])
def test_fused_head_matches_float64(device, n, cin, cmid, ncls, terms):
    c = _case(device, n, cin, cmid, ncls, seed=n, affine=(n % 2 == 1) or terms == 2)
    xs = ops.split_f32(c["x"], terms)
    w1 = ops.pack_weight(c["w1"], ops.PRECISION_OF_TERMS[terms], device)
    tp = ops.head_pack_text(c["text"], cin=cin, terms=terms)
    label, logits = ops.head_fused(xs, w1, tp, scale=c["scale"], shift=c["shift"], want_logits=True)
    got = logits[:, :ncls].double().cpu()
    rel = ((got - c["ref"]).norm(dim=1) / c["ref"].norm(dim=1)).max().item()
    assert rel < {2: 1e-4, 3: 2e-6, 4: 2e-5}[terms], rel
    assert logits.shape[1] == tp.classes_padded and tp.classes_padded % 32 == 0
    assert torch.equal(label.cpu().long(), logits[:, :ncls].cpu().argmax(dim=1))
    agree = (label.cpu().long() == c["ref"].argmax(dim=1)).float().mean().item()
    assert agree >= 0.999, agree
    # labels alone (no logits written) are the same labels
    label2, none = ops.head_fused(xs, w1, tp, scale=c["scale"], shift=c["shift"])
    assert none is None and torch.equal(label2, label)


@pytest.mark.parametrize("terms", [2, 3, 4])
def test_fused_head_agrees_with_unfused_kernels(device, terms):
    prec = ops.PRECISION_OF_TERMS[terms]
    c = _case(device, 5000, 96, 768, 200, seed=5, affine=False)
    xs = ops.split_f32(c["x"], terms)
    w1 = ops.pack_weight(c["w1"], prec, device)
    label, logits = ops.head_fused(xs, w1, ops.head_pack_text(c["text"], terms=terms), shift=c["shift"], want_logits=True)
    _, fsplit = ops.conv_fwd(xs, w1, None, 5000, shift=c["shift"], want_f32=False, split_terms=terms)
    wt = torch.zeros(256, 1, 768, device=device)
    wt[:200, 0] = c["text"]
    ulog = ops.conv_fwd(fsplit, ops.pack_weight(wt, prec, device), None, 5000)
    ulab = ops.argmax_rows(ulog, 200)
    rel = ((logits[:, :200] - ulog[:, :200]).norm(dim=1) / ulog[:, :200].norm(dim=1)).max().item()
    assert rel < {2: 2e-5, 3: 1e-6, 4: 5e-6}[terms], rel
    assert (label == ulab).float().mean().item() >= 0.9995


@pytest.mark.parametrize("terms", [2, 3, 4])
def test_fused_head_ties_take_the_first_class(device, terms):
    c = _case(device, 256, 96, 768, 200, seed=9)
    text = c["text"].clone()
    text[150] = text[3]          # classes 3 and 150 score identically on every row
    text[199] = text[0]
    xs = ops.split_f32(c["x"], terms)
    w1 = ops.pack_weight(c["w1"], ops.PRECISION_OF_TERMS[terms], device)
    label, logits = ops.head_fused(xs, w1, ops.head_pack_text(text, terms=terms), scale=c["scale"], shift=c["shift"],
                                   want_logits=True)
    lab = label.cpu().numpy()
    assert not np.isin(lab, [150, 199]).any()
    assert torch.equal(logits[:, 150], logits[:, 3])
    assert torch.equal(label.cpu().long(), logits[:, :200].cpu().argmax(dim=1))


def test_fused_head_rejects_unsupported_shapes(device):
    x = ops.split_f32(torch.randn(64, 160, device=device), 2)
    w1 = ops.pack_weight(torch.randn(64, 1, 160, device=device), "bf16x3", device)
    tp = ops.head_pack_text(torch.randn(10, 64, device=device), cin=160)
    with pytest.raises(RuntimeError, match="no kernel for 160 input channels"):
        ops.head_fused(x, w1, tp)
