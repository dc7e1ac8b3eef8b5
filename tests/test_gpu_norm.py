"""Batch-norm reductions on the GPU (m3d_bn_stats, m3d_bn_backward_reduce through the C ABI) and the training-mode
PDBatchNorm built on them, against float64 torch on the CPU.  Bars: statistics 1e-6 relative, outputs and gradients
1e-5 relative L2 (float32 elementwise passes)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from segmentation_amd import ops, spunet

pytestmark = pytest.mark.gpu


def _rel(a, b):
    b = b.double()
    return ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("n,c", [(1, 32), (7, 4), (1000, 96), (65537, 32), (30001, 256), (5000, 1024)])
def test_column_reductions(device, n, c):
    g = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, c, generator=g) * 3.0 + torch.linspace(-50, 50, c)  # large means: E[x^2] - mean^2 must hold up
    dy = torch.randn(n, c, generator=g)
    mean, var = ops.bn_stats(x.to(device))
    xd = x.double()
    assert _rel(mean, xd.mean(0)) < 1e-6
    assert _rel(var, xd.var(0, unbiased=False)) < 1e-5
    s1, s2 = ops.bn_backward_reduce(x.to(device), dy.to(device), mean)
    mu = mean.cpu().double()
    assert _rel(s1, dy.double().sum(0)) < 1e-6
    assert _rel(s2, (dy.double() * (xd - mu)).sum(0)) < 1e-6


def test_reductions_are_deterministic_and_reject_bad_shapes(device):
    x = torch.randn(50000, 96, device=device)
    a, b = ops.bn_stats(x), ops.bn_stats(x)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    with pytest.raises(Exception):
        ops.bn_stats(torch.randn(100, 6, device=device))  # channels not a multiple of 4


@pytest.mark.parametrize("n", [300, 40000])  # below / above the row count from which the kernels are used
def test_training_mode_pdnorm_matches_float64(device, n):
    torch.manual_seed(3)
    m = spunet.PDBatchNorm(64, adaptive=True)
    for bn in m.bns:
        bn.weight.data.normal_()
        bn.bias.data.normal_()
    ref = copy.deepcopy(m).double()
    m = m.to(device).train()
    ref.train()
    x = torch.randn(n, 64) * 2.0 + 1.0
    ctx = torch.randn(1, 256)
    gout = torch.randn(n, 64)
    xg = x.to(device).requires_grad_(True)
    y = m(xg, "ScanNet", ctx.to(device))
    y.backward(gout.to(device))
    xr = x.double().requires_grad_(True)
    yr = ref(xr, "ScanNet", ctx.double())
    yr.backward(gout.double())
    assert _rel(y.detach(), yr.detach()) < 1e-5
    assert _rel(xg.grad, xr.grad) < 1e-5
    for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if q.grad is not None:
            assert _rel(p.grad, q.grad) < 1e-4, name
    bn, bnr = m.pick("ScanNet"), ref.pick("ScanNet")
    assert _rel(bn.running_mean, bnr.running_mean) < 1e-5 and _rel(bn.running_var, bnr.running_var) < 1e-5
    assert bn.num_batches_tracked.item() == 1


@pytest.mark.parametrize("n", [300, 40000])
@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (True, False)])
def test_pdnorm_with_residual_and_relu_matches_float64(device, n, with_res, relu):
    """relu(bn(x) + residual) in one pass each way (BasicBlock.forward, spconv_unet_v1m3_pdnorm.py:121-140) against
    the same expression in float64 on the CPU; the row count selects the kernel path or torch's."""
    torch.manual_seed(5)
    m = spunet.PDBatchNorm(96, adaptive=True)
    for bn in m.bns:
        bn.weight.data.normal_()
        bn.bias.data.normal_()
    m.modulation[-1].weight.data.normal_(std=0.05)
    ref = copy.deepcopy(m).double().train()
    m = m.to(device).train()
    x, res = torch.randn(n, 96) * 1.5 - 0.5, torch.randn(n, 96)
    ctx, gout = torch.randn(1, 256), torch.randn(n, 96)
    xg, rg = x.to(device).requires_grad_(True), res.to(device).requires_grad_(True)
    y = m(xg, "Matterport3D", ctx.to(device), residual=rg if with_res else None, relu=relu)
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    yr = ref(xr, "Matterport3D", ctx.double())
    if with_res:
        yr = yr + rr
    if relu:
        # an element within rounding distance of the ReLU's kink may fall on either side in f32: no gradient there
        gout = gout * (yr.detach().abs() > 1e-4).float()
        yr = F.relu(yr)
    y.backward(gout.to(device))
    yr.backward(gout.double())
    assert _rel(y.detach(), yr.detach()) < 1e-5
    assert _rel(xg.grad, xr.grad) < 1e-5
    if with_res:
        assert _rel(rg.grad, rr.grad) < 1e-6
    for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if q.grad is not None:
            assert _rel(p.grad, q.grad) < 1e-4, name
