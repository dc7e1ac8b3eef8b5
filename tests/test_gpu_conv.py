"""GPU sparse convolution (m3d_conv_fwd_f32 through the C ABI) against the CPU oracle.
Tolerance: fp32, per-row relative L2 error <= 1e-5 for one layer (the network-level bar is 1e-4)."""
import numpy as np
import pytest
import torch

import oracle
from segmentation_amd import ops, scenes

pytestmark = pytest.mark.gpu


TOL = {"f32": 1e-5, "bf16x6": 1e-5, "bf16x3": 6e-5, "bf16": 1e-2}


def _rel(out, ref):
    ref = ref.double()
    return ((out.double().cpu() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-12)).max().item()


@pytest.fixture(scope="module")
def level(device):
    sc = scenes.make_scene(seed=4, target_voxels=12000)
    off = np.array([0, sc["coord"].shape[0]], dtype=np.int64)
    sp = oracle.sparsify(sc["coord"], sc["feat"], off, 0.02)
    idx = torch.from_numpy(sp["voxel_coords"]).to(device)
    pair = ops.subm_rulebook(idx, sp["spatial_shape"])
    sched = ops.build_schedule(pair, idx.shape[0], idx, 1, sp["spatial_shape"], 1)
    oi, oshape, pf, pb = ops.strided_rulebook(idx, sp["spatial_shape"], 1)
    return dict(sp=sp, idx=idx, pair=pair.cpu().numpy(), sched=sched, oi=oi, pf=pf, pb=pb,
                sched_down=ops.build_schedule(pf, oi.shape[0], None, 0),
                sched_up=ops.build_schedule(pb, idx.shape[0], None, 2, base_order=sched.out_row))


# (cin, cout): every SubM shape of spunet34c plus odd shapes for the direct kernel
@pytest.mark.parametrize("prec", ["f32", "bf16x6", "bf16x3", "bf16"])
@pytest.mark.parametrize("cin,cout", [(3, 32), (32, 32), (64, 64), (96, 96), (128, 96), (128, 128),
                                        (192, 128), (256, 256), (384, 256), (6, 32), (40, 24)])
def test_subm_conv(device, level, cin, cout, prec):
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    n = level["idx"].shape[0]
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, 3, 3, 3, cin, generator=g) / np.sqrt(9 * cin)
    ref = oracle.subm_conv(x, w, level["pair"])
    out = ops.conv_fwd(x.to(device), ops.pack_weight(w, prec, device), level["sched"], n)
    assert _rel(out, ref) < TOL[prec]


@pytest.mark.parametrize("prec", ["f32", "bf16x6", "bf16x3", "bf16"])
def test_subm_conv_fused_epilogue_and_two_sources(device, level, prec):
    g = torch.Generator().manual_seed(7)
    n = level["idx"].shape[0]
    xa, xb = torch.randn(n, 96, generator=g), torch.randn(n, 32, generator=g)
    w = torch.randn(96, 3, 3, 3, 128, generator=g) / np.sqrt(9 * 128)
    scale, shift = torch.rand(96, generator=g) + 0.5, torch.randn(96, generator=g)
    res = torch.randn(n, 96, generator=g)
    ref = torch.relu(oracle.subm_conv(torch.cat((xa, xb), 1), w, level["pair"]) * scale + shift + res)
    out = ops.conv_fwd(xa.to(device), ops.pack_weight(w, prec, device), level["sched"], n, in1=xb.to(device),
                       scale=scale.to(device), shift=shift.to(device), residual=res.to(device), relu=True)
    assert (out.cpu() - ref).abs().max().item() < 10 * TOL[prec]
    assert _rel(out, ref) < TOL[prec]


@pytest.mark.parametrize("prec", ["f32", "bf16x6", "bf16x3", "bf16"])
@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 128), (128, 256)])
def test_strided_and_inverse_conv(device, level, cin, cout, prec):
    g = torch.Generator().manual_seed(cin + cout)
    n, m = level["idx"].shape[0], level["oi"].shape[0]
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, 2, 2, 2, cin, generator=g) / np.sqrt(8 * cin)
    ref = oracle.strided_conv(x, w, level["pf"].cpu().numpy())
    out = ops.conv_fwd(x.to(device), ops.pack_weight(w, prec, device), level["sched_down"], m)
    assert _rel(out, ref) < TOL[prec]
    y = torch.randn(m, cout, generator=g)
    wi = torch.randn(cin, 2, 2, 2, cout, generator=g) / np.sqrt(cout)
    ref_i = oracle.inverse_conv(y, wi, level["pb"].cpu().numpy())
    out_i = ops.conv_fwd(y.to(device), ops.pack_weight(wi, prec, device), level["sched_up"], n)
    assert _rel(out_i, ref_i) < TOL[prec]


@pytest.mark.parametrize("prec", ["f32", "bf16x6", "bf16x3", "bf16"])
@pytest.mark.parametrize("n,cin,cout", [(5000, 96, 768), (700, 384, 256), (130, 128, 96), (1, 96, 32), (70000, 128, 96)])
def test_dense_1x1_with_bias(device, n, cin, cout, prec):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, 1, 1, 1, cin, generator=g) / np.sqrt(cin)
    b = torch.randn(cout, generator=g)
    ref = oracle.conv1x1(x, w, b)
    out = ops.conv_fwd(x.to(device), ops.pack_weight(w, prec, device), None, n, shift=b.to(device))
    assert _rel(out, ref) < TOL[prec]


@pytest.mark.parametrize("prec", ["f32", "bf16x6", "bf16x3", "bf16"])
@pytest.mark.parametrize("target", [300, 1500, 30000])
def test_all_tile_configurations(device, target, prec):
    """Small, medium and large levels pick different tile shapes / split-K; all must agree with the oracle."""
    sc = scenes.make_scene(seed=9, target_voxels=target)
    off = np.array([0, sc["coord"].shape[0]], dtype=np.int64)
    sp = oracle.sparsify(sc["coord"], sc["feat"], off, 0.02)
    idx = torch.from_numpy(sp["voxel_coords"]).to(device)
    n = idx.shape[0]
    pair = ops.subm_rulebook(idx, sp["spatial_shape"])
    sched = ops.build_schedule(pair, n, idx, 1, sp["spatial_shape"], 1)
    for cin, cout in [(32, 32), (64, 64), (96, 96), (128, 128), (256, 256)]:
        g = torch.Generator().manual_seed(cin)
        x = torch.randn(n, cin, generator=g)
        w = torch.randn(cout, 3, 3, 3, cin, generator=g) / np.sqrt(9 * cin)
        ref = oracle.subm_conv(x, w, pair.cpu().numpy())
        out = ops.conv_fwd(x.to(device), ops.pack_weight(w, prec, device), sched, n)
        assert _rel(out, ref) < TOL[prec], (target, cin, cout)


def test_argmax_first_index_on_ties_and_label_gather(device):
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(1000, 224, generator=g)
    logits[5, 7] = logits[5, 150] = 9.0
    logits[6, 199] = 50.0
    logits[:, 200:] = 100.0  # padding columns must be ignored
    lab = ops.argmax_rows(logits.to(device), 200)
    assert torch.equal(lab.cpu().long(), logits[:, :200].max(1)[1])
    assert lab[5].item() == 7
    v2p = torch.randint(0, 1000, (5000,), generator=g)
    pl = ops.gather_labels(lab, v2p.to(device))
    assert torch.equal(pl.cpu(), logits[:, :200].max(1)[1][v2p])
