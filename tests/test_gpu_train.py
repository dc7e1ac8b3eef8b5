"""Training path on the GPU: the backward of every sparse convolution type against torch autograd on the CPU
oracle (float64), the caption loss, and one whole optimisation step of a small PPT/SpUNet against the same
network evaluated on the CPU with the oracle's convolutions.  Tolerances: single layers 1e-5 relative L2 (exact
f32 products), whole-network gradients 2e-3 per parameter (float32 batch-norm statistics through 30+ layers)."""
import copy

import numpy as np
import pytest
import torch

import oracle
from oracle import caption_loss as oracle_loss
from segmentation_amd import ops, scenes, sparse, spunet, train

pytestmark = pytest.mark.gpu


def _rel(a, b):
    b = b.double()
    return ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def level():
    sc = scenes.make_scene(seed=11, target_voxels=5000)
    off = np.array([0, sc["coord"].shape[0]], dtype=np.int64)
    sp = oracle.sparsify(sc["coord"], sc["feat"], off, 0.02)
    pair = oracle.subm_rulebook(sp["voxel_coords"], sp["spatial_shape"])
    oi, oshape, pf, pb = oracle.strided_rulebook(sp["voxel_coords"], sp["spatial_shape"])
    return dict(sp=sp, pair=pair, oi=oi, oshape=oshape, pf=pf, pb=pb)


def _tensor(device, level, feats):
    sp = level["sp"]
    return sparse.SparseConvTensor(feats, torch.from_numpy(sp["voxel_coords"]).to(device), sp["spatial_shape"], 1)


@pytest.mark.parametrize("cin,cout,bias", [(3, 32, False), (32, 64, True), (96, 96, False), (192, 128, False)])
def test_subm_conv_backward(device, level, cin, cout, bias):
    g = torch.Generator().manual_seed(cin + cout)
    n = level["sp"]["voxel_coords"].shape[0]
    x = torch.randn(n, cin, generator=g)
    conv = sparse.SubMConv3d(cin, cout, 3, bias=bias, indice_key="k").to(device)
    xg = x.to(device).requires_grad_(True)
    out = conv(_tensor(device, level, xg)).features
    gout = torch.randn(n, cout, generator=g)
    out.backward(gout.to(device))
    xr = x.double().requires_grad_(True)
    wr = conv.weight.detach().cpu().double().requires_grad_(True)
    br = None if not bias else conv.bias.detach().cpu().double().requires_grad_(True)
    ref = oracle.subm_conv(xr, wr, level["pair"], br)
    ref.backward(gout.double())
    assert _rel(out.detach(), ref.detach()) < 1e-5
    assert _rel(xg.grad, xr.grad) < 1e-5
    assert _rel(conv.weight.grad, wr.grad) < 1e-5
    if bias:
        assert _rel(conv.bias.grad, br.grad) < 1e-5


def test_strided_and_inverse_conv_backward(device, level):
    g = torch.Generator().manual_seed(3)
    n, m = level["sp"]["voxel_coords"].shape[0], level["oi"].shape[0]
    x = torch.randn(n, 32, generator=g)
    down = sparse.SparseConv3d(32, 64, 2, stride=2, bias=False, indice_key="s").to(device)
    up = sparse.SparseInverseConv3d(64, 32, 2, indice_key="s", bias=False).to(device)
    xg = x.to(device).requires_grad_(True)
    mid = down(_tensor(device, level, xg))
    assert mid.features.shape[0] == m
    out = up(mid).features
    gout = torch.randn(n, 32, generator=g)
    out.backward(gout.to(device))
    xr = x.double().requires_grad_(True)
    wd = down.weight.detach().cpu().double().requires_grad_(True)
    wu = up.weight.detach().cpu().double().requires_grad_(True)
    ref = oracle.inverse_conv(oracle.strided_conv(xr, wd, level["pf"]), wu, level["pb"])
    ref.backward(gout.double())
    assert _rel(out.detach(), ref.detach()) < 1e-5
    assert _rel(xg.grad, xr.grad) < 1e-5
    assert _rel(down.weight.grad, wd.grad) < 1e-5
    assert _rel(up.weight.grad, wu.grad) < 1e-5


@pytest.mark.parametrize("cin,cout", [(3, 32), (32, 64), (96, 96), (192, 128)])
def test_backward_with_split_bf16_products(device, level, cin, cout):
    """set_precision("bf16x3"): forward, input gradient and weight gradient (m3d_conv_bwd_weight_bf16x3) form every
    product from two-term bf16 splits; bar 5e-5 relative L2 (measured 3e-6..1e-5), against 1e-5 for exact f32."""
    g = torch.Generator().manual_seed(cin * 7 + cout)
    n, m = level["sp"]["voxel_coords"].shape[0], level["oi"].shape[0]
    x = torch.randn(n, cin, generator=g)
    sparse.set_precision("bf16x3")
    try:
        conv = sparse.SubMConv3d(cin, cout, 3, bias=False, indice_key="k").to(device)
        down = sparse.SparseConv3d(cout, cout, 2, stride=2, bias=False, indice_key="s").to(device)
        up = sparse.SparseInverseConv3d(cout, cout, 2, indice_key="s", bias=False).to(device)
        one = sparse.SubMConv3d(cout, 32, 1, bias=False).to(device)
        xg = x.to(device).requires_grad_(True)
        out = one(up(down(conv(_tensor(device, level, xg))))).features
        gout = torch.randn(n, 32, generator=g)
        out.backward(gout.to(device))
    finally:
        sparse.set_precision("f32")
    xr = x.double().requires_grad_(True)
    ws = [mod.weight.detach().cpu().double().requires_grad_(True) for mod in (conv, down, up, one)]
    ref = oracle.subm_conv(xr, ws[0], level["pair"], None)
    ref = oracle.inverse_conv(oracle.strided_conv(ref, ws[1], level["pf"]), ws[2], level["pb"])
    ref = oracle.conv1x1(ref, ws[3], None)
    ref.backward(gout.double())
    errs = [_rel(out.detach(), ref.detach()), _rel(xg.grad, xr.grad)]
    errs += [_rel(mod.weight.grad, w.grad) for mod, w in zip((conv, down, up, one), ws)]
    assert max(errs) < 5e-5, errs


def test_f16_pairs_are_refused_for_training(device):
    conv = sparse.SubMConv3d(32, 32, 1, bias=False).to(device)
    x = sparse.SparseConvTensor(torch.randn(100, 32, device=device, requires_grad=True),
                                torch.zeros((100, 4), dtype=torch.int32, device=device), [8, 8, 8], 1)
    sparse.set_precision("f16x3")
    try:
        with pytest.raises(ValueError):
            conv(x)
        with torch.no_grad():  # inference through the modules is fine
            y = conv(x).features
        ref = x.features.detach() @ conv.weight.detach().reshape(32, 32).T
        assert ((y - ref).norm() / ref.norm()).item() < 1e-6
    finally:
        sparse.set_precision("f32")


def test_conv1x1_backward(device):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3000, 96, generator=g)
    conv = sparse.SubMConv3d(96, 768, 1, bias=True).to(device)
    idx = torch.zeros((3000, 4), dtype=torch.int32, device=device)
    xg = x.to(device).requires_grad_(True)
    out = conv(sparse.SparseConvTensor(xg, idx, [8, 8, 8], 1)).features
    gout = torch.randn(3000, 768, generator=g)
    out.backward(gout.to(device))
    xr = x.double().requires_grad_(True)
    wr = conv.weight.detach().cpu().double().requires_grad_(True)
    br = conv.bias.detach().cpu().double().requires_grad_(True)
    ref = oracle.conv1x1(xr, wr, br)
    ref.backward(gout.double())
    assert _rel(xg.grad, xr.grad) < 1e-5 and _rel(conv.weight.grad, wr.grad) < 1e-5 and _rel(conv.bias.grad, br.grad) < 1e-5


def _caption_batch(rng, n_pts, n_caps, d, device):
    uniq = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(n_caps // 2 + 1, d)).astype(np.float32)), dim=1)
    emb = [[uniq[int(rng.integers(0, uniq.shape[0]))] for _ in range(n_caps)]]
    idx = [[torch.from_numpy(rng.choice(n_pts, size=int(rng.integers(20, 200)), replace=False)) for _ in range(n_caps)]]
    pi, co, npc = train.flatten_caption_indices(idx, torch.tensor([0, n_pts]))
    return dict(embedding=[[e.to(device) for e in emb[0]]], point_indices=pi.to(device), caption_offsets=co,
                num_points_per_caption=npc)


def test_caption_loss_matches_the_oracle(device):
    rng = np.random.default_rng(0)
    n, d = 4000, 64
    feats = torch.from_numpy(rng.normal(size=(n, d)).astype(np.float32))
    cd = _caption_batch(rng, n, 12, d, device)
    text, seg, _ = train.unique_text_features(cd["embedding"])
    fg = feats.to(device).requires_grad_(True)
    loss = train.caption_loss(fg, text, seg, cd["point_indices"], cd["caption_offsets"], cd["num_points_per_caption"])
    loss.backward()
    # the CPU oracle in float64 (pinned to the reference's own CaptionLoss.loss by tests/golden/reference_loss.npz)
    fr = feats.double().requires_grad_(True)
    emb64 = [[e.cpu().double() for e in cd["embedding"][0]]]
    ref = oracle_loss.caption_loss(fr, emb64, cd["point_indices"].cpu(), cd["caption_offsets"],
                                   cd["num_points_per_caption"])
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    assert _rel(fg.grad, fr.grad) < 1e-4


def test_caption_loss_row_map_equals_gathered_rows(device):
    """clip_feat = features[v2p_map] applied inside the loss (once per voxel) equals the per-point form."""
    rng = np.random.default_rng(1)
    n_vox, n_pts, d = 1500, 5000, 64
    vf = torch.from_numpy(rng.normal(size=(n_vox, d)).astype(np.float32)).to(device)
    v2p = torch.from_numpy(rng.integers(0, n_vox, size=n_pts)).to(device)
    cd = _caption_batch(rng, n_pts, 10, d, device)
    text, seg, first = train.unique_text_features(cd["embedding"])
    stacked = torch.stack(cd["embedding"][0])
    assert torch.equal(text[seg], stacked) and torch.equal(stacked[first], text)  # unique rows, first occurrences
    a = vf.clone().requires_grad_(True)
    la = train.caption_loss(a[v2p], text, seg, cd["point_indices"], cd["caption_offsets"], cd["num_points_per_caption"])
    la.backward()
    b = vf.clone().requires_grad_(True)
    lb = train.caption_loss(b, text, seg, cd["point_indices"], cd["caption_offsets"], cd["num_points_per_caption"],
                            row_map=v2p)
    lb.backward()
    assert abs(la.item() - lb.item()) <= 1e-6 * abs(la.item())
    assert _rel(b.grad, a.grad.cpu()) < 1e-5


from _refmodel import CpuConvs as _CpuConvs  # noqa: E402  (sparse convolutions through the CPU oracle)


@pytest.mark.parametrize("fold_head", [True, False])
def test_whole_training_step_matches_cpu_oracle(device, monkeypatch, fold_head):
    torch.manual_seed(0)
    cfg = dict(spunet.SPUNET18A, out_channels=64, channels=(32, 64, 64, 96, 96, 64, 32, 32), layers=(1, 1, 1, 1, 1, 1, 1, 1))
    net = spunet.PPT(spunet.SpUNet(in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                                   base_channels=cfg["base_channels"], context_channels=cfg["context_channels"],
                                   channels=cfg["channels"], layers=cfg["layers"], zero_init=False))
    ref = copy.deepcopy(net).double()
    net = net.to(device)
    sc = scenes.make_scene(seed=5, target_voxels=4000)
    n = sc["coord"].shape[0]
    rng = np.random.default_rng(1)
    cd = _caption_batch(rng, n, 10, 64, device)
    batch = dict(coord=torch.from_numpy(sc["coord"]).to(device), feat=torch.from_numpy(sc["feat"]).to(device),
                 offset=torch.tensor([0, n]), grid_size=0.02, condition=["ScanNet"], caption_data=cd)
    step = train.TrainStep(net, total_steps=10, fold_head=fold_head)
    assert (step._fold_backbone is not None) == fold_head
    net.train()
    loss = step.loss(batch)
    loss.backward()
    # the same on the CPU through the oracle
    cdc = dict(embedding=[[e.cpu().double() for e in cd["embedding"][0]]], point_indices=cd["point_indices"].cpu(),
               caption_offsets=cd["caption_offsets"], num_points_per_caption=cd["num_points_per_caption"])
    bc = dict(coord=torch.from_numpy(sc["coord"]), feat=torch.from_numpy(sc["feat"]), offset=torch.tensor([0, n]),
              grid_size=0.02, condition=["ScanNet"], caption_data=cdc)
    ref.train()
    # the CPU copy forms every norm's modulation on its own; the GPU model above folds all of them in one product
    monkeypatch.setenv("M3D_FOLD_MODULATIONS", "0")
    with _CpuConvs():
        out = ref(bc)
        # the loss by the CPU oracle (oracle/caption_loss.py, pinned to the reference's CaptionLoss.loss by
        # tests/golden/reference_loss.npz), not by the product's own function
        loss_ref = oracle_loss.training_loss(out["sparse_conv_feat"].features, out["v2p_map"], cdc["embedding"],
                                             cdc["point_indices"], cdc["caption_offsets"],
                                             cdc["num_points_per_caption"])
        loss_ref.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * abs(loss_ref.item())
    checked = 0
    for (name, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert p.grad is None or p.grad.abs().max() == 0, name
            continue
        assert p.grad is not None, name
        if q.grad.norm() > 1e-12:
            assert _rel(p.grad, q.grad) < 2e-3, (name, _rel(p.grad, q.grad))
            checked += 1
    assert checked > 40
    monkeypatch.delenv("M3D_FOLD_MODULATIONS")
    # and the optimiser step runs and changes the weights
    l1 = step(batch)
    l2 = step(batch)
    assert np.isfinite(l1) and np.isfinite(l2)


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16x6"])
@pytest.mark.parametrize("cout,k,cin,flip", [(64, 27, 32, True), (96, 8, 128, False), (32, 1, 96, False), (32, 27, 3, True)])
def test_transposed_weight_packing_equals_packing_the_transposed_weight(device, precision, cout, k, cin, flip):
    """m3d_conv_pack_weight_transposed forms the input-gradient weight W'[ci][k'][co] = W[co][k-1-k' | k'][ci] in
    one pass; its bytes must be those of m3d_conv_pack_weight on an explicitly transposed (and flipped) copy."""
    w = torch.randn(cout, k, cin, generator=torch.Generator().manual_seed(cout + k + cin)).to(device)
    wt = w.permute(2, 1, 0)
    if flip:
        wt = wt.flip(1)
    a = ops.pack_weight(wt.contiguous(), precision)
    b = ops.pack_weight(w, precision, transposed=True, flip=flip)
    assert (a.k, a.cout, a.cin) == (b.k, b.cout, b.cin) == (k, cin, cout)
    assert torch.equal(a.data, b.data)
