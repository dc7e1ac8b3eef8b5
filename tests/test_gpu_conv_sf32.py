"""The LDS-DMA convolution on split-format (sf32) activations (m3d_conv_fwd_sf32, m3d_split_f32) against the
CPU oracle.  Tolerances, per-row relative L2 over one layer: terms=3 (bf16x6) 1e-5, terms=2 (bf16x3) 6e-5,
terms=1 (plain bf16 operands, f32 accumulation) 1e-2, format 4 (f16x3: operands as f16 pairs exact to 2^-22) 1e-5."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from segmentation_amd import ops, scenes

pytestmark = pytest.mark.gpu
TOL = {3: 1e-5, 2: 6e-5, 1: 1e-2, 4: 1e-5}
PREC = {3: "bf16x6", 2: "bf16x3", 1: "bf16", 4: "f16x3"}
SPLIT_TOL = {3: 2.0 ** -23, 2: 2.0 ** -16, 1: 2.0 ** -8, 4: 2.0 ** -21}
FORMATS = [1, 2, 3, 4]


def _zero_row_is_zero(st):
    row_bytes = st.c * (2 if st.terms == 4 else st.terms) * 2
    return bool((st.data[st.n * row_bytes:(st.n + 1) * row_bytes] == 0).all()) and st.data.numel() == (st.n + 1) * row_bytes


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


@pytest.mark.parametrize("terms", FORMATS)
def test_split_roundtrip(device, terms):
    g = torch.Generator().manual_seed(0)
    # the f16 pair covers the f16 range only, and resolves 2^-36 at best
    x = torch.randn(1000, 96, generator=g) * (torch.logspace(-6, 6, 96) if terms != 4 else torch.logspace(-5, 3.5, 96))
    st = ops.split_f32(x.to(device), terms)
    back = ops.unsplit(st).cpu()
    assert ((back - x).abs() <= SPLIT_TOL[terms] * x.abs() + (2.0 ** -35 if terms == 4 else 0.0)).all()
    assert _zero_row_is_zero(st)
    if terms == 1:  # one term is plain row-major bf16
        assert torch.equal(st.data.view(torch.bfloat16).view(1001, 96)[:1000].cpu(), x.to(torch.bfloat16))


@pytest.mark.parametrize("terms", FORMATS)
@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64), (96, 96), (128, 96), (128, 128), (256, 256), (384, 256)])
def test_subm_conv(device, level, cin, cout, terms):
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    n = level["idx"].shape[0]
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, 3, 3, 3, cin, generator=g) / np.sqrt(9 * cin)
    ref = oracle.subm_conv(x, w, level["pair"])
    out = ops.conv_fwd(ops.split_f32(x.to(device), terms), ops.pack_weight(w, PREC[terms], device), level["sched"], n)
    assert _rel(out, ref) < TOL[terms]


@pytest.mark.parametrize("terms", FORMATS)
def test_fused_epilogue_two_sources_and_split_output(device, level, terms):
    g = torch.Generator().manual_seed(7)
    n = level["idx"].shape[0]
    xa, xb = torch.randn(n, 96, generator=g), torch.randn(n, 32, generator=g)
    w = torch.randn(96, 3, 3, 3, 128, generator=g) / np.sqrt(9 * 128)
    scale, shift = torch.rand(96, generator=g) + 0.5, torch.randn(96, generator=g)
    res = torch.randn(n, 96, generator=g)
    ref = torch.relu(oracle.subm_conv(torch.cat((xa, xb), 1), w, level["pair"]) * scale + shift + res)
    out, osp = ops.conv_fwd(ops.split_f32(xa.to(device), terms), ops.pack_weight(w, PREC[terms], device), level["sched"],
                            n, in1=ops.split_f32(xb.to(device), terms), scale=scale.to(device), shift=shift.to(device),
                            residual=res.to(device), relu=True, split_terms=terms)
    assert _rel(out, ref) < TOL[terms]
    # the split copy is the same tensor in sf32, zero row included
    back = ops.unsplit(osp)
    assert ((back - out).abs() <= SPLIT_TOL[terms] * out.abs() + (2.0 ** -35 if terms == 4 else 0.0)).all()
    assert _zero_row_is_zero(osp)
    # split-only output
    none, osp2 = ops.conv_fwd(ops.split_f32(xa.to(device), terms), ops.pack_weight(w, PREC[terms], device),
                              level["sched"], n, in1=ops.split_f32(xb.to(device), terms), scale=scale.to(device),
                              shift=shift.to(device), residual=res.to(device), relu=True, want_f32=False,
                              split_terms=terms)
    assert none is None and torch.equal(osp2.data, osp.data)


@pytest.mark.parametrize("terms", FORMATS)
@pytest.mark.parametrize("cin,cout", [(32, 64), (128, 256)])
def test_strided_and_inverse(device, level, cin, cout, terms):
    g = torch.Generator().manual_seed(cin + cout)
    n, m = level["idx"].shape[0], level["oi"].shape[0]
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, 2, 2, 2, cin, generator=g) / np.sqrt(8 * cin)
    ref = oracle.strided_conv(x, w, level["pf"].cpu().numpy())
    out = ops.conv_fwd(ops.split_f32(x.to(device), terms), ops.pack_weight(w, PREC[terms], device), level["sched_down"], m)
    assert _rel(out, ref) < TOL[terms]
    y = torch.randn(m, cout, generator=g)
    wi = torch.randn(cin, 2, 2, 2, cout, generator=g) / np.sqrt(cout)
    ref_i = oracle.inverse_conv(y, wi, level["pb"].cpu().numpy())
    out_i = ops.conv_fwd(ops.split_f32(y.to(device), terms), ops.pack_weight(wi, PREC[terms], device), level["sched_up"], n)
    assert _rel(out_i, ref_i) < TOL[terms]


@pytest.mark.parametrize("terms", FORMATS)
@pytest.mark.parametrize("n,cin,cout", [(5000, 96, 768), (700, 384, 256), (130, 128, 96), (1, 96, 32), (70000, 128, 96)])
def test_dense_1x1(device, n, cin, cout, terms):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, 1, 1, 1, cin, generator=g) / np.sqrt(cin)
    b = torch.randn(cout, generator=g)
    ref = oracle.conv1x1(x, w, b)
    out = ops.conv_fwd(ops.split_f32(x.to(device), terms), ops.pack_weight(w, PREC[terms], device), None, n, shift=b.to(device))
    assert _rel(out, ref) < TOL[terms]


@pytest.mark.parametrize("terms", FORMATS)
@pytest.mark.parametrize("target", [300, 1500, 30000])
def test_all_tile_configurations(device, target, terms):
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
        out = ops.conv_fwd(ops.split_f32(x.to(device), terms), ops.pack_weight(w, PREC[terms], device), sched, n)
        assert _rel(out, ref) < TOL[terms], (target, cin, cout)


@pytest.mark.parametrize("terms", [2, 3, 4])
@pytest.mark.parametrize("target", [300, 1500, 12000])
def test_f32_rows_with_split_products_equal_the_presplit_path(device, target, terms):
    """m3d_conv_fwd_f32 at precision bf16x3 / bf16x6 gathers plain f32 rows into LDS and forms the terms after the
    fragment read; m3d_conv_fwd_sf32 reads terms split beforehand.  Same terms, same products, same order: the
    results are equal bit for bit (submanifold, two sources + epilogue + split output, strided, inverse, dense)."""
    prec = PREC[terms]
    sc = scenes.make_scene(seed=21, target_voxels=target)
    off = np.array([0, sc["coord"].shape[0]], dtype=np.int64)
    sp = oracle.sparsify(sc["coord"], sc["feat"], off, 0.02)
    idx = torch.from_numpy(sp["voxel_coords"]).to(device)
    n = idx.shape[0]
    pair = ops.subm_rulebook(idx, sp["spatial_shape"])
    sched = ops.build_schedule(pair, n, idx, 1, sp["spatial_shape"], 1)
    oi, oshape, pf, pb = ops.strided_rulebook(idx, sp["spatial_shape"], 1)
    m = oi.shape[0]
    down, up = ops.build_schedule(pf, m, None, 0), ops.build_schedule(pb, n, None, 2, base_order=sched.out_row)
    g = torch.Generator().manual_seed(target)
    for cin, cout in [(32, 32), (96, 96), (64, 128), (256, 256)]:
        x = torch.randn(n, cin, generator=g).to(device)
        wref = torch.randn(cout, 3, 3, 3, cin, generator=g) / np.sqrt(9 * cin)
        w = ops.pack_weight(wref, prec, device)
        a = ops.conv_fwd(x, w, sched, n)
        b = ops.conv_fwd(ops.split_f32(x, terms), w, sched, n)
        assert torch.equal(a, b), (cin, cout)
        assert _rel(a, oracle.subm_conv(x.cpu(), wref, pair.cpu().numpy())) < TOL[terms]
    # two sources, affine + residual + ReLU, f32 and split outputs
    xa, xb = torch.randn(n, 96, generator=g).to(device), torch.randn(n, 32, generator=g).to(device)
    w = ops.pack_weight(torch.randn(96, 3, 3, 3, 128, generator=g) / np.sqrt(9 * 128), prec, device)
    scale, shift = (torch.rand(96, generator=g) + 0.5).to(device), torch.randn(96, generator=g).to(device)
    res = torch.randn(n, 96, generator=g).to(device)
    a, asp = ops.conv_fwd(xa, w, sched, n, in1=xb, scale=scale, shift=shift, residual=res, relu=True, split_terms=terms)
    b, bsp = ops.conv_fwd(ops.split_f32(xa, terms), w, sched, n, in1=ops.split_f32(xb, terms), scale=scale, shift=shift,
                          residual=res, relu=True, split_terms=terms)
    assert torch.equal(a, b) and torch.equal(asp.data, bsp.data)
    # strided, inverse, dense
    x = torch.randn(n, 64, generator=g).to(device)
    w = ops.pack_weight(torch.randn(128, 2, 2, 2, 64, generator=g) / np.sqrt(8 * 64), prec, device)
    assert torch.equal(ops.conv_fwd(x, w, down, m), ops.conv_fwd(ops.split_f32(x, terms), w, down, m))
    y = torch.randn(m, 128, generator=g).to(device)
    wi = ops.pack_weight(torch.randn(64, 2, 2, 2, 128, generator=g) / np.sqrt(128), prec, device)
    assert torch.equal(ops.conv_fwd(y, wi, up, n), ops.conv_fwd(ops.split_f32(y, terms), wi, up, n))
    w1 = ops.pack_weight(torch.randn(768, 1, 1, 1, 96, generator=g) / np.sqrt(96), prec, device)
    x96 = torch.randn(n, 96, generator=g).to(device)
    assert torch.equal(ops.conv_fwd(x96, w1, None, n), ops.conv_fwd(ops.split_f32(x96, terms), w1, None, n))



def test_three_workgroups_per_cu_variant(device, tmp_path):
    """Launches of 1024 and more 128-row x 96-column tiles at three terms keep the kernel map in global memory (three
    workgroups per CU).  Same bits as the two-workgroup kernel (run in a child process with the variant switched off
    through M3D_GMAP_TILES), and both within the three-term tolerance of the oracle."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _conv_large_child as child
    r = child.run(device)
    assert r["n"] >= 1024 * 128
    ref1 = oracle.subm_conv(r["xa"], r["w1"], r["pair"])
    assert _rel(r["out1"], ref1) < TOL[3]
    ref2 = torch.relu(oracle.subm_conv(torch.cat((r["xa"], r["xb"]), 1), r["w2"], r["pair"]) * r["scale"] + r["shift"] + r["res"])
    assert _rel(r["out2"], ref2) < TOL[3]
    path = str(tmp_path / "off.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, M3D_GMAP_TILES="1000000000", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    subprocess.run([sys.executable, child.__file__, path], check=True, env=env, timeout=300)
    off = np.load(path)
    assert np.array_equal(off["out1"], r["out1"].numpy())
    assert np.array_equal(off["out2"], r["out2"].numpy())
    assert np.array_equal(off["osp"], r["osp"].numpy())


def test_f16_status_reports_values_beyond_the_f16_range(device):
    """The f16 pair cannot hold |x| > 65504: a kernel that meets such a value sets the status word."""
    ops.f16_status(clear=True)
    x = torch.randn(500, 64, device=device)
    ops.split_f32(x, 4)
    assert ops.f16_status(clear=True) is False
    x[17, 5] = 7.0e4
    ops.split_f32(x, 4)
    assert ops.f16_status(clear=True) is True
    assert ops.f16_status(clear=True) is False
    # a convolution whose output leaves the range reports it through its split output
    w = ops.pack_weight(torch.full((32, 1, 1, 1, 64), 100.0), "f16x3", device)
    big = torch.full((300, 64), 50.0, device=device)
    ops.conv_fwd(ops.split_f32(big, 4), w, None, 300, split_terms=4)
    assert ops.f16_status(clear=True) is True


def test_operands_beyond_the_buffer_addressing_are_refused(device, level):
    """The kernel addresses an operand with 32-bit byte offsets inside one buffer descriptor: a row count that would
    put any row of a split tensor at or beyond 4 GB is an argument error, raised before anything is launched (the
    tensor handed over here is small; only the declared row count is large)."""
    from segmentation_amd import _lib
    lv = level
    n = lv["idx"].shape[0]
    x = ops.split_f32(torch.randn(n, 96, device=device), 3)
    w = ops.pack_weight(torch.randn(96, 27, 96) * 0.05, "bf16x6", device)
    big = ops.SplitTensor(x.data, 8_000_000, 96, 3)   # 8 M rows x 96 channels x 6 bytes = 4.6 GB
    with pytest.raises(_lib.M3DError, match="4 GB"):
        ops.conv_fwd(big, w, lv["sched"], n)
    out = ops.conv_fwd(x, w, lv["sched"], n)          # the same call with the true row count still runs
    out = out[0] if isinstance(out, tuple) else out
    assert torch.isfinite(out).all()
