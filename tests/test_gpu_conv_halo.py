"""Halo-tiled convolution (m3d_build_halo + the halo path of m3d_conv_fwd_sf32) against the CPU oracle and
against the structure of the kernel map.  Tolerance: bf16x3 products, 6e-5 per-row relative L2, one layer."""
import numpy as np
import pytest
import torch

import oracle
from segmentation_amd import ops, scenes

pytestmark = pytest.mark.gpu
TOL = 6e-5


def _rel(out, ref):
    ref = ref.double()
    return ((out.double().cpu() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-12)).max().item()


def _level(device, seed, target, mode):
    sc = scenes.make_scene(seed=seed, target_voxels=target)
    off = np.array([0, sc["coord"].shape[0]], dtype=np.int64)
    sp = oracle.sparsify(sc["coord"], sc["feat"], off, 0.02)
    idx = torch.from_numpy(sp["voxel_coords"]).to(device)
    pair = ops.subm_rulebook(idx, sp["spatial_shape"])
    sched = ops.build_schedule(pair, idx.shape[0], idx, mode, sp["spatial_shape"], 1)
    ops.build_halo(sched, idx.shape[0])
    return dict(idx=idx, pair=pair.cpu().numpy(), sched=sched, n=idx.shape[0])


@pytest.fixture(scope="module")
def morton_level(device):
    return _level(device, 4, 12000, 1)


@pytest.fixture(scope="module")
def scattered_level(device):
    # pattern-grouped schedule with a single cell: tiles reference rows all over the scene, halos overflow
    return _level(device, 4, 12000, 3 | (15 << 8))


def test_halo_structure(morton_level, scattered_level):
    for lv, expect_overflow in ((morton_level, False), (scattered_level, True)):
        sched = lv["sched"]
        cnt, rows, loc, order = (t.cpu().numpy() for t in sched.halo)
        nov = int((cnt < 0).sum())
        assert np.array_equal(np.sort(order), np.arange(len(cnt))) and (cnt[order[:nov]] < 0).all()
        assert (np.diff(order[:nov]) > 0).all() and (np.diff(order[nov:]) > 0).all()
        loc = loc.view(np.uint16)
        nbr = sched.nbr.cpu().numpy()
        orow = sched.out_row.cpu().numpy()
        pos = np.zeros(lv["n"], dtype=np.int64)
        pos[orow[orow >= 0]] = np.nonzero(orow >= 0)[0]
        frac = (cnt < 0).mean()
        assert frac > 0.5 if expect_overflow else frac < 0.2
        for t in range(sched.npad // 256):
            blk = nbr[:, t * 256:(t + 1) * 256]
            ids = np.unique(blk[blk >= 0])
            if cnt[t] < 0:
                assert len(ids) > 640
                continue
            assert cnt[t] == len(ids)
            mine = rows[t, :cnt[t]]
            assert np.array_equal(np.sort(mine), ids)
            assert (np.diff(pos[mine]) > 0).all()  # listed in schedule order
            l = loc[t].astype(np.int64)
            assert ((l == 0xFFFF) == (blk < 0)).all()
            assert np.array_equal(mine[l[blk >= 0]], blk[blk >= 0])


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64), (96, 96), (128, 96), (128, 128), (256, 256)])
def test_subm_conv_halo(device, morton_level, cin, cout):
    lv = morton_level
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn(lv["n"], cin, generator=g)
    w = torch.randn(cout, 3, 3, 3, cin, generator=g) / np.sqrt(9 * cin)
    ref = oracle.subm_conv(x, w, lv["pair"])
    prof = ops.PROFILE = []
    try:
        out = ops.conv_fwd(ops.split_f32(x.to(device), 2), ops.pack_weight(w, "bf16x3", device), lv["sched"], lv["n"])
    finally:
        ops.PROFILE = None
    assert prof[0][0] == "halo"
    assert _rel(out, ref) < TOL


def test_halo_overflow_tiles_take_the_per_offset_path(device, scattered_level):
    lv = scattered_level
    g = torch.Generator().manual_seed(3)
    x = torch.randn(lv["n"], 96, generator=g)
    w = torch.randn(96, 3, 3, 3, 96, generator=g) / np.sqrt(9 * 96)
    ref = oracle.subm_conv(x, w, lv["pair"])
    out = ops.conv_fwd(ops.split_f32(x.to(device), 2), ops.pack_weight(w, "bf16x3", device), lv["sched"], lv["n"])
    assert _rel(out, ref) < TOL


def test_halo_fused_epilogue_two_sources_and_split_output(device, morton_level):
    lv = morton_level
    n = lv["n"]
    g = torch.Generator().manual_seed(7)
    xa, xb = torch.randn(n, 96, generator=g), torch.randn(n, 32, generator=g)
    w = torch.randn(96, 3, 3, 3, 128, generator=g) / np.sqrt(9 * 128)
    scale, shift = torch.rand(96, generator=g) + 0.5, torch.randn(96, generator=g)
    res = torch.randn(n, 96, generator=g)
    ref = torch.relu(oracle.subm_conv(torch.cat((xa, xb), 1), w, lv["pair"]) * scale + shift + res)
    out, osp = ops.conv_fwd(ops.split_f32(xa.to(device), 2), ops.pack_weight(w, "bf16x3", device), lv["sched"], n,
                            in1=ops.split_f32(xb.to(device), 2), scale=scale.to(device), shift=shift.to(device),
                            residual=res.to(device), relu=True, split_terms=2)
    assert _rel(out, ref) < TOL
    back = ops.unsplit(osp)
    assert ((back - out).abs() <= 2.0 ** -16 * out.abs()).all()
    raw = osp.data.view(torch.bfloat16).view(n + 1, 3, 2, 32)
    assert (raw[n] == 0).all()
