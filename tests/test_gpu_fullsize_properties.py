"""Benchmark-size scene (~273 k voxels, 725 k points) through size-independent properties, where the
CPU oracle would take minutes: sortedness and uniqueness of the voxel keys, consistency of v2p_map, symmetry of
the submanifold map, parent/child consistency of the strided maps, schedules being permutations, linearity of
the convolution, determinism, and label agreement between the split-bf16 and the exact-f32 engines."""
import numpy as np
import pytest
import torch

import bench
from segmentation_amd import ops, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(device):
    sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
    n = sc["coord"].shape[0]
    coord, feat = torch.from_numpy(sc["coord"]).to(device), torch.from_numpy(sc["feat"]).to(device)
    vox = ops.voxelize(coord, feat, np.array([0, n]), 0.02, return_keys=True, return_grid_coord=True)
    return dict(coord=coord, feat=feat, n=n, vox=vox)


def test_voxelize_properties(big):
    vox, n = big["vox"], big["n"]
    keys = vox["keys"].cpu().numpy().view(np.uint64)
    n0 = vox["n_voxels"]
    assert 250_000 < n0 < 300_000 and keys.shape[0] == n0
    assert (keys[1:] > keys[:-1]).all()                       # strictly ascending = sorted and unique
    v2p = vox["v2p_map"]
    assert int(v2p.min()) == 0 and int(v2p.max()) == n0 - 1
    # every point's grid coordinate equals its voxel's coordinate
    gc = vox["grid_coord"]
    vc = vox["voxel_coords"][:, 1:]
    assert torch.equal(vc[v2p], gc)
    # voxel count equals the number of distinct grid cells; mean pooling conserves the feature sum
    lin = (gc[:, 0].long() * 4096 + gc[:, 1].long()) * 4096 + gc[:, 2].long()
    assert torch.unique(lin).numel() == n0
    cnt = torch.bincount(v2p, minlength=n0).double()
    total = (vox["voxel_feats"].double() * cnt[:, None]).sum(0)
    assert torch.allclose(total, big["feat"].double().sum(0), rtol=1e-6, atol=1e-3)
    shape = torch.tensor(vox["spatial_shape"])
    assert torch.equal(shape, gc.max(0).values.cpu().long() + 128)   # structure.py:224


def test_kernel_map_properties(big):
    vox = big["vox"]
    idx, shape = vox["voxel_coords"], vox["spatial_shape"]
    n0 = idx.shape[0]
    pair = ops.subm_rulebook(idx, shape)
    ar = torch.arange(n0, device=idx.device, dtype=torch.int32)
    assert torch.equal(pair[13], ar)                          # centre offset is the identity
    for k in (0, 5, 12, 20, 26):                               # pair[26-k][pair[k][j]] == j
        p = pair[k].long()
        ok = p >= 0
        assert torch.equal(pair[26 - k][p[ok]], ar[ok])
        d = idx[p[ok]][:, 1:] - idx[ok][:, 1:]                 # and the neighbour sits at the offset
        off = torch.tensor([k // 9 - 1, (k // 3) % 3 - 1, k % 3 - 1], device=idx.device, dtype=torch.int32)
        assert (d == off).all()
    chain = ops.strided_chain(idx, shape, 1, 4)
    cur, n_in = idx, n0
    for oi, oshape, pf, pb in chain:
        m = oi.shape[0]
        assert (pb >= 0).sum(0).eq(1).all()                    # exactly one parent per input
        k = pb.max(0).indices
        parent = pb.max(0).values.long()
        assert torch.equal(oi[parent][:, 1:], cur[:, 1:] >> 1)
        assert torch.equal(pf[k, parent], torch.arange(n_in, device=idx.device, dtype=torch.int32))
        lin = (oi[:, 1].long() * oshape[1] + oi[:, 2].long()) * oshape[2] + oi[:, 3].long()
        assert (lin[1:] > lin[:-1]).all()                      # ascending linear index, no duplicates
        assert int((pf >= 0).sum()) == n_in
        cur, n_in = oi.contiguous(), m
    sched = ops.build_schedule(pair, n0, idx, 3 | (6 << 8), shape, 1)
    orow = sched.out_row[:n0].long()
    assert torch.equal(torch.sort(orow).values, torch.arange(n0, device=idx.device))
    assert torch.equal(sched.nbr[:, :n0], pair[:, orow]) and (sched.out_row[n0:] == -1).all()


def test_convolution_is_linear_and_deterministic(big):
    vox = big["vox"]
    idx, shape, n0 = vox["voxel_coords"], vox["spatial_shape"], vox["n_voxels"]
    pair = ops.subm_rulebook(idx, shape)
    sched = ops.build_schedule(pair, n0, idx, 3 | (6 << 8), shape, 1)
    g = torch.Generator(device=idx.device).manual_seed(0)
    x = torch.randn(n0, 96, device=idx.device, generator=g)
    y = torch.randn(n0, 96, device=idx.device, generator=g)
    w = ops.pack_weight(torch.randn(96, 27, 96, generator=torch.Generator().manual_seed(1)) * 0.03, "bf16x3", idx.device)
    fx = ops.conv_fwd(ops.split_f32(x, 2), w, sched, n0)
    fy = ops.conv_fwd(ops.split_f32(y, 2), w, sched, n0)
    fz = ops.conv_fwd(ops.split_f32(2.0 * x - 0.5 * y, 2), w, sched, n0)
    ref = 2.0 * fx - 0.5 * fy
    rel = ((fz - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-12)).max().item()
    assert rel < 1e-4                                          # bf16x3 products: 2^-16 per term
    assert torch.equal(fx, ops.conv_fwd(ops.split_f32(x, 2), w, sched, n0))


def test_labels_agree_between_precisions(big, device):
    preds = {}
    for prec in ("bf16x3", "f32"):
        eng, _, _ = bench.build_engine(device, prec)
        with torch.no_grad():
            a = eng.predict(big["coord"], big["feat"], np.array([0, big["n"]]), 0.02)["pred"]
            b = eng.predict(big["coord"], big["feat"], np.array([0, big["n"]]), 0.02)["pred"]
        assert torch.equal(a, b)                               # run-to-run deterministic
        preds[prec] = a
        del eng
    agree = (preds["bf16x3"] == preds["f32"]).float().mean().item()
    assert agree >= 0.999
