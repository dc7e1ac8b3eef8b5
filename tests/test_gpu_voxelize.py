"""GPU voxelizer (m3d_voxelize through the C ABI) against the CPU oracle and the reference's golden
vectors: keys, voxel order, coordinates and v2p_map bit-exact; means within 1e-6."""
import numpy as np
import pytest
import torch

import oracle
from segmentation_amd import ops, scenes

pytestmark = pytest.mark.gpu


def _run(coord, feat, offset, grid, device, **kw):
    out = ops.voxelize(torch.from_numpy(coord).to(device), torch.from_numpy(feat).to(device), offset, grid,
                       return_keys=True, return_grid_coord=True, **kw)
    torch.cuda.synchronize()
    return out


def _check(out, sp, hash_method):
    assert out["spatial_shape"] == sp["spatial_shape"]
    assert out["batch_size"] == sp["batch_size"]
    assert np.array_equal(out["grid_coord"].cpu().numpy(), sp["grid_coord"])
    assert np.array_equal(out["voxel_coords"].cpu().numpy(), sp["voxel_coords"])
    assert np.array_equal(out["v2p_map"].cpu().numpy(), sp["v2p_map"])
    keys = out["keys"].cpu().numpy().view(np.uint64)
    assert np.array_equal(keys, sp["key"].astype(np.uint64))
    np.testing.assert_allclose(out["voxel_feats"].cpu().numpy(), sp["voxel_feats"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", ["s1", "s2", "s3"])
@pytest.mark.parametrize("hm", ["fnv", "ravel"])
def test_golden_reference_vectors(golden_voxelize, device, name, hm):
    g = golden_voxelize
    coord, feat, off = g[f"vox_{name}_coord"], g[f"vox_{name}_feat"], g[f"vox_{name}_offset"]
    grid = float(g[f"vox_{name}_grid"][0])
    out = _run(coord, feat, off, grid, device, hash_method=hm)
    pre = f"vox_{name}_{hm}_"
    assert np.array_equal(out["voxel_coords"].cpu().numpy(), g[pre + "voxel_coords"])
    assert np.array_equal(out["v2p_map"].cpu().numpy(), g[pre + "v2p_map"])
    assert out["spatial_shape"] == g[pre + "spatial_shape"].tolist()
    assert out["batch_size"] == int(g[pre + "batch_size"][0])
    _check(out, oracle.sparsify(coord, feat, off, grid, hash_method=hm), hm)


@pytest.mark.parametrize("seed,target", [(0, 3000), (1, 60000)])
def test_scene_vs_oracle(device, seed, target):
    sc = scenes.make_scene(seed=seed, target_voxels=target)
    off = np.array([0, sc["coord"].shape[0]], dtype=np.int64)
    out = _run(sc["coord"], sc["feat"], off, 0.02, device)
    _check(out, oracle.sparsify(sc["coord"], sc["feat"], off, 0.02), "fnv")


@pytest.mark.parametrize("n", [2, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 5000])
def test_point_counts_around_the_one_workgroup_sort_limit(device, n):
    """Up to 4096 keys the radix sort runs in one workgroup (sixteen waves, equal contiguous shares), above that in
    three kernels per pass; the voxel order (ascending 64-bit FNV key, all eight passes) must not depend on it."""
    rng = np.random.default_rng(n)
    coord = (rng.random((n, 3)) * 40.0 - 20.0).astype(np.float32)  # nearly every point its own voxel
    feat = rng.standard_normal((n, 3)).astype(np.float32)
    off = np.array([0, n // 3, n], dtype=np.int64)
    out = _run(coord, feat, off, 0.05, device)
    _check(out, oracle.sparsify(coord, feat, off, 0.05), "fnv")


def test_batch_with_empty_sample_and_offset_without_leading_zero(device):
    rng = np.random.default_rng(3)
    coord = (rng.random((900, 3)) * 1.5 - 0.7).astype(np.float32)
    feat = rng.standard_normal((900, 6)).astype(np.float32)
    off = np.array([400, 400, 900], dtype=np.int64)  # second sample empty, no leading 0 (misc.py:8-13)
    out = _run(coord, feat, off, 0.05, device)
    _check(out, oracle.sparsify(coord, feat, off, 0.05), "fnv")


def test_single_point_and_all_points_in_one_voxel(device):
    coord = np.array([[0.3, -0.2, 0.1]], dtype=np.float32)
    feat = np.array([[1.0, 2.0, 3.0]], dtype=np.float32)
    out = _run(coord, feat, np.array([0, 1]), 0.02, device)
    _check(out, oracle.sparsify(coord, feat, np.array([0, 1]), 0.02), "fnv")
    coord = np.full((257, 3), 0.011, dtype=np.float32)
    feat = np.arange(257 * 3, dtype=np.float32).reshape(257, 3)
    out = _run(coord, feat, np.array([0, 257]), 0.02, device)
    assert out["n_voxels"] == 1
    _check(out, oracle.sparsify(coord, feat, np.array([0, 257]), 0.02), "fnv")


def test_negative_coordinates_truncate_toward_zero(device):
    # voxels straddling 0 are double width under the trunc rule (structure.py:219)
    coord = np.array([[-0.019, 0.0, 0.0], [0.019, 0.0, 0.0], [-0.021, 0.0, 0.0], [0.021, 0.0, 0.0]], np.float32)
    feat = np.eye(4, 3, dtype=np.float32)
    out = _run(coord, feat, np.array([0, 4]), 0.02, device)
    sp = oracle.sparsify(coord, feat, np.array([0, 4]), 0.02)
    assert sp["voxel_coords"].shape[0] == 3
    _check(out, sp, "fnv")


def test_recip_grid_mode(device):
    sc = scenes.make_scene(seed=2, target_voxels=2000)
    off = np.array([0, sc["coord"].shape[0]], dtype=np.int64)
    out = _run(sc["coord"], sc["feat"], off, 0.02, device, grid_mode="recip")
    _check(out, oracle.sparsify(sc["coord"], sc["feat"], off, 0.02, grid_mode="recip"), "fnv")


def test_full_size_properties(device):
    """At the benchmark's size the oracle is not run; check size-independent properties instead."""
    sc = scenes.make_scene(seed=0, target_voxels=250_000)
    n = sc["coord"].shape[0]
    out = _run(sc["coord"], sc["feat"], np.array([0, n]), 0.02, device)
    keys = out["keys"].cpu().numpy().view(np.uint64)
    assert (np.diff(keys.astype(np.float64)) >= 0).all() and len(np.unique(keys)) == len(keys)  # sorted, unique
    v2p = out["v2p_map"].cpu().numpy()
    n0 = out["n_voxels"]
    assert v2p.min() == 0 and v2p.max() == n0 - 1 and len(np.unique(v2p)) == n0
    # every point's grid coordinate equals its voxel's coordinate; keys are the FNV of the coordinates
    vc = out["voxel_coords"].cpu().numpy()
    gc = out["grid_coord"].cpu().numpy()
    assert np.array_equal(vc[v2p][:, 1:], gc)
    assert np.array_equal(oracle.fnv_hash_vec(vc), keys)
    # means: sum over voxels of count * mean == sum over points
    cnt = np.bincount(v2p, minlength=n0)
    vf = out["voxel_feats"].cpu().numpy().astype(np.float64)
    np.testing.assert_allclose((vf * cnt[:, None]).sum(0), sc["feat"].astype(np.float64).sum(0), rtol=1e-5, atol=1e-2)


def test_random_pooling(device):
    """pooling_method="random" (structure.py:69-110): voxel set, order and v2p_map bit-exact (they do not depend on
    the draw); every voxel's feature is the feature of one of its own points; a seed reproduces the draw, another
    seed changes it, and over many seeds every point of a voxel gets drawn."""
    sc = scenes.make_batch([3, 4], target_voxels=3000)
    sp = oracle.sparsify(sc["coord"], sc["feat"], sc["offset"], 0.02)
    c, f = torch.from_numpy(sc["coord"]).to(device), torch.from_numpy(sc["feat"]).to(device)
    outs = [ops.voxelize(c, f, sc["offset"], 0.02, pooling_method="random", seed=s) for s in (7, 7, 8)]
    for o in outs:
        assert np.array_equal(o["voxel_coords"].cpu().numpy(), sp["voxel_coords"])
        assert np.array_equal(o["v2p_map"].cpu().numpy(), sp["v2p_map"])
        ch = o["choice"].cpu().numpy()
        assert np.array_equal(sp["v2p_map"][ch], np.arange(sp["voxel_coords"].shape[0]))   # chosen point lies in its voxel
        assert np.array_equal(o["voxel_feats"].cpu().numpy(), sc["feat"][ch])              # and its feature is copied as is
    assert torch.equal(outs[0]["choice"], outs[1]["choice"]) and not torch.equal(outs[0]["choice"], outs[2]["choice"])
    # one voxel with 5 points: 200 seeds draw each of them, roughly uniformly
    coord = np.full((5, 3), 0.011, dtype=np.float32)
    feat = np.arange(15, dtype=np.float32).reshape(5, 3)
    ct, ft = torch.from_numpy(coord).to(device), torch.from_numpy(feat).to(device)
    hits = np.bincount([int(ops.voxelize(ct, ft, [0, 5], 0.02, pooling_method="random", seed=s)["choice"][0]) for s in range(200)],
                       minlength=5)
    assert hits.min() >= 15 and hits.sum() == 200
    with pytest.raises(ValueError):
        ops.voxelize(ct, ft, [0, 5], 0.02, pooling_method="max")


def test_module_sparsify_options(device):
    """SpUNet.sparsify honours hash_method / pooling_method (v1m1:177-178) and a supplied grid_coord / sparse_shape
    (structure.py:213-224)."""
    from segmentation_amd import spunet
    sc = scenes.make_batch([5, 6], target_voxels=2000)
    c, f = torch.from_numpy(sc["coord"]).to(device), torch.from_numpy(sc["feat"]).to(device)
    net = spunet.SpUNet(channels=(32, 32, 32, 32, 32, 32, 32, 32), layers=(1,) * 8, hash_method="ravel")
    x, vox = net.sparsify(dict(coord=c, feat=f, offset=sc["offset"], grid_size=0.02))
    sp = oracle.sparsify(sc["coord"], sc["feat"], sc["offset"], 0.02, hash_method="ravel")
    assert np.array_equal(x.indices.cpu().numpy(), sp["voxel_coords"]) and x.spatial_shape == list(sp["spatial_shape"])
    # grid_coord as GridSample would hand it over, with a fixed sparse_shape
    gc = vox["voxel_coords"][vox["v2p_map"]][:, 1:]
    net2 = spunet.SpUNet(channels=(32, 32, 32, 32, 32, 32, 32, 32), layers=(1,) * 8)
    x2, vox2 = net2.sparsify(dict(grid_coord=gc, feat=f, offset=sc["offset"], sparse_shape=[512, 512, 256]))
    spf = oracle.sparsify(sc["coord"], sc["feat"], sc["offset"], 0.02)
    assert np.array_equal(x2.indices.cpu().numpy(), spf["voxel_coords"]) and x2.spatial_shape == [512, 512, 256]
    assert np.array_equal(vox2["v2p_map"].cpu().numpy(), spf["v2p_map"])
    with pytest.raises(ValueError):
        net2.sparsify(dict(grid_coord=gc + 1, feat=f, offset=sc["offset"]))


def test_deferred_completion_equals_the_blocking_call(device):
    """voxelize(defer=True) enqueues the same kernels and hands the sizes over through a pinned buffer: several calls
    may be in flight before the first is finished, in any order of completion, and each yields what the blocking
    call yields."""
    scs = [scenes.make_scene(seed=60 + i, target_voxels=t) for i, t in enumerate([3000, 9000, 500])]
    args = [(torch.from_numpy(s["coord"]).to(device), torch.from_numpy(s["feat"]).to(device),
             np.array([0, s["coord"].shape[0]], dtype=np.int64), 0.02) for s in scs]
    ref = [ops.voxelize(*a, return_keys=True) for a in args]
    fins = [ops.voxelize(*a, return_keys=True, defer=True) for a in args]
    for i in (2, 0, 1):
        got = fins[i]()
        assert got["n_voxels"] == ref[i]["n_voxels"] and got["spatial_shape"] == ref[i]["spatial_shape"]
        for k in ("voxel_coords", "voxel_feats", "v2p_map", "keys"):
            assert torch.equal(got[k], ref[i][k]), k
    with pytest.raises(ValueError):
        ops.voxelize(*args[0], pooling_method="random", defer=True)


@pytest.mark.parametrize("name", ["s2", "s3"])
@pytest.mark.parametrize("hm", ["fnv", "ravel"])
def test_reference_pooling_signature_on_the_gpu(golden_voxelize, device, name, hm):
    """compat.gpu_mean_pooling / gpu_random_pooling: the functions compat.install(gpu_voxelizer=True) puts in place of
    the reference's mean_pooling / random_pooling (structure.py:34-110), called the way Point.sparsify calls them
    (structure.py:226-237: [batch, grid] integer rows, features) -- against the reference's golden outputs."""
    from segmentation_amd import compat
    g = golden_voxelize
    pre = f"vox_{name}_{hm}_"
    batch = oracle.offset2batch(g[f"vox_{name}_offset"])
    coords = torch.from_numpy(np.concatenate([batch[:, None].astype(np.int32), g[pre + "grid_coord"]], 1)).to(device)
    feat = torch.from_numpy(g[f"vox_{name}_feat"]).to(device)
    vc, vf, v2p = compat.gpu_mean_pooling(coords, feat, return_inverse=True, hash_method=hm)
    assert vc.dtype == coords.dtype and v2p.dtype == torch.int64 and vf.dtype == feat.dtype
    assert np.array_equal(vc.cpu().numpy(), g[pre + "voxel_coords"])
    assert np.array_equal(v2p.cpu().numpy(), g[pre + "v2p_map"])
    sp = oracle.sparsify(g[f"vox_{name}_coord"], g[f"vox_{name}_feat"], g[f"vox_{name}_offset"],
                         float(g[f"vox_{name}_grid"][0]), hash_method=hm)
    np.testing.assert_allclose(vf.cpu().numpy(), sp["voxel_feats"], rtol=1e-6, atol=1e-7)
    two = compat.gpu_mean_pooling(coords, feat, return_inverse=False, hash_method=hm)
    assert len(two) == 2 and torch.equal(two[0], vc)
    rc, rf, rv = compat.gpu_random_pooling(coords, feat, hash_method=hm)
    assert torch.equal(rc, vc) and torch.equal(rv, v2p)
    # every voxel's features are those of one of its own points
    hit = torch.zeros(rc.shape[0], dtype=torch.bool, device=device)
    hit[rv[(feat - rf[rv]).abs().sum(1) == 0]] = True
    assert bool(hit.all())
    # inputs the wrappers do not take are refused, not mangled
    with pytest.raises(ValueError):
        compat.gpu_mean_pooling(coords + 1, feat, hash_method=hm)
    with pytest.raises(ValueError):
        compat.gpu_mean_pooling(coords.flip(0), feat, hash_method=hm)


@pytest.mark.gpu
def test_infinite_and_huge_coordinates_are_refused_and_nan_does_no_harm(device):
    """Coordinates whose grid cells span 2^31 or more along an axis wrap the 32-bit grid coordinates: the voxelizer
    reports it (meta[12]) and the wrapper raises instead of handing out a wrong grid.  A NaN coordinate converts to
    cell 0 on this hardware: the call completes and every other point keeps its voxel."""
    rng = np.random.default_rng(5)
    n = 4000
    coord = rng.uniform(0, 3, size=(n, 3)).astype(np.float32)
    feat = rng.normal(size=(n, 3)).astype(np.float32)
    good = ops.voxelize(torch.from_numpy(coord).to(device), torch.from_numpy(feat).to(device), [0, n], 0.02)
    for bad_value in (np.inf, -np.inf, 1e30, -3e38):
        c2 = coord.copy()
        c2[17, 1] = bad_value
        with pytest.raises(ValueError, match="2\\^31"):
            ops.voxelize(torch.from_numpy(c2).to(device), torch.from_numpy(feat).to(device), [0, n], 0.02)
        done = ops.voxelize(torch.from_numpy(c2).to(device), torch.from_numpy(feat).to(device), [0, n], 0.02, defer=True)
        with pytest.raises(ValueError, match="2\\^31"):
            done()
    c3 = coord.copy()
    c3[17, 1] = np.nan
    out = ops.voxelize(torch.from_numpy(c3).to(device), torch.from_numpy(feat).to(device), [0, n], 0.02)
    assert abs(out["n_voxels"] - good["n_voxels"]) <= 1
    keep = np.arange(n) != 17
    # the other points are grouped exactly as before (voxel ids may shift by the one voxel that changed)
    a, b = good["v2p_map"].cpu().numpy()[keep], out["v2p_map"].cpu().numpy()[keep]
    assert len(set(zip(a.tolist(), b.tolist()))) <= max(len(set(a.tolist())), len(set(b.tolist()))) + 1
    # the library is usable afterwards
    again = ops.voxelize(torch.from_numpy(coord).to(device), torch.from_numpy(feat).to(device), [0, n], 0.02)
    assert torch.equal(again["voxel_coords"], good["voxel_coords"]) and torch.equal(again["v2p_map"], good["v2p_map"])
