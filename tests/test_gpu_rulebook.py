"""GPU kernel maps (m3d_subm_rulebook / m3d_strided_rulebook / m3d_build_schedule) against the CPU
oracle: bit-exact."""
import numpy as np
import pytest
import torch

import oracle
from segmentation_amd import ops, scenes

pytestmark = pytest.mark.gpu


def _voxels(seed, target, batch=1):
    if batch == 1:
        sc = scenes.make_scene(seed=seed, target_voxels=target)
        off = np.array([0, sc["coord"].shape[0]], dtype=np.int64)
        return oracle.sparsify(sc["coord"], sc["feat"], off, 0.02)
    b = scenes.make_batch(range(seed, seed + batch), target_voxels=target)
    return oracle.sparsify(b["coord"], b["feat"], b["offset"], 0.02)


# lookup: "bitmap" = batch size known and the grid small enough for the occupancy bitmap, "hash" = batch size
# unknown (0): the open-addressing table
@pytest.mark.parametrize("lookup", ["bitmap", "hash"])
@pytest.mark.parametrize("seed,target,batch", [(0, 2000, 1), (1, 40000, 1), (2, 5000, 3)])
def test_subm_rulebook_bit_exact(device, seed, target, batch, lookup):
    sp = _voxels(seed, target, batch)
    idx = torch.from_numpy(sp["voxel_coords"]).to(device)
    pair = ops.subm_rulebook(idx, sp["spatial_shape"], sp["batch_size"] if lookup == "bitmap" else 0)
    ref = oracle.subm_rulebook(sp["voxel_coords"], sp["spatial_shape"])
    assert np.array_equal(pair.cpu().numpy(), ref)
    # the levels below the first come in ascending linear order (bit rank == row)
    oi, oshape, _, _ = ops.strided_rulebook(idx, sp["spatial_shape"], sp["batch_size"])
    pair1 = ops.subm_rulebook(oi.contiguous(), oshape, sp["batch_size"] if lookup == "bitmap" else 0)
    assert np.array_equal(pair1.cpu().numpy(), oracle.subm_rulebook(oi.cpu().numpy(), oshape))


@pytest.mark.parametrize("batch_size", [0, 2])
def test_subm_rulebook_respects_bounds_and_batches(device, batch_size):
    # neighbours across the spatial boundary or in another batch must not pair
    idx = np.array([[0, 0, 0, 0], [0, 0, 0, 1], [1, 0, 0, 1], [0, 3, 3, 3], [1, 3, 3, 2]], dtype=np.int32)
    pair = ops.subm_rulebook(torch.from_numpy(idx).to(device), [4, 4, 4], batch_size)
    assert np.array_equal(pair.cpu().numpy(), oracle.subm_rulebook(idx, [4, 4, 4]))


def test_subm_rulebook_large_grid_falls_back_to_the_hash_table(device):
    # 2^30 cells: no bitmap; rows far apart and at the upper corner of the grid
    idx = np.array([[0, 1023, 1023, 1023], [0, 1023, 1023, 1022], [0, 0, 0, 0], [0, 1, 1, 1], [0, 512, 512, 512]], dtype=np.int32)
    pair = ops.subm_rulebook(torch.from_numpy(idx).to(device), [1024, 1024, 1024], 1)
    assert np.array_equal(pair.cpu().numpy(), oracle.subm_rulebook(idx, [1024, 1024, 1024]))


@pytest.mark.parametrize("seed,target,batch", [(0, 2000, 1), (1, 40000, 1), (2, 5000, 3)])
def test_strided_rulebook_chain_bit_exact(device, seed, target, batch):
    sp = _voxels(seed, target, batch)
    idx_ref, shape = sp["voxel_coords"], sp["spatial_shape"]
    idx = torch.from_numpy(idx_ref).to(device)
    for level in range(4):
        oi, oshape, pf, pb = ops.strided_rulebook(idx, shape, sp["batch_size"])
        roi, roshape, rpf, rpb = oracle.strided_rulebook(idx_ref, shape)
        assert oshape == roshape
        assert np.array_equal(oi.cpu().numpy(), roi)
        assert np.array_equal(pf.cpu().numpy(), rpf)
        assert np.array_equal(pb.cpu().numpy(), rpb)
        idx, idx_ref, shape = oi.contiguous(), roi, oshape


def test_strided_rulebook_drops_rows_outside_the_output_shape(device):
    idx = np.array([[0, 4, 0, 0], [0, 0, 0, 0], [0, 1, 1, 1], [0, 2, 3, 2]], dtype=np.int32)  # shape 5 -> out 2
    oi, oshape, pf, pb = ops.strided_rulebook(torch.from_numpy(idx).to(device), [5, 5, 5], 1)
    roi, roshape, rpf, rpb = oracle.strided_rulebook(idx, [5, 5, 5])
    assert np.array_equal(oi.cpu().numpy(), roi) and np.array_equal(pf.cpu().numpy(), rpf)
    assert np.array_equal(pb.cpu().numpy(), rpb)


def test_strided_rulebook_large_grid_uses_the_sort_path(device):
    # output grid of 2048^3 cells: too large for the occupancy bitmap, the hash + sort variant runs
    rng = np.random.default_rng(5)
    base = rng.integers(0, 4090, size=(3000, 3))
    pts = np.concatenate([base, base + rng.integers(0, 2, size=base.shape), base[:500]], 0)
    idx = np.concatenate([rng.integers(0, 2, size=(len(pts), 1)), pts], 1).astype(np.int32)
    idx = np.unique(idx, axis=0)
    idx = idx[rng.permutation(len(idx))]
    oi, oshape, pf, pb = ops.strided_rulebook(torch.from_numpy(idx).to(device), [4096, 4096, 4096], 2)
    roi, roshape, rpf, rpb = oracle.strided_rulebook(idx, [4096, 4096, 4096])
    assert oshape == roshape and np.array_equal(oi.cpu().numpy(), roi)
    assert np.array_equal(pf.cpu().numpy(), rpf) and np.array_equal(pb.cpu().numpy(), rpb)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_schedule_is_a_permutation_of_the_kernel_map(device, mode):
    sp = _voxels(3, 9000)
    idx = torch.from_numpy(sp["voxel_coords"]).to(device)
    n = idx.shape[0]
    if mode == 2:
        _, _, _, pair = ops.strided_rulebook(idx, sp["spatial_shape"], 1)
        base = ops.build_schedule(pair, n, idx, 1, sp["spatial_shape"], 1).out_row
        sched = ops.build_schedule(pair, n, None, 2, base_order=base)
    else:
        pair = ops.subm_rulebook(idx, sp["spatial_shape"])
        sched = ops.build_schedule(pair, n, idx if mode else None, mode, sp["spatial_shape"], 1)
    torch.cuda.synchronize()
    orow = sched.out_row.cpu().numpy()
    nbr = sched.nbr.cpu().numpy()
    mask = sched.tile_mask.cpu().numpy().view(np.uint32)
    p = pair.cpu().numpy()
    assert sched.npad % 128 == 0 and sched.npad >= n
    assert np.array_equal(np.sort(orow[:n]), np.arange(n)) and (orow[n:] == -1).all()
    assert np.array_equal(nbr[:, :n], p[:, orow[:n]]) and (nbr[:, n:] == -1).all()
    ref_mask = np.zeros(sched.npad // 32, dtype=np.uint32)
    valid = (nbr >= 0).reshape(nbr.shape[0], -1, 32).any(2)
    for k in range(nbr.shape[0]):
        ref_mask |= (valid[k].astype(np.uint32) << np.uint32(k))
    assert np.array_equal(mask, ref_mask)
    if mode == 2:  # rows grouped by their single active kernel offset, Morton order kept inside a group
        kk = np.argmax(p[:, orow[:n]] >= 0, axis=0)
        assert (np.diff(kk) >= 0).all()
        pos = np.empty(n, dtype=np.int64)
        pos[base.cpu().numpy()[:n]] = np.arange(n)
        for k in range(8):
            assert (np.diff(pos[orow[:n][kk == k]]) > 0).all()
    if mode == 1:  # Morton order: keys non-decreasing
        from tools.mask_study_helpers import morton_key
        mk = morton_key(sp["voxel_coords"][orow[:n]])
        assert (np.diff(mk.astype(np.float64)) >= 0).all()
    if mode == 3:  # coarse cells (64 voxels per side) in Morton order, inside a cell by corner/edge pattern
        from tools.mask_study_helpers import morton_key
        nz = lambda k: (k // 9 != 1) + ((k // 3) % 3 != 1) + (k % 3 != 1)
        order_bits = [k for k in range(27) if nz(k) == 3] + [k for k in range(27) if nz(k) == 2]
        pat = np.zeros(n, dtype=np.uint64)
        for k in order_bits:
            pat = (pat << np.uint64(1)) | (p[k] >= 0).astype(np.uint64)
        vc = sp["voxel_coords"].copy()
        vc[:, 1:] >>= 6
        key = (morton_key(vc).astype(np.uint64) << np.uint64(20)) | pat
        assert np.array_equal(orow[:n], np.argsort(key, kind="stable"))


def test_avgpool_over_the_strided_map(device):
    """ops.avgpool (spconv.pytorch.ops.indice_avgpool_implicit_gemm) against the oracle; parity unpinned by the
    reference (spconv absent), 1e-6 relative."""
    from segmentation_amd import compat
    sp = _voxels(7, 6000)
    idx = torch.from_numpy(sp["voxel_coords"]).to(device)
    oi, oshape, pf, pb = ops.strided_rulebook(idx, sp["spatial_shape"], 1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(idx.shape[0], 37, generator=g)
    fn = compat.build_modules()["spconv.pytorch.ops"].indice_avgpool_implicit_gemm
    out, cnt = fn(x.to(device), pf, pf.shape[1], True)
    ref, rcnt = oracle.sparse_conv.avgpool(x, pf.cpu().numpy())
    assert out.shape == (oi.shape[0], 37) and np.array_equal(cnt.cpu().numpy(), rcnt.numpy())
    assert ((out.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 1e-6
    out2, none = fn(x[:, 0].to(device), pf, pf.shape[1], False)
    assert none is None and torch.allclose(out2.cpu(), out[:, 0].cpu())


def test_tile_order_is_heaviest_first_and_changes_no_result(device):
    """m3d_schedule_tile_order: a permutation of the 128-row tiles, three classes by active-offset count (thirds of
    the tiles, heaviest first), schedule order inside a class; a convolution run with it gives bit-identical output."""
    sc = scenes.make_scene(seed=4, target_voxels=30000)
    n = sc["coord"].shape[0]
    sp = oracle.sparsify(sc["coord"], sc["feat"], np.array([0, n]), 0.02)
    idx = torch.from_numpy(sp["voxel_coords"]).to(device)
    pair = ops.subm_rulebook(idx, sp["spatial_shape"])
    nv = idx.shape[0]
    plain = ops.build_schedule(pair, nv, idx, 3 | (6 << 8), sp["spatial_shape"], 1)
    sched = ops.build_schedule(pair, nv, idx, 3 | (6 << 8), sp["spatial_shape"], 1, tile_order=True)
    assert torch.equal(plain.out_row, sched.out_row) and plain.tile_order is None
    order = sched.tile_order.cpu().numpy()
    nt = sched.npad // 128
    assert order.shape == (nt,) and np.array_equal(np.sort(order), np.arange(nt))
    m = sched.tile_mask.cpu().numpy().view(np.uint32).reshape(nt, 4)
    u = m[:, 0] | m[:, 1] | m[:, 2] | m[:, 3]
    work = np.array([bin(int(v)).count("1") for v in u])
    hist = np.bincount(work, minlength=28)
    cls, seen = np.zeros(28, int), 0
    for wk in range(27, -1, -1):  # thirds of the tiles, heaviest offset counts first
        cls[wk] = 0 if 3 * seen < nt else (1 if 3 * seen < 2 * nt else 2)
        seen += hist[wk]
    expect = np.argsort(cls[work], kind="stable")
    assert np.array_equal(order, expect)
    assert work[order[: nt // 4]].mean() > work[order[-(nt // 4):]].mean()
    x = ops.split_f32(torch.randn(nv, 96, device=device), 2)
    w = ops.pack_weight(torch.randn(96, 27, 96) * 0.05, "bf16x3", device)
    a = ops.conv_fwd(x, w, plain, nv)
    b = ops.conv_fwd(x, w, sched, nv)
    assert torch.equal(a, b)


@pytest.mark.parametrize("seeds,target", [([3], 30000), ([4, 5], 6000), ([6], 700)])
def test_level_chain_in_two_calls_equals_level_by_level(device, seeds, target):
    """m3d_levels_begin / m3d_levels_finish (all maps and schedules of a scene in two library calls, arrays carved
    out of two buffers) against the level-by-level entry points: coordinates, kernel maps, schedules, tile masks and
    launch orders equal bit for bit on every level."""
    import bench
    eng, _, _ = bench.build_engine(device, "bf16x6")
    b = scenes.make_batch(seeds, target_voxels=target)
    vox = ops.voxelize(torch.from_numpy(b["coord"]).to(device), torch.from_numpy(b["feat"]).to(device), b["offset"], 0.02)
    assert eng.use_level_chain
    got_one = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], vox["batch_size"])
    # the independent parts of the two calls spread over three more streams (`parts` / `level_mask` of the entry points)
    aux = [torch.cuda.Stream(device=device) for _ in range(3)]
    modes = [eng.sched_modes[min(l, len(eng.sched_modes) - 1)] for l in range(5)]
    dicts = ops.level_chain_begin(vox["voxel_coords"], vox["spatial_shape"], vox["batch_size"], 4, modes, True,
                                  aux_streams=aux)()
    got_streams = [type("L", (), dict(d, down=d.get("down"), up=d.get("up"), pair_fwd=d.get("pair_fwd"),
                                      pair_bwd=d.get("pair_bwd"), next_indices=d.get("next_indices"),
                                      next_shape=d.get("next_shape"), n_next=d.get("n_next", 0)))() for d in dicts]
    eng.use_level_chain = False
    ref = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], vox["batch_size"])
    torch.cuda.synchronize()
    assert len(got_streams) == len(got_one) == len(ref) == 5
    for l, (a, r) in enumerate(list(zip(got_streams, ref)) + list(zip(got_one, ref))):
        l %= 5
        assert isinstance(a.subm, ops.ArenaSchedule) and not isinstance(r.subm, ops.ArenaSchedule)
        assert a.n == r.n and list(a.shape) == list(r.shape) and a.n_next == r.n_next, l
        assert torch.equal(a.indices, r.indices) and torch.equal(a.pair_subm, r.pair_subm), l
        pairs = [(a.subm, r.subm)]
        if r.down is not None:
            assert torch.equal(a.pair_fwd, r.pair_fwd) and torch.equal(a.pair_bwd, r.pair_bwd), l
            assert torch.equal(a.next_indices, r.next_indices) and list(a.next_shape) == list(r.next_shape), l
            pairs += [(a.down, r.down), (a.up, r.up)]
        else:
            assert a.down is None and a.up is None and a.pair_fwd is None
        for x, y in pairs:
            assert (x.n, x.npad, x.k) == (y.n, y.npad, y.k), l
            assert torch.equal(x.nbr, y.nbr) and torch.equal(x.out_row, y.out_row) and torch.equal(x.tile_mask, y.tile_mask), l
            assert (x.tile_order is None) == (y.tile_order is None), l
            if y.tile_order is not None:
                assert torch.equal(x.tile_order, y.tile_order), l
