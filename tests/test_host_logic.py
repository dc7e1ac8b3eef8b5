"""Host-side logic that needs no GPU: norm folding, synthetic scenes, the spconv/torch_scatter aliases,
module construction and state-dict naming."""
import sys

import numpy as np
import pytest
import torch

from oracle import spunet as ospunet
from segmentation_amd import compat, scenes, sparse, spunet
from segmentation_amd.engine import _fold_norm


def test_fold_norm_matches_oracle_pdnorm():
    cfg = dict(ospunet.SPUNET34C, layers=(1,) * 8)
    P = ospunet.make_spunet_params(cfg, seed=3, trained_like=True)
    ctx = P["embedding_table.weight"][2:3]
    x = torch.randn(64, 96, generator=torch.Generator().manual_seed(0))
    ref = ospunet.pdnorm(P, "dec.0.block0.bn2", x, 2, ctx)
    a, b = _fold_norm(P, "dec.0.block0.bn2", 2, ctx)
    torch.testing.assert_close(x * a + b, ref, rtol=1e-5, atol=1e-6)


def test_scene_generator_is_seeded_and_centred():
    a = scenes.make_scene(seed=3, target_voxels=2000)
    b = scenes.make_scene(seed=3, target_voxels=2000)
    assert np.array_equal(a["coord"], b["coord"]) and np.array_equal(a["color"], b["color"])
    c = a["coord"]
    assert c.dtype == np.float32 and a["feat"].dtype == np.float32 and a["color"].dtype == np.uint8
    assert abs(c[:, 0].min() + c[:, 0].max()) < 1e-4 and abs(c[:, 2].min()) < 1e-6  # CenterShift(apply_z)
    assert a["feat"].min() >= -1.0 and a["feat"].max() <= 1.0
    e = scenes.text_embeddings(20, 768)
    np.testing.assert_allclose(np.linalg.norm(e, axis=1), 1.0, rtol=1e-5)


def test_compat_aliases_resolve():
    mods = compat.build_modules()
    spp = mods["spconv.pytorch"]
    for name in ("SparseConvTensor", "SubMConv3d", "SparseConv3d", "SparseInverseConv3d", "SparseSequential",
                 "SparseModule", "ConvAlgo"):
        assert hasattr(spp, name)
    assert spp.ConvAlgo.Native is sparse.ConvAlgo.Native
    assert spp.modules.is_spconv_module(sparse.SubMConv3d(4, 4, 3))
    assert not spp.modules.is_spconv_module(torch.nn.ReLU())
    assert hasattr(spp.core, "ImplicitGemmIndiceData")
    with pytest.raises(RuntimeError, match="no CPU path"):  # the op exists and, like every op, refuses CPU tensors
        spp.ops.indice_avgpool_implicit_gemm(torch.zeros(4, 2), torch.zeros((8, 2), dtype=torch.int32), 2, False)
    # the reference's isinstance-based init must not treat the other conv classes as SubMConv3d
    assert not isinstance(sparse.SparseConv3d(4, 4, 2, 2), sparse.SubMConv3d)
    assert not isinstance(sparse.SparseInverseConv3d(4, 4, 2, "k"), sparse.SubMConv3d)


def test_scatter_and_segment_csr():
    src = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0], [7.0, 8.0]])
    idx = torch.tensor([1, 0, 1, 1])
    out = sparse.scatter(src, idx, dim=0, reduce="mean")
    torch.testing.assert_close(out, torch.tensor([[3.0, 4.0], [13.0 / 3, 16.0 / 3]]))
    out = sparse.scatter(src, idx, dim=0, reduce="sum")
    torch.testing.assert_close(out, torch.tensor([[3.0, 4.0], [13.0, 16.0]]))
    seg = sparse.segment_csr(src, torch.tensor([0, 1, 4]), reduce="mean")
    torch.testing.assert_close(seg, torch.tensor([[1.0, 2.0], [5.0, 6.0]]))


def test_sparse_sequential_passes_lists_to_sparse_modules():
    class Tag(sparse.SparseModule):
        def forward(self, x):
            t, cond, ctx = x
            return t.replace_feature(t.features + 1), cond, ctx

    seq = sparse.SparseSequential(Tag(), Tag())
    t = sparse.SparseConvTensor(torch.zeros(3, 2), torch.zeros(3, 4, dtype=torch.int32), [4, 4, 4], 1)
    out, cond, ctx = seq([t, "ScanNet", None])
    assert cond == "ScanNet" and torch.equal(out.features, torch.full((3, 2), 2.0))
    assert out.indice_dict is t.indice_dict  # replace_feature shares the kernel-map cache
    seq2 = sparse.SparseSequential(torch.nn.ReLU())
    assert torch.equal(seq2(t.replace_feature(torch.tensor([[-1.0, 2.0]] * 3))).features[0], torch.tensor([0.0, 2.0]))


def test_model_state_dict_matches_oracle_parameter_names():
    cfg = ospunet.SPUNET34C
    P = ospunet.make_spunet_params(cfg, seed=0)
    model = spunet.PPT(spunet.SpUNet(), cfg["conditions"], 256)
    sd = model.state_dict()
    want = {("backbone." + k if not k.startswith("embedding_table") else k) for k in P}
    have = {k for k in sd if not k.endswith("num_batches_tracked")}
    assert want == have
    for k, v in P.items():
        name = "backbone." + k if not k.startswith("embedding_table") else k
        assert tuple(sd[name].shape) == tuple(v.shape), name
    spunet.load_oracle_params(model, P)
    # reference init rules (v1m3:362-378): SubM convs trunc_normal(0.02), modulation zero-initialised
    m2 = spunet.SpUNet()
    assert abs(float(m2.enc[0].block0.conv1.weight.std()) - 0.02) < 2e-3
    assert float(m2.enc[0].block0.bn1.modulation[1].weight.abs().max()) == 0.0
    n_params = sum(p.numel() for p in m2.parameters())
    assert 35e6 < n_params < 60e6


def test_pred_record_and_label_ply_round_trip(tmp_path):
    import numpy as np
    import torch
    from segmentation_amd import export
    rng = np.random.default_rng(0)
    coord = torch.from_numpy(rng.normal(size=(100, 3)).astype(np.float32))
    pred = torch.from_numpy(rng.integers(0, 7, size=100))
    logits = torch.from_numpy(rng.normal(size=(100, 7)).astype(np.float32))
    p = export.save_pred(str(tmp_path / "pred" / "pred_0.pth"), coord, torch.tensor([100]), pred, logits,
                         class_names=[f"c{i}" for i in range(7)])
    rec = torch.load(p, map_location="cpu", weights_only=True)
    assert sorted(rec) == ["class_names", "color", "coord", "logits", "pc_count", "pred"]  # language_module.py:349-356
    pts, lab = export.load_pred_for_ply(p)
    assert np.array_equal(pts, coord.numpy()) and np.array_equal(lab, pred.numpy())
    ply = export.write_label_ply(str(tmp_path / "final_result.ply"), pts, lab, seed=3)
    raw = open(ply, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert b"element vertex 100" in head and len(body) == 100 * 15
    v = np.frombuffer(body, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    assert np.array_equal(v["x"], coord.numpy()[:, 0])
    cm = export.class_colormap(int(lab.max()) + 1, 3)
    assert np.array_equal(v["r"], (cm[lab, 0] * 255).astype(np.uint8))


def test_tensor_walker_reaches_slots_dataclasses_and_containers():
    """predict_many marks every map tensor of a scene as used on the network's stream before letting go of it; a
    tensor the walker misses would be handed out again while kernels still read it."""
    import torch
    from segmentation_amd import engine, ops
    lv = engine._Level()                      # a __slots__ class
    lv.indices, lv.shape, lv.n = torch.zeros(3), [1, 2, 3], 3
    lv.subm = ops.Schedule(torch.zeros(2), torch.zeros(2), torch.zeros(2), 1, 256, 27)   # a dataclass
    lv.subm.tile_order = torch.zeros(1)
    lv.pair_fwd = lv.indices                  # the same tensor twice: reported once
    vox = {"voxel_coords": torch.zeros(4), "spatial_shape": [4, 4, 4], "nested": (torch.zeros(1), [torch.zeros(1)])}
    got = list(engine._tensors_of((vox, [lv]), cuda_only=False))
    assert len(got) == 8 and len({id(t) for t in got}) == 8
    assert list(engine._tensors_of((vox, [lv]))) == []   # CPU tensors are not stream-managed


def test_mix3d_offset_is_the_reference_line_for_both_offset_conventions():
    """language_module.py:212-214 / collate.py:154-157 applied literally; the voxelizer accepts either result."""
    import torch
    from segmentation_amd import train
    import oracle
    # offsets without a leading zero: samples merged in pairs (an odd last sample stays alone)
    assert train.mix3d_offset(torch.tensor([3, 5, 9, 12])).tolist() == [5, 12]
    assert train.mix3d_offset(torch.tensor([3, 5, 9])).tolist() == [5, 9]
    assert train.mix3d_offset(torch.tensor([3, 5])).tolist() == [5]
    # offsets as the reference's collate builds them (leading zero, collate.py:96-102): what the reference computes
    assert train.mix3d_offset(torch.tensor([0, 3, 5, 9, 12])).tolist() == [3, 9, 12]
    assert train.mix3d_offset(torch.tensor([0, 3, 5, 9])).tolist() == [3, 9]
    assert train.mix3d_offset(torch.tensor([0, 3, 5])).tolist() == [3, 5]
    # ... which offset2bincount (misc.py:8-13) reads as the samples [0,3) [3,9) [9,12)
    assert oracle.offset2bincount(np.array([3, 9, 12])).tolist() == [3, 6, 3]


def test_operators_refuse_a_second_host_thread(monkeypatch):
    """ops._raw_stream() (every launch and every workspace lookup goes through it) remembers the first thread per
    device and raises from any other, unless M3D_ALLOW_THREADS=1."""
    import threading
    import torch
    from segmentation_amd import ops
    monkeypatch.setattr(ops, "_OWNER_THREAD", {})
    monkeypatch.setattr(torch._C, "_cuda_getDevice", lambda: 5, raising=False)
    monkeypatch.setattr(torch._C, "_cuda_getCurrentRawStream", lambda dev: 1234, raising=False)
    assert ops._raw_stream() == 1234 and ops._raw_stream() == 1234     # the owner may call as often as it likes
    result = {}

    def other():
        try:
            ops._stream()
            result["error"] = None
        except RuntimeError as e:
            result["error"] = str(e)

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert result["error"] is not None and "one host thread" in result["error"] and "cuda:5" in result["error"]
    # another device has no owner yet: the second thread may take it
    monkeypatch.setattr(torch._C, "_cuda_getDevice", lambda: 6, raising=False)
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert result["error"] is None
    with pytest.raises(RuntimeError):
        ops._raw_stream()   # ... and the first thread is now the stranger there
    monkeypatch.setattr(ops, "_ALLOW_THREADS", True)
    assert ops._raw_stream() == 1234
