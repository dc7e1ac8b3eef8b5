"""The network as one library call per scene (m3d_net_create / m3d_net_run, driven by SpUNetEngine's traced layer list)
against the layer-by-layer calls of the same engine: the same kernels on the same operands, so labels, per-voxel
labels and logits must be equal bit for bit, in every precision, for scenes of different sizes run back to back
through one plan, for one scene at a time and in the scene pipeline."""
import ctypes

import numpy as np
import pytest
import torch

import bench
from segmentation_amd import _lib, scenes


def _scene(seed, target, device):
    sc = scenes.make_scene(seed=seed, target_voxels=target)
    n = sc["coord"].shape[0]
    return (torch.from_numpy(sc["coord"]).to(device), torch.from_numpy(sc["feat"]).to(device),
            np.array([0, n], dtype=np.int64), 0.02)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["bf16x6", "bf16x3", "f16x3", "bf16", "f32"])
def test_one_call_network_equals_layer_by_layer(device, prec):
    eng, _, _ = bench.build_engine(device, prec)
    assert eng.use_plan
    seq = [_scene(90 + i, t, device) for i, t in enumerate([6000, 45000, 900, 20000])]
    with torch.no_grad():
        for want_logits in (False, True):
            eng.use_plan = True
            got = [eng.predict(*sc, return_logits=want_logits) for sc in seq]
            assert eng._plan is not None
            eng.use_plan = False
            ref = [eng.predict(*sc, return_logits=want_logits) for sc in seq]
            for a, b in zip(got, ref):
                assert torch.equal(a["pred"], b["pred"]) and torch.equal(a["voxel_pred"], b["voxel_pred"])
                assert a["n_voxels"] == b["n_voxels"] and torch.equal(a["v2p_map"], b["v2p_map"])
                if want_logits:
                    assert torch.equal(a["voxel_logits"], b["voxel_logits"])
        eng.use_plan = True
        many = eng.predict_many(seq + seq, net_streams=3)
        for a, b in zip(many, ref + ref):
            assert torch.equal(a["pred"], b["pred"]) and torch.equal(a["voxel_pred"], b["voxel_pred"])
        # new text rows: the plan is rebuilt around them
        old_plan = eng._plan
        eng.set_text_embeddings(torch.from_numpy(scenes.text_embeddings(20, 768, seed=9)))
        a = eng.predict(*seq[0])
        assert eng._plan is not old_plan
        eng.use_plan = False
        b = eng.predict(*seq[0])
        assert torch.equal(a["pred"], b["pred"]) and int(a["pred"].max()) < 20


@pytest.mark.gpu
def test_labels_only_head_precision_through_the_plan(device):
    eng, _, _ = bench.build_engine(device, "bf16x6", label_head="bf16x3")
    sc = _scene(97, 15000, device)
    with torch.no_grad():
        a = eng.predict(*sc)
        eng.use_plan = False
        b = eng.predict(*sc)
    assert torch.equal(a["pred"], b["pred"]) and torch.equal(a["voxel_pred"], b["voxel_pred"])


def test_network_description_is_checked():
    """m3d_net_create refuses descriptions that do not hang together (no GPU work involved)."""
    lib = _lib.load()
    one = 8  # a non-null pointer that is never dereferenced
    vals = (_lib.NetValue * 3)(_lib.NetValue(0, 32, 0), _lib.NetValue(0, 32, 2), _lib.NetValue(1, 32, 2))
    h = ctypes.c_void_p()

    def layer(**kw):
        d = dict(op=0, in0=0, in1=-1, residual=-1, out_f32=-1, out_split=1, sched=1, k=27, cout=32, c0=32, c1=0, relu=0,
                 precision=1, n_classes=0, classes_padded=0, weight=one, scale=None, shift=None, text=None)
        d.update(kw)
        return _lib.NetLayer(**d)

    ok = (_lib.NetLayer * 1)(layer())
    assert lib.m3d_net_create(ok, 1, vals, 3, 2, ctypes.byref(h)) == 0
    rows = (ctypes.c_int64 * 2)(1000, 300)
    assert lib.m3d_net_arena_bytes(h, rows) >= 1001 * 32 * 4
    lib.m3d_net_destroy(h)
    for bad, msg in ((layer(out_split=5), b"out of range"), (layer(in0=2), b"no earlier layer"),
                     (layer(sched=2), b"level"), (layer(c0=64), b"channel"), (layer(out_split=-1), b"without an output"),
                     (layer(weight=None), b"null weight"), (layer(op=9), b"unknown op")):
        arr = (_lib.NetLayer * 1)(bad)
        assert lib.m3d_net_create(arr, 1, vals, 3, 2, ctypes.byref(h)) == -1
        assert msg in lib.m3d_last_error(), (msg, lib.m3d_last_error())


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["bf16x6", "bf16x3", "f16x3"])
def test_plain_f32_activation_levels_give_the_same_bits(device, prec, monkeypatch):
    """Levels whose activations travel as plain f32 rows (the default for level 0 at three terms) against the engine
    that keeps every level in the split format: the convolutions form the same terms after the fragment read, so
    features, logits and labels are the same bits, through the plan and layer by layer."""
    sc = _scene(98, 30000, device)
    outs = {}
    for levels in ("", "0", "0,1", "0,1,2,3,4"):
        monkeypatch.setenv("M3D_ACT_F32_LEVELS", levels)
        eng, _, _ = bench.build_engine(device, prec)
        assert eng.act_f32 == {int(v) for v in levels.split(",") if v}
        with torch.no_grad():
            o = eng.predict(*sc, return_logits=True)
            eng.use_plan = False
            o2 = eng.predict(*sc, return_logits=True)
            f, _ = eng.forward_features(*sc)
        assert torch.equal(o["pred"], o2["pred"]) and torch.equal(o["voxel_logits"], o2["voxel_logits"])
        outs[levels] = (o["pred"], o["voxel_logits"], f)
    for levels, (p, l, f) in outs.items():
        assert torch.equal(p, outs[""][0]) and torch.equal(l, outs[""][1]) and torch.equal(f, outs[""][2]), levels
    monkeypatch.delenv("M3D_ACT_F32_LEVELS")
    assert bench.build_engine(device, "bf16x6")[0].act_f32 == {0}
    assert bench.build_engine(device, "f16x3")[0].act_f32 == set()
