"""The reference's untouched forward code (spconv_unet_v1m3_pdnorm.SpUNetBase.forward, ppt.model.PPT.forward,
structure.Point.sparsify / Custom1x1Subm3d.forward, spconv_unet_v1m1_base.SpUNetBase.forward) executed on
segmentation_amd's classes in the build container, with the sparse convolutions evaluated by the CPU oracle.
This pins the wiring of oracle/spunet.py to the reference's real forward and proves the drop-in protocol
(lists passed through SparseSequential, Identity projections, indice_dict / grid hand-over, indices from the
reference's own voxelizer).  The golden file it reproduces is what the GPU tests compare the engine with."""
import os
import sys

import numpy as np
import pytest
import torch

import _refmodel

pytestmark = pytest.mark.skipif(not _refmodel.have_reference(), reason="reference tree not present")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


@pytest.fixture(scope="module")
def forward_run():
    import make_golden_forward as mg
    return mg.run()


def test_reference_forward_equals_the_oracle_and_the_committed_golden(forward_run):
    rec, outs, point, params, b, cfg = forward_run
    import oracle
    from oracle import spunet as ospunet
    sp = oracle.sparsify(b["coord"], b["feat"], b["offset"], 0.02)
    # the reference's own voxelizer (numpy, on its Point) and the oracle agree bit for bit
    assert np.array_equal(rec["voxel_coords"], sp["voxel_coords"])
    assert np.array_equal(rec["v2p_map"], sp["v2p_map"])
    assert list(rec["sparse_shape"]) == list(sp["spatial_shape"])
    p64 = {k: v.double() for k, v in params.items()}
    # voxel means in float64, as the reference's scatter(reduce="mean") forms them from the float64 features it was fed
    v2p = torch.from_numpy(sp["v2p_map"])
    n0 = sp["voxel_coords"].shape[0]
    vf = torch.zeros(n0, 3, dtype=torch.float64).index_add_(0, v2p, torch.from_numpy(b["feat"]).double())
    vf /= torch.bincount(v2p, minlength=n0).double().unsqueeze(1)
    assert np.allclose(vf.numpy(), sp["voxel_feats"], rtol=1e-6, atol=1e-7)
    ref, inter = ospunet.spunet_forward(p64, cfg, vf, sp["voxel_coords"],
                                        sp["spatial_shape"], "Structured3D", return_intermediates=True,
                                        dtype=torch.float64)
    assert set(inter) == set(outs)
    for name, f in outs.items():
        err = ((f - inter[name]).norm() / inter[name].norm()).item()
        assert err < 1e-12, (name, err)
    assert torch.equal(point.feat, outs["final"]) and point.sparse_conv_feat.features is point.feat
    golden = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_forward_v1m3.npz"))
    assert set(golden.files) == set(rec)
    for k in golden.files:
        if k.startswith("vals:") or k.startswith("norm:"):
            assert np.allclose(golden[k], rec[k], rtol=1e-12, atol=1e-14), k   # MKL thread counts may reorder sums
        else:
            assert np.array_equal(golden[k], rec[k]), k


def test_reference_loss_and_head_golden_regenerate_and_pin_the_oracle_head(forward_run):
    """tests/golden/reference_loss.npz is what the reference's own CaptionLoss / CLIPAlignmentEval produce today
    (regenerated here from /root/reference), and oracle.head_logits / head_predict reproduce its head part."""
    import make_golden_loss as mgl
    import oracle
    from segmentation_amd import scenes
    CL, CA = mgl.reference_loss_modules()
    rec = mgl.run_loss(CL)
    rec.update(mgl.run_head(CA, forward_run))
    golden = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_loss.npz"))
    assert set(golden.files) == set(rec)
    for k in golden.files:
        if golden[k].dtype.kind == "f":
            assert np.allclose(golden[k], rec[k], rtol=1e-10, atol=1e-13), k
        else:
            assert np.array_equal(golden[k], rec[k]), k
    _, outs, point, params, b, cfg = forward_run
    emb = torch.from_numpy(scenes.text_embeddings(mgl.HEAD_CLASSES, cfg["out_channels"])).double()
    feats, v2p = outs["final"], point.v2p_map.numpy()
    for tag, norm in (("raw", False), ("norm", True)):
        clip_feat = feats[torch.from_numpy(v2p)]
        if norm:
            clip_feat = torch.nn.functional.normalize(clip_feat, p=2, dim=1)
        logits = clip_feat @ emb.t()   # oracle.head_logits casts the text rows to float32; float64 here as the golden
        assert np.allclose(logits[torch.from_numpy(golden["head_rows"])].numpy(), golden[f"head_{tag}_logits"],
                           rtol=1e-12, atol=1e-12)
        got = oracle.head_predict(feats.float(), v2p, emb.float(), normalize_input=norm).numpy()
        differ = got != golden[f"head_{tag}_max"]
        assert differ.mean() < 1e-3 and (golden[f"head_{tag}_margin"][differ] < 1e-3 * float(logits.abs().max())).all()


def _v1m1_inputs(seed=31, target=2500):
    from segmentation_amd import scenes
    b = scenes.make_batch([seed, seed + 1], target_voxels=target)
    return dict(coord=torch.from_numpy(b["coord"]), feat=torch.from_numpy(b["feat"]).double(),
                offset=torch.from_numpy(b["offset"]), grid_size=0.02), b


def test_v1m1_backbone_runs_on_our_classes_with_fpn_grid_coord_and_pooling_options():
    """spconv_unet_v1m1_base.SpUNetBase (plain BatchNorm1d): the Point path with hash_method / pooling_method, the
    grid_coord input path (v1m1:319-332), out_fpn (:342-372) and the Bottleneck variant with algo=ConvAlgo.Native."""
    import oracle
    with _refmodel.reference_modules() as (v1m3, v1m1, ppt, structure):
        torch.manual_seed(0)
        kw = dict(in_channels=3, out_channels=20, base_channels=32, channels=[32, 64, 64, 96, 96, 64, 32, 32],
                  layers=[1, 1, 1, 1, 1, 1, 1, 1])
        batch, b = _v1m1_inputs()
        sp = oracle.sparsify(b["coord"], b["feat"], b["offset"], 0.02)
        with torch.no_grad(), _refmodel.CpuConvs():
            net = v1m1.SpUNetBase(out_fpn=True, **kw).double().eval()
            point, fmaps = net(dict(batch))
            assert np.array_equal(point.sparse_conv_feat.indices.numpy(), sp["voxel_coords"])
            assert len(fmaps) == 5 and fmaps[-1] is point.sparse_conv_feat
            assert [f.features.shape[1] for f in fmaps] == [96, 96, 64, 32, 20]
            # the same voxels handed over as grid_coord (already voxelised input, no Point): identical features
            vc = torch.from_numpy(sp["voxel_coords"])
            counts = np.bincount(sp["voxel_coords"][:, 0], minlength=2)
            # grid_coord input keeps the caller's row order, so sort rows by batch for the offset convention
            order = torch.from_numpy(np.argsort(sp["voxel_coords"][:, 0], kind="stable"))
            v2p = torch.from_numpy(sp["v2p_map"])
            vf = torch.zeros(vc.shape[0], 3, dtype=torch.float64).index_add_(0, v2p, batch["feat"])
            vf /= torch.bincount(v2p, minlength=vc.shape[0]).double().unsqueeze(1)   # float64 means, as the Point path forms
            gdict = dict(grid_coord=vc[order, 1:], feat=vf[order],
                         offset=torch.from_numpy(np.concatenate([[0], np.cumsum(counts)])))
            x2, fmaps2 = net(gdict)
            assert torch.allclose(x2.features, point.sparse_conv_feat.features[order], rtol=1e-10, atol=1e-12)
            # ravel hash: voxels in lexicographic (b, x, y, z) order, same set, same per-voxel result
            net.hash_method = "ravel"
            pr, _ = net(dict(batch))
            ir = pr.sparse_conv_feat.indices.numpy()
            assert np.array_equal(ir, ir[np.lexsort((ir[:, 3], ir[:, 2], ir[:, 1], ir[:, 0]))])
            key = lambda a: {tuple(r): i for i, r in enumerate(a.tolist())}
            kf, kr = key(sp["voxel_coords"]), key(ir)
            assert set(kf) == set(kr)
            perm = torch.tensor([kf[tuple(r)] for r in ir.tolist()])
            assert torch.allclose(pr.sparse_conv_feat.features, point.sparse_conv_feat.features[perm], rtol=1e-9, atol=1e-11)
            # random pooling: same voxels, order and v2p_map; every voxel feature is one of its points' features
            net.hash_method, net.pooling_method = "fnv", "random"
            np.random.seed(0)
            pt = structure.Point(dict(batch))
            pt.sparsify(pad=128, hash_method="fnv", pooling_method="random")
            assert np.array_equal(pt.sparse_conv_feat.indices.numpy(), sp["voxel_coords"])
            assert np.array_equal(pt.v2p_map.numpy(), sp["v2p_map"])
            vf, feat, v2p = pt.sparse_conv_feat.features, batch["feat"], pt.v2p_map
            d = (feat - vf[v2p]).abs().sum(1)
            hit = torch.zeros(vf.shape[0], dtype=torch.bool).index_put_((v2p[d == 0],), torch.tensor(True))
            assert bool(hit.all())
            # bottleneck blocks pass algo=spconv.ConvAlgo.Native to every convolution
            bott = v1m1.SpUNetBottleneck(in_channels=3, out_channels=20, base_channels=32, bottleneck_ratio=2,
                                         channels=[32, 64, 64, 96, 96, 64, 32, 32], layers=[1, 1, 1, 1, 1, 1, 1, 1]).double().eval()
            out = bott(dict(batch))
            assert out.sparse_conv_feat.features.shape == (sp["voxel_coords"].shape[0], 20)
            assert torch.isfinite(out.sparse_conv_feat.features).all()
