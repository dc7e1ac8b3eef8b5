"""The reference's untouched model files import and construct against segmentation_amd.compat, and
their state_dict is key-for-key the one segmentation_amd.spunet exposes (so Mosaic3D checkpoints load).
Runs only where /root/reference exists (the build container); no forward pass, hence no GPU."""
import os

import pytest
import torch

REF = "/root/reference/mosaic3d"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


@pytest.fixture(scope="module")
def reference_modules():
    import _refmodel
    with _refmodel.reference_modules() as (v1m3, v1m1, ppt, structure):
        yield v1m3, ppt, structure


def test_reference_model_builds_on_our_modules_and_shares_the_state_dict(reference_modules):
    v1m3, ppt, structure = reference_modules
    from functools import partial
    from segmentation_amd import sparse, spunet
    cfg = spunet.SPUNET34C
    backbone = partial(v1m3.SpUNetBase, in_channels=3, out_channels=768, base_channels=32, channels=cfg["channels"],
                       layers=cfg["layers"], conditions=cfg["conditions"], norm_decouple=True, norm_adaptive=True,
                       norm_affine=True, zero_init=True)
    ref = ppt.PPT(backbone, list(cfg["conditions"]), 256)
    ours = spunet.PPT(spunet.SpUNet(), cfg["conditions"], 256)
    rs, os_ = ref.state_dict(), ours.state_dict()
    assert set(rs) == set(os_)
    for k in rs:
        assert tuple(rs[k].shape) == tuple(os_[k].shape), k
    ours.load_state_dict(rs)  # a reference checkpoint loads into our model unchanged
    # the reference's containers / 1x1 conv are built from our classes
    assert isinstance(ref.backbone.enc[0], sparse.SparseSequential)
    assert isinstance(ref.backbone.final, sparse.SparseConv3d)
    assert issubclass(structure.Custom1x1Subm3d, sparse.SparseConv3d)
    # reference init rules were applied through isinstance(m, spconv.SubMConv3d)
    assert abs(float(ref.backbone.enc[0].block0.conv1.weight.std()) - 0.02) < 2e-3


def test_engine_accepts_the_reference_state_dict_layout(reference_modules):
    """Folding reads exactly the reference's parameter names (no GPU needed for the fold itself)."""
    v1m3, ppt, _ = reference_modules
    from segmentation_amd.engine import _fold_norm
    norm = v1m3.PDBatchNorm(32, 256, conditions=("A", "B"), decouple=True, adaptive=True, affine=True)
    torch.nn.init.normal_(norm.modulation[1].weight, std=0.02)
    norm.bns[1].running_mean.normal_()
    norm.bns[1].running_var.uniform_(0.5, 1.5)
    norm.eval()
    sd = {"n." + k: v for k, v in norm.state_dict().items()}
    ctx = torch.randn(1, 256)
    x = torch.randn(10, 32)
    a, b = _fold_norm(sd, "n", 1, ctx)
    torch.testing.assert_close(x * a + b, norm(x, "B", ctx), rtol=1e-5, atol=1e-6)
