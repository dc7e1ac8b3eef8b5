"""The drop-in packages at the repository root: `import spconv.pytorch as spconv` and `from torch_scatter import
scatter, segment_csr` resolve to spconv/ and torch_scatter/ there, so the reference's model files
(mosaic3d/src/models/utils/structure.py:6,10, spconv_unet_v1m3_pdnorm.py:10,14) import with nothing but sys.path set
-- no call into segmentation_amd.compat -- and their forward gives the committed golden rows.  Also: the packages
and compat.build_modules() expose the same objects, and compat.install(gpu_voxelizer=True) rebinds the reference's
host voxelizer."""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

import _refmodel
from segmentation_amd import compat, sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def test_packages_and_compat_aliases_expose_the_same_objects():
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("spconv", "torch_scatter")}
    for k in saved:
        del sys.modules[k]
    try:
        for name, alias in compat.build_modules().items():
            real = importlib.import_module(name)
            assert os.path.dirname(real.__file__).startswith(ROOT), real.__file__
            for attr, obj in vars(alias).items():
                if attr.startswith("_"):
                    continue
                other = getattr(real, attr)
                if isinstance(obj, types.ModuleType):
                    assert isinstance(other, types.ModuleType)
                elif isinstance(obj, types.FunctionType) and obj.__module__.endswith("compat"):
                    assert other.__name__ == obj.__name__
                else:
                    assert other is obj, (name, attr)
        import spconv.pytorch as spconv
        from torch_scatter import scatter, segment_csr
        assert spconv.SubMConv3d is sparse.SubMConv3d and scatter is sparse.scatter and segment_csr is sparse.segment_csr
        assert spconv.ConvAlgo.Native is sparse.ConvAlgo.Native
        assert spconv.modules.is_spconv_module(spconv.SparseSequential())
    finally:
        for k in list(sys.modules):
            if k.split(".")[0] in ("spconv", "torch_scatter"):
                del sys.modules[k]
        sys.modules.update(saved)


@pytest.mark.skipif(not _refmodel.have_reference(), reason="reference tree not present")
def test_reference_forward_with_only_sys_path_set():
    """No compat.install(): the reference's `import spconv.pytorch` finds the package at the repository root."""
    import make_golden_forward as mg
    assert ROOT in sys.path
    with _refmodel.reference_modules(install_compat=False) as (v1m3, v1m1, ppt, structure):
        assert os.path.samefile(sys.modules["spconv"].__file__, os.path.join(ROOT, "spconv", "__init__.py"))
        assert os.path.samefile(sys.modules["torch_scatter"].__file__, os.path.join(ROOT, "torch_scatter", "__init__.py"))
        assert structure.spconv.SparseConvTensor is sparse.SparseConvTensor
        assert issubclass(structure.Custom1x1Subm3d, sparse.SparseConv3d)
    rec, *_ = mg.run(install_compat=False)
    golden = np.load(os.path.join(ROOT, "tests", "golden", "reference_forward_v1m3.npz"))
    assert set(golden.files) == set(rec)
    for k in golden.files:
        if k.startswith("vals:") or k.startswith("norm:"):
            assert np.allclose(golden[k], rec[k], rtol=1e-12, atol=1e-14), k
        else:
            assert np.array_equal(golden[k], rec[k]), k


@pytest.mark.skipif(not _refmodel.have_reference(), reason="reference tree not present")
@pytest.mark.parametrize("already_imported", [False, True])
def test_install_with_gpu_voxelizer_rebinds_the_reference_pooling(already_imported):
    if already_imported:
        with _refmodel.reference_modules() as (v1m3, v1m1, ppt, structure):
            assert structure.mean_pooling is not compat.gpu_mean_pooling
            compat.install(gpu_voxelizer=True)
            assert structure.mean_pooling is compat.gpu_mean_pooling
            assert structure.random_pooling is compat.gpu_random_pooling
    else:
        with _refmodel.reference_modules(gpu_voxelizer=True) as (v1m3, v1m1, ppt, structure):
            assert structure.mean_pooling is compat.gpu_mean_pooling
            assert structure.random_pooling is compat.gpu_random_pooling
            # Point.sparsify reaches the wrapper; without a GPU it says so instead of falling back
            p = structure.Point(dict(coord=torch.rand(10, 3), feat=torch.rand(10, 3), offset=torch.tensor([0, 10]),
                                     grid_size=0.1))
            with pytest.raises(RuntimeError, match="GPU voxelizer"):
                p.sparsify(pad=128)
    assert not any(isinstance(f, compat._PatchAfterImport) for f in sys.meta_path)
