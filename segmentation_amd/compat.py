"""Make `import spconv.pytorch as spconv` and `from torch_scatter import scatter, segment_csr`
resolve to this package, so that Mosaic3D's model files run untouched
(imports at mosaic3d/src/models/utils/structure.py:6,10 and
mosaic3d/src/models/networks/spunet/spconv_unet_v1m3_pdnorm.py:10,14).

    import segmentation_amd.compat as compat; compat.install()
"""
import sys
import types

from . import sparse


def _indice_avgpool_implicit_gemm(features, pair_fwd, num_out, calc_count=False):
    """spconv.pytorch.ops.indice_avgpool_implicit_gemm as OpenSegment3D's decoder calls it
    (mosaic3d/src/models/networks/opensegment3d/model.py:274-276, :301-306)."""
    from . import ops
    return ops.avgpool(features, pair_fwd, num_out, calc_count)


def build_modules():
    sp = types.ModuleType("spconv")
    spp = types.ModuleType("spconv.pytorch")
    for name in ("SparseConvTensor", "SubMConv3d", "SparseConv3d", "SparseInverseConv3d", "SparseSequential",
                 "SparseModule", "ConvAlgo"):
        setattr(spp, name, getattr(sparse, name))
    mods = types.ModuleType("spconv.pytorch.modules")
    mods.is_spconv_module = sparse.is_spconv_module
    mods.SparseModule = sparse.SparseModule
    mods.SparseSequential = sparse.SparseSequential
    opsm = types.ModuleType("spconv.pytorch.ops")
    opsm.indice_avgpool_implicit_gemm = _indice_avgpool_implicit_gemm
    core = types.ModuleType("spconv.pytorch.core")
    core.SparseConvTensor = sparse.SparseConvTensor
    core.ImplicitGemmIndiceData = sparse.ImplicitGemmIndiceData
    spp.modules, spp.ops, spp.core = mods, opsm, core
    sp.pytorch = spp
    sp.__version__ = "2.3.6+segmentation_amd"
    ts = types.ModuleType("torch_scatter")
    ts.scatter = sparse.scatter
    ts.segment_csr = sparse.segment_csr
    return {"spconv": sp, "spconv.pytorch": spp, "spconv.pytorch.modules": mods, "spconv.pytorch.ops": opsm,
            "spconv.pytorch.core": core, "torch_scatter": ts}


def install(force=False):
    """Register the alias modules in sys.modules (no-op for names that already resolve, unless force)."""
    for name, mod in build_modules().items():
        if force or name not in sys.modules:
            sys.modules[name] = mod
