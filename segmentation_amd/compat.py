"""Make `import spconv.pytorch as spconv` and `from torch_scatter import scatter, segment_csr`
resolve to this package, so that Mosaic3D's model files run untouched
(imports at mosaic3d/src/models/utils/structure.py:6,10 and
mosaic3d/src/models/networks/spunet/spconv_unet_v1m3_pdnorm.py:10,14).

    import segmentation_amd.compat as compat; compat.install()
"""
import sys
import types

from . import sparse


def indice_avgpool_implicit_gemm(features, pair_fwd, num_out, calc_count=False):
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
    opsm.indice_avgpool_implicit_gemm = indice_avgpool_implicit_gemm
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


_STRUCTURE = "src.models.utils.structure"


def install(force=False, gpu_voxelizer=False):
    """Register the alias modules in sys.modules (no-op for names that already resolve, unless force).  The same
    surface also exists as real packages next to this one (spconv/, torch_scatter/ at the repository root): with that
    directory on sys.path the reference imports them without this call.

    gpu_voxelizer=True additionally replaces the reference's host voxelizer: mean_pooling / random_pooling of
    mosaic3d/src/models/utils/structure.py:34-110 (a device-to-host copy, numpy FNV keys, argsort, unique and a copy
    back -- the dominant host cost of the reference's forward) are rebound to wrappers over ops.voxelize, with the
    same arguments and bit-identical voxel order, coordinates and v2p_map.  The module is patched now if it is
    already imported, otherwise right after its import."""
    for name, mod in build_modules().items():
        if force or name not in sys.modules:
            sys.modules[name] = mod
    if gpu_voxelizer:
        if _STRUCTURE in sys.modules:
            patch_voxelizer(sys.modules[_STRUCTURE])
        elif not any(isinstance(f, _PatchAfterImport) for f in sys.meta_path):
            sys.meta_path.insert(0, _PatchAfterImport())


def _pooling_on_gpu(pooling_method):
    def pooling(coordinates, features, return_inverse=True, hash_method="fnv"):
        """coordinates [N, 4] int (batch, x, y, z) with rows grouped by batch and every spatial minimum 0 -- what
        Point.sparsify passes (structure.py:219-228); features [N, C]."""
        import torch
        from . import ops
        if not (coordinates.is_cuda and features.is_cuda):
            raise RuntimeError("the GPU voxelizer needs its inputs on the GPU (compat.install(gpu_voxelizer=False) "
                               "keeps the reference's host voxelizer)")
        batch = coordinates[:, 0]
        lo, hi = coordinates.min(0)[0], coordinates.max(0)[0]
        ordered = bool((batch[1:] >= batch[:-1]).all()) if batch.numel() > 1 else True
        if not ordered or int(lo.min()) < 0 or int(lo[1:].max()) != 0 or int(hi[1:].max()) >= 1 << 24:
            raise ValueError("GPU voxelizer: rows must be grouped by batch index and the grid coordinates must start "
                             "at 0 on every axis and stay below 2**24 (as Point.sparsify passes them)")
        counts = torch.bincount(batch.long(), minlength=int(hi[0]) + 1)
        offset = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
        # integer grid coordinates below 2**24 are exact in float32: grid size 1 with the truncating rule returns them
        vox = ops.voxelize(coordinates[:, 1:].float(), features.float(), offset, 1.0, pad=0, hash_method=hash_method,
                           pooling_method=pooling_method)
        voxel_coords = vox["voxel_coords"].to(coordinates.dtype)
        voxel_feats = vox["voxel_feats"].to(features.dtype)
        if not return_inverse:
            return voxel_coords, voxel_feats
        return voxel_coords, voxel_feats, vox["v2p_map"]

    pooling.__name__ = pooling_method + "_pooling"
    return pooling


gpu_mean_pooling = _pooling_on_gpu("mean")
gpu_random_pooling = _pooling_on_gpu("random")


def patch_voxelizer(structure_module):
    """Rebind mean_pooling / random_pooling of an imported mosaic3d/src/models/utils/structure.py to the GPU
    wrappers (Point.sparsify looks them up as module globals at call time, structure.py:230-237)."""
    structure_module.mean_pooling = gpu_mean_pooling
    structure_module.random_pooling = gpu_random_pooling
    return structure_module


class _PatchAfterImport:
    """Meta-path hook: lets the normal finders locate src.models.utils.structure and patches it once executed."""

    def find_spec(self, name, path=None, target=None):
        if name != _STRUCTURE:
            return None
        import importlib.machinery
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(name, path, target)
            if spec is None or spec.loader is None:
                continue
            loader, execute = spec.loader, spec.loader.exec_module

            def exec_module(module, _execute=execute):
                _execute(module)
                patch_voxelizer(module)

            try:
                loader.exec_module = exec_module
            except AttributeError:
                return None
            return spec
        return None
