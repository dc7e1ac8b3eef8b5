"""Helpers shared by the tests that run the reference's own model files (build container only; the reference
tree is never read on the GPU box): import them against segmentation_amd.compat with annotation/container-only
stand-ins for the packages that are not installed, and run sparse convolutions through the CPU oracle."""
import contextlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/mosaic3d"


@contextlib.contextmanager
def reference_modules(install_compat=True, gpu_voxelizer=False):
    """Yields (v1m3, v1m1, ppt, structure) imported from /root/reference with spconv / torch_scatter resolved to
    segmentation_amd.compat -- or, with install_compat=False, found by the import system itself as the packages
    spconv/ and torch_scatter/ at the repository root (no call into this package before the reference's own
    imports).  Nothing is written to the reference tree; sys.modules is restored afterwards."""
    from segmentation_amd import compat
    saved = dict(sys.modules)
    saved_meta = list(sys.meta_path)
    if install_compat:
        compat.install(force=True, gpu_voxelizer=gpu_voxelizer)
    else:
        for k in list(sys.modules):
            if k.split(".")[0] in ("spconv", "torch_scatter"):
                del sys.modules[k]
    jt = types.ModuleType("jaxtyping")

    class _Ann:
        def __class_getitem__(cls, item):
            return cls

    for n in ("Int", "Float", "Bool"):
        setattr(jt, n, _Ann)
    sys.modules["jaxtyping"] = jt
    ad = types.ModuleType("addict")

    class Dict(dict):
        __getattr__ = dict.get
        __setattr__ = dict.__setitem__

    ad.Dict = Dict
    sys.modules["addict"] = ad
    timm = types.ModuleType("timm")
    layers = types.ModuleType("timm.layers")
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    timm.layers = layers
    sys.modules["timm"], sys.modules["timm.layers"] = timm, layers
    sys.path.insert(0, REF)
    old_flag = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        from src.models.networks.spunet import spconv_unet_v1m3_pdnorm as v1m3
        from src.models.networks.spunet import spconv_unet_v1m1_base as v1m1
        from src.models.networks.ppt import model as ppt
        from src.models.utils import structure
        yield v1m3, v1m1, ppt, structure
    finally:
        sys.dont_write_bytecode = old_flag
        sys.path.remove(REF)
        sys.meta_path[:] = saved_meta
        for k in list(sys.modules):
            if k not in saved:
                del sys.modules[k]
            elif sys.modules[k] is not saved[k]:
                sys.modules[k] = saved[k]


class CpuConvs:
    """Runs a model built on segmentation_amd.sparse's modules on the CPU: every sparse convolution (and, for our
    own SpUNet, the voxelisation) goes through the oracle, in the dtype of the features."""

    def __enter__(self):
        import oracle
        from segmentation_amd import sparse, spunet
        self._sparse, self._spunet = sparse, spunet
        self._fwd, self._sparsify = sparse.SparseConvolution.forward, spunet.SpUNet.sparsify

        def sparsify(self_, input_dict, pad=128):
            sp = oracle.sparsify(input_dict["coord"].numpy(), input_dict["feat"].numpy(), np.asarray(input_dict["offset"]),
                                 input_dict["grid_size"], pad=pad)
            x = sparse.SparseConvTensor(torch.from_numpy(sp["voxel_feats"]).double(), torch.from_numpy(sp["voxel_coords"]),
                                        sp["spatial_shape"], sp["batch_size"])
            return x, dict(v2p_map=torch.from_numpy(sp["v2p_map"]))

        def forward(self_, x):
            w = self_.weight.to(x.features.dtype)
            b = None if self_.bias is None else self_.bias.to(x.features.dtype)
            if self_.conv1x1:
                return x.replace_feature(oracle.conv1x1(x.features, w, b))
            key = ("oracle", self_.indice_key)
            if self_.inverse:
                pf, pb, idx, shape = x.indice_dict[key]
                return sparse.SparseConvTensor(oracle.inverse_conv(x.features, w, pb, b), idx, shape, x.batch_size,
                                               indice_dict=x.indice_dict)
            if self_.subm:
                if key not in x.indice_dict:
                    x.indice_dict[key] = oracle.subm_rulebook(x.indices.numpy(), x.spatial_shape)
                return x.replace_feature(oracle.subm_conv(x.features, w, x.indice_dict[key], b))
            oi, oshape, pf, pb = oracle.strided_rulebook(x.indices.numpy(), x.spatial_shape)
            x.indice_dict[key] = (pf, pb, x.indices, x.spatial_shape)
            return sparse.SparseConvTensor(oracle.strided_conv(x.features, w, pf, b), torch.from_numpy(oi), oshape,
                                           x.batch_size, indice_dict=x.indice_dict)

        sparse.SparseConvolution.forward, spunet.SpUNet.sparsify = forward, sparsify
        return self

    def __exit__(self, *exc):
        self._sparse.SparseConvolution.forward, self._spunet.SpUNet.sparsify = self._fwd, self._sparsify


def have_reference():
    return os.path.isdir(REF)
