"""Generate golden vectors by running the reference's own pure numpy/torch functions.

Runs ONLY in the build container (needs /root/reference).  The reference modules
import four packages that are not installed here (jaxtyping, addict, torch_scatter,
spconv); annotation/container-only stand-ins are registered in sys.modules so that
the modules import.  Only outputs that do not depend on a stand-in's arithmetic are
recorded: hash keys, voxel order/coordinates, v2p_map, spatial shape, offset helpers,
ravel keys and GridSample's hash/inverse.  voxel_feats (torch_scatter) is NOT recorded.

Usage:  python -B tests/golden/make_golden.py
"""
import sys
import types
import os

import numpy as np
import torch

REF = "/root/reference/mosaic3d"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_standins():
    jt = types.ModuleType("jaxtyping")

    class _Ann:
        def __class_getitem__(cls, item):
            return cls

    for n in ("Int", "Float", "Bool", "Shaped"):
        setattr(jt, n, _Ann)
    sys.modules["jaxtyping"] = jt

    ad = types.ModuleType("addict")

    class Dict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    ad.Dict = Dict
    sys.modules["addict"] = ad

    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=0, out=None, reduce="sum"):
        n = int(index.max()) + 1
        res = torch.zeros((n,) + src.shape[1:], dtype=src.dtype)
        res.index_add_(0, index, src)
        if reduce == "mean":
            cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
            res = res / cnt.view(-1, *([1] * (src.dim() - 1)))
        return res

    ts.scatter = scatter
    ts.segment_csr = None
    sys.modules["torch_scatter"] = ts

    sp = types.ModuleType("spconv")
    spp = types.ModuleType("spconv.pytorch")

    class SparseConvTensor:
        def __init__(self, features, indices, spatial_shape, batch_size):
            self.features = features
            self.indices = indices
            self.spatial_shape = spatial_shape
            self.batch_size = batch_size

    class SparseConv3d(torch.nn.Module):
        pass

    spp.SparseConvTensor = SparseConvTensor
    spp.SparseConv3d = SparseConv3d
    sp.pytorch = spp
    sys.modules["spconv"] = sp
    sys.modules["spconv.pytorch"] = spp


def scene(rng, n, extent=(3.0, 2.5, 1.5), neg=True):
    pts = rng.uniform(0, 1, size=(n, 3)) * np.array(extent)
    if neg:
        pts -= np.array(extent) / 2.0
    # quantise a fraction so that several points share a voxel
    return pts.astype(np.float32)


def main():
    sys.dont_write_bytecode = True
    _install_standins()
    sys.path.insert(0, REF)
    from src.models.utils import structure as S
    from src.models.utils import misc as M

    rng = np.random.default_rng(20240607)
    out = {}

    # --- FNV known answers (structure.py:21-31)
    kat_in = np.array(
        [[0, 0, 0, 0], [0, 1, 2, 3], [1, 0, 0, 0], [3, 511, 1023, 77], [0, -1, -2, -3],
         [65535, 65535, 65535, 65535], [7, 2147483647, 0, 1]], dtype=np.int32)
    out["fnv_in"] = kat_in
    out["fnv_out"] = S.fnv_hash_vec(kat_in)
    kat3 = rng.integers(0, 700, size=(64, 3)).astype(np.int64)
    out["fnv3_in"] = kat3
    out["fnv3_out"] = S.fnv_hash_vec(kat3)

    # --- offset helpers (misc.py:8-26)
    for name, off in (("a", [0, 3, 5]), ("b", [4, 9, 9, 12]), ("c", [7])):
        t = torch.tensor(off, dtype=torch.int64)
        out[f"off_{name}_in"] = np.array(off, dtype=np.int64)
        out[f"off_{name}_batch"] = M.offset2batch(t).numpy()
        out[f"off_{name}_bincount"] = M.offset2bincount(t).numpy()
    out["b2o_in"] = np.array([0, 0, 0, 1, 1, 3], dtype=np.int64)
    out["b2o_out"] = M.batch2offset(torch.tensor([0, 0, 0, 1, 1, 3])).numpy()

    # --- ravel key (misc.py:60-68)
    rc = torch.from_numpy(rng.integers(0, 50, size=(40, 4)).astype(np.int32))
    out["ravel_in"] = rc.numpy()
    out["ravel_out"] = M.ravel_multi_index_auto_shape(rc).numpy()

    # --- Point.sparsify / mean_pooling (structure.py:34-66,197-251)
    cases = {
        "s1": dict(ns=[400], grid=0.05, ext=(1.5, 1.2, 0.8)),
        "s2": dict(ns=[300, 500], grid=0.05, ext=(1.5, 1.2, 0.8)),
        "s3": dict(ns=[2000, 1, 700], grid=0.02, ext=(0.9, 0.7, 0.5)),
    }
    for name, c in cases.items():
        coords = [scene(rng, n, c["ext"]) for n in c["ns"]]
        coord = np.concatenate(coords, 0)
        feat = rng.uniform(-1, 1, size=(coord.shape[0], 3)).astype(np.float32)
        offset = np.concatenate([[0], np.cumsum(c["ns"])]).astype(np.int64)
        for hm in ("fnv", "ravel"):
            p = S.Point(dict(coord=torch.from_numpy(coord), feat=torch.from_numpy(feat),
                             offset=torch.from_numpy(offset), grid_size=c["grid"]))
            p.sparsify(pad=128, hash_method=hm)
            sct = p.sparse_conv_feat
            pre = f"vox_{name}_{hm}_"
            out[pre + "voxel_coords"] = sct.indices.numpy().astype(np.int32)
            out[pre + "v2p_map"] = p.v2p_map.numpy().astype(np.int64)
            out[pre + "spatial_shape"] = np.array(sct.spatial_shape, dtype=np.int64)
            out[pre + "batch_size"] = np.array([sct.batch_size], dtype=np.int64)
            out[pre + "grid_coord"] = p.grid_coord.numpy().astype(np.int32)
        out[f"vox_{name}_coord"] = coord
        out[f"vox_{name}_feat"] = feat
        out[f"vox_{name}_offset"] = offset
        out[f"vox_{name}_grid"] = np.array([c["grid"]], dtype=np.float64)
        bc = torch.cat([M.offset2batch(torch.from_numpy(offset)).unsqueeze(-1).int(),
                        torch.from_numpy(out[f"vox_{name}_fnv_grid_coord"])], 1).numpy()
        out[f"vox_{name}_key"] = S.fnv_hash_vec(bc)

    np.savez_compressed(os.path.join(OUT, "reference_voxelize.npz"), **out)
    print("wrote", os.path.join(OUT, "reference_voxelize.npz"), len(out), "arrays")
    print("fnv KAT:", out["fnv_out"][:3])


if __name__ == "__main__":
    main()
