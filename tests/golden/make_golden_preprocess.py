"""Golden vectors for the steps in front of the voxelizer, from the reference's own transform / collate code.

Runs ONLY in the build container (needs /root/reference).  Uses the stand-ins of make_golden.py for the absent
packages and registers `src.utils` as a bare package so that its __init__ (hydra, lightning) does not run; only
mosaic3d/src/data/utils/{transform,collate}.py and src/utils/{registry,misc}.py execute.

Usage:  python -B tests/golden/make_golden_preprocess.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402


def main():
    sys.dont_write_bytecode = True
    mg._install_standins()
    sys.path.insert(0, mg.REF)
    pk = types.ModuleType("src.utils")
    pk.__path__ = [mg.REF + "/src/utils"]
    sys.modules["src.utils"] = pk
    from src.data.utils import transform as T
    from src.data.utils import collate as C

    rng = np.random.default_rng(20240611)
    out = {}
    clouds = {"a": (700, (2.0, 1.6, 1.1)), "b": (1500, (0.9, 1.3, 0.4)), "c": (1, (1.0, 1.0, 1.0))}
    samples = []
    for name, (n, ext) in clouds.items():
        coord = (rng.uniform(-0.5, 0.5, size=(n, 3)) * np.array(ext) + rng.uniform(-3, 3, size=3)).astype(np.float32)
        color = rng.integers(0, 256, size=(n, 3)).astype(np.uint8)
        out[f"{name}_coord"], out[f"{name}_color"] = coord, color
        # the val pipeline of configs/data/local/stray_refined_only.yaml:12-24
        d = dict(coord=coord.copy(), color=color.copy())
        d = T.Copy(keys_dict=dict(coord="origin_coord"))(d)
        d = T.CenterShift(apply_z=True)(d)
        d = T.NormalizeColor()(d)
        d = T.Add(keys_dict=dict(condition="ARKitScenes"))(d)
        d = T.ToTensor()(d)
        d = T.Collect(keys=["origin_coord", "coord", "color", "condition"],
                      offset_keys_dict=dict(offset="coord", pc_count="origin_coord"), feat_keys=["color"])(d)
        out[f"{name}_val_coord"] = d["coord"].numpy()
        out[f"{name}_val_feat"] = d["feat"].numpy()
        out[f"{name}_val_offset"] = d["offset"].numpy()
        samples.append(d)
        nz = T.CenterShift(apply_z=False)(dict(coord=coord.copy()))
        out[f"{name}_shift_noz"] = nz["coord"]
        # GridSample, deterministic outputs
        for hash_type in ("fnv", "ravel"):
            for g in (0.05, 0.02):
                gs = T.GridSample(grid_size=g, hash_type=hash_type, mode="train", keys=("coord", "color"),
                                  return_inverse=True, return_grid_coord=True, return_min_coord=True)
                np.random.seed(0)
                r = gs(dict(coord=coord.copy(), color=color.copy()))
                pre = f"{name}_gs_{hash_type}_{int(g * 100)}_"
                out[pre + "inverse"] = r["inverse"].astype(np.int64)
                out[pre + "voxel_grid_coord"] = r["grid_coord"].astype(np.int32)
                out[pre + "min_coord"] = r["min_coord"]
                gt = T.GridSample(grid_size=g, hash_type=hash_type, mode="test", keys=("coord", "color"),
                                  return_grid_coord=True)
                parts = gt(dict(coord=coord.copy(), color=color.copy()))
                out[pre + "n_parts"] = np.array([len(parts)], dtype=np.int64)
        grid = np.floor(coord / np.array(0.05)).astype(int)
        grid -= grid.min(0)
        out[f"{name}_hash_in"] = grid.astype(np.int64)
        out[f"{name}_fnv3"] = T.GridSample.fnv_hash_vec(grid)
        out[f"{name}_ravel3"] = T.GridSample.ravel_hash_vec(grid)
    batch = C.point_collate_fn_with_masks(samples, grid_size=0.02)
    out["batch_coord"] = batch["coord"].numpy()
    out["batch_feat"] = batch["feat"].numpy()
    out["batch_offset"] = batch["offset"].numpy().astype(np.int64)
    out["batch_pc_count"] = batch["pc_count"].numpy().astype(np.int64)
    out["batch_grid_size"] = np.array([batch["grid_size"]])
    np.savez_compressed(os.path.join(mg.OUT, "reference_preprocess.npz"), **out)
    print("wrote reference_preprocess.npz", len(out), "arrays; batch offset", out["batch_offset"], "condition", batch["condition"])


if __name__ == "__main__":
    main()
