"""Golden vectors of the reference's OWN forward code (spconv_unet_v1m3_pdnorm.SpUNetBase + ppt.model.PPT, imported
untouched from /root/reference) for the GPU box, where the reference tree does not exist.

    python tests/golden/make_golden_forward.py        # writes tests/golden/reference_forward_v1m3.npz

The reference's model runs on segmentation_amd.sparse's classes (spconv.pytorch resolved by segmentation_amd.compat);
Point.sparsify is the reference's own numpy voxelizer; every sparse convolution is evaluated by the CPU oracle in
float64 (the spconv wheel is absent everywhere).  Recorded per hooked module: 48 sampled rows of its output features
and the norm of the whole output.  Inputs are not stored: the scene (scenes.make_batch seeds) and the parameters
(oracle.make_spunet_params seed) are regenerated from their seeds by the test."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEEDS, TARGET, PARAM_SEED, CONDITION, ROWS = [21, 22], 5000, 3, "Structured3D", 48


def hooked_outputs(model):
    """Forward hooks on the U-Net stages of a reference SpUNetBase: name -> features tensor."""
    out, handles = {}, []

    def grab(name):
        def hook(_m, _i, o):
            o = o[0] if isinstance(o, tuple) else o
            out[name] = o.features.detach()
        return hook

    bb = model.backbone
    handles.append(bb.conv_input.register_forward_hook(grab("conv_input")))
    for s in range(bb.num_stages):
        handles.append(bb.down[s].register_forward_hook(grab(f"down.{s}")))
        handles.append(bb.enc[s].register_forward_hook(grab(f"enc.{s}")))
        handles.append(bb.up[s].register_forward_hook(grab(f"up.{s}")))
        handles.append(bb.dec[s].register_forward_hook(grab(f"dec.{s}")))
    handles.append(bb.final.register_forward_hook(grab("final")))
    return out, handles


def build_reference_ppt(v1m3, ppt, cfg, params):
    from functools import partial
    backbone = partial(v1m3.SpUNetBase, in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                       base_channels=cfg["base_channels"], channels=cfg["channels"], layers=cfg["layers"],
                       conditions=cfg["conditions"], norm_decouple=True, norm_adaptive=True, norm_affine=True,
                       zero_init=True)
    model = ppt.PPT(backbone, list(cfg["conditions"]), cfg["context_channels"])
    sd = {(k if k.startswith("embedding_table") else "backbone." + k): v for k, v in params.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.endswith("num_batches_tracked") for m in missing), (missing, unexpected)
    return model.double().eval()


def run(install_compat=True):
    import _refmodel
    from oracle import spunet as ospunet
    from segmentation_amd import scenes
    cfg = ospunet.SPUNET34C
    params = ospunet.make_spunet_params(cfg, seed=PARAM_SEED, trained_like=True)
    b = scenes.make_batch(SEEDS, target_voxels=TARGET)
    with _refmodel.reference_modules(install_compat=install_compat) as (v1m3, v1m1, ppt, structure):
        model = build_reference_ppt(v1m3, ppt, cfg, params)
        outs, handles = hooked_outputs(model)
        batch = dict(coord=torch.from_numpy(b["coord"]), feat=torch.from_numpy(b["feat"]).double(),
                     offset=torch.from_numpy(b["offset"]), grid_size=0.02, condition=[CONDITION])
        with torch.no_grad(), _refmodel.CpuConvs():
            point = model(batch)
        for h in handles:
            h.remove()
        rec = dict(voxel_coords=point.sparse_conv_feat.indices.numpy().astype(np.int32),
                   v2p_map=point.v2p_map.numpy().astype(np.int64),
                   sparse_shape=np.asarray(point.sparse_shape, dtype=np.int64))
        rng = np.random.default_rng(0)
        for name, f in outs.items():
            rows = np.sort(rng.choice(f.shape[0], size=min(ROWS, f.shape[0]), replace=False))
            rec[f"rows:{name}"] = rows.astype(np.int64)
            rec[f"vals:{name}"] = f[rows].numpy().astype(np.float64)
            rec[f"norm:{name}"] = np.array([float(f.norm()), f.shape[0], f.shape[1]])
        return rec, outs, point, params, b, cfg


if __name__ == "__main__":
    rec, *_ = run()
    path = os.path.join(ROOT, "tests", "golden", "reference_forward_v1m3.npz")
    np.savez_compressed(path, **rec)
    print(path, os.path.getsize(path), "bytes;", len([k for k in rec if k.startswith("rows:")]), "layers")
