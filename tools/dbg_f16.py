import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from oracle import spunet as ospunet
from segmentation_amd import scenes, ops
from segmentation_amd.engine import SpUNetEngine
dev = torch.device("cuda:0")
cfg = ospunet.SPUNET34C
P = ospunet.make_spunet_params(cfg, seed=0, trained_like=True)
b = scenes.make_batch([11, 12], target_voxels=9000)
sp = oracle.sparsify(b["coord"], b["feat"], b["offset"], 0.02)
ref, inter = ospunet.spunet_forward(P, cfg, torch.from_numpy(sp["voxel_feats"]), sp["voxel_coords"], sp["spatial_shape"], "Matterport3D", return_intermediates=True)
sd = {("backbone." + k if not k.startswith("embedding_table") else k): v for k, v in P.items()}
coord, feat = torch.from_numpy(b["coord"]).to(dev), torch.from_numpy(b["feat"]).to(dev)
for prec in ("f16x3", "bf16x3"):
    eng = SpUNetEngine(sd, cfg, "Matterport3D", dev, precision=prec)
    keep = {}
    feats, vox = eng.forward_features(coord, feat, b["offset"], 0.02, keep=keep)
    print(prec, "status", ops.f16_status())
    for n, r in inter.items():
        k = keep[n].double().cpu()
        e = ((k - r.double()).norm(dim=1) / r.double().norm(dim=1).clamp_min(1e-20))
        print(f"  {n:10s} rows {k.shape[0]:6d} ch {k.shape[1]:4d} max err {e.max():.3e} |k| {k.abs().max():.3e} |ref| {r.abs().max():.3e} nan {int(torch.isnan(k).sum())}")
    print("  down.3 weight max", float(sd["backbone.down.3.conv.weight"].abs().max()), "exp?")
