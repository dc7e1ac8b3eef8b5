import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import oracle
from oracle import spunet as ospunet
from segmentation_amd import scenes
from segmentation_amd.engine import SpUNetEngine
dev = torch.device("cuda:0")
cfg = ospunet.SPUNET34C
for trained in (True, False):
    P = ospunet.make_spunet_params(cfg, seed=0, trained_like=trained)
    b = scenes.make_batch([11, 12], target_voxels=9000)
    sp = oracle.sparsify(b["coord"], b["feat"], b["offset"], 0.02)
    ref, inter = ospunet.spunet_forward(P, cfg, torch.from_numpy(sp["voxel_feats"]), sp["voxel_coords"], sp["spatial_shape"], "Matterport3D", return_intermediates=True)
    emb = torch.from_numpy(scenes.text_embeddings(200, cfg["out_channels"]))
    ref_pred = oracle.head_predict(ref, sp["v2p_map"], emb)
    sd = {("backbone." + k if not k.startswith("embedding_table") else k): v for k, v in P.items()}
    coord, feat = torch.from_numpy(b["coord"]).to(dev), torch.from_numpy(b["feat"]).to(dev)
    for prec in ("bf16", "bf16x3", "bf16x6", "f32"):
        eng = SpUNetEngine(sd, cfg, "Matterport3D", dev, precision=prec)
        eng.set_text_embeddings(emb)
        keep = {}
        feats, vox = eng.forward_features(coord, feat, b["offset"], 0.02, keep=keep)
        d = ((feats.double().cpu() - ref.double()).norm(dim=1) / ref.double().norm(dim=1))
        out = eng.predict(coord, feat, b["offset"], 0.02)
        agree = (out["pred"].cpu() == ref_pred).float().mean().item()
        worst = max(((keep[n].double().cpu() - r.double()).norm(dim=1) / r.double().norm(dim=1).clamp_min(1e-20)).max().item() for n, r in inter.items())
        print(f"trained_like={trained} {prec:7s} final max {d.max():.3e} mean {d.mean():.3e} worst layer {worst:.3e} labels {agree:.5f}", flush=True)
