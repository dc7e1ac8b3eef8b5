"""The seeded batch of tests/golden/reference_loss.npz rebuilt from the stored inputs, in the form the loss takes it
(what mosaic3d/src/data/utils/collate.py:32-70,123-130 hands over)."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SETTINGS = {  # CaptionLoss settings of the generator (tests/golden/make_golden_loss.py): normalize, scaled, reduction
    "cfg": (True, True, "weighted_sum"),
    "plain": (True, False, "weighted_sum"),
    "mean": (True, True, "mean"),
    "raw": (False, False, "weighted_sum"),
}


def load():
    return np.load(os.path.join(HERE, "golden", "reference_loss.npz"))


def batch(g, device="cpu", dtype=torch.float64):
    lens = g["in_idx_len"]
    starts = np.concatenate([[0], np.cumsum(lens)])
    offset = g["in_offset"]
    local = [torch.from_numpy(g["in_idx_flat"][starts[i]:starts[i + 1]]) for i in range(len(lens))]
    scene = g["in_cap_scene"]
    point_indices = torch.cat([l + int(offset[s]) for l, s in zip(local, scene)])
    counts = torch.from_numpy(lens.astype(np.int64))
    caption_offsets = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)])
    pool = torch.from_numpy(g["in_pool"]).to(device=device, dtype=dtype)
    embeddings = [[pool[int(r)] for r, s in zip(g["in_cap_rows"], scene) if s == sc] for sc in range(int(scene.max()) + 1)]
    t = lambda k: torch.from_numpy(g[k]).to(device=device, dtype=dtype)
    return dict(pre=t("in_pre"), w=t("in_w"), b=t("in_b"), v2p=torch.from_numpy(g["in_v2p"]).to(device),
                embeddings=embeddings, point_indices=point_indices.to(device), caption_offsets=caption_offsets,
                counts=counts, local_indices=local, scene=scene, offset=offset)


def scale_of(g, name, device="cpu", dtype=torch.float64):
    return torch.tensor(float(g[f"{name}_scale"][0]), dtype=dtype, device=device, requires_grad=True)
