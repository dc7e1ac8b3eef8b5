"""Writers for the end of the inference path: the `pred_*.pth` record of
DenseLanguageLitModule.validation_step in infer-only mode
(mosaic3d/src/models/lightning_modules/language_module.py:346-359) and the labelled point cloud of
mosaic3d/save_final_ply.py:18-36 (which needs open3d; here a binary little-endian PLY is written directly)."""
import os
from typing import Optional, Sequence

import numpy as np
import torch


def save_pred(path: str, origin_coord, pc_count, pred, logits=None, color=None, class_names: Optional[Sequence[str]] = None):
    """Same keys as the reference's record: coord, color, pc_count, pred, logits, class_names (CPU tensors)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    cpu = lambda t: None if t is None else torch.as_tensor(t).detach().cpu()
    torch.save({"coord": cpu(origin_coord), "color": cpu(color), "pc_count": cpu(pc_count), "pred": cpu(pred),
                "logits": cpu(logits), "class_names": None if class_names is None else list(class_names)}, path)
    return path


def class_colormap(num_classes: int, seed: int = 0) -> np.ndarray:
    """One random colour per class (save_final_ply.py:23-25 draws them with np.random.rand; seeded here)."""
    return np.random.default_rng(seed).random((num_classes + 1, 3))


def write_label_ply(path: str, coord, pred, colormap: Optional[np.ndarray] = None, seed: int = 0):
    """Points coloured by predicted class, binary little-endian PLY (x y z float32, red green blue uchar)."""
    pts = torch.as_tensor(coord).detach().cpu().numpy().astype("<f4")
    lab = torch.as_tensor(pred).detach().cpu().numpy().astype(np.int64)
    if pts.ndim != 2 or pts.shape[1] != 3 or lab.shape != (pts.shape[0],):
        raise ValueError("write_label_ply: coord must be [N,3] and pred [N]")
    if colormap is None:
        colormap = class_colormap(int(lab.max()) + 1 if lab.size else 1, seed)
    rgb = np.clip(np.asarray(colormap)[lab] * 255.0, 0, 255).astype(np.uint8)
    rec = np.empty(pts.shape[0], dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    rec["r"], rec["g"], rec["b"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    header = ("ply\nformat binary_little_endian 1.0\n" f"element vertex {pts.shape[0]}\n"
              "property float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rec.tobytes())
    return path


def load_pred_for_ply(path: str):
    """What save_final_ply.py reads back (its torch.load, restricted to tensors and plain containers)."""
    data = torch.load(path, map_location="cpu", weights_only=True)
    return data["coord"].numpy(), data["pred"].numpy()
