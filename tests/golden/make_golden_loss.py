"""Golden vectors of the reference's OWN caption loss and open-vocabulary head.

Runs ONLY in the build container (needs /root/reference).  Imports, untouched,
    mosaic3d/src/models/losses/caption_loss.py        (CaptionLoss.loss :195-235, extract_text_features :25-49)
    mosaic3d/src/models/losses/clip_alignment_loss.py (CLIPAlignmentEval.forward/predict :126-147)
with annotation/plumbing-only stand-ins for the packages that are not installed (jaxtyping, omegaconf.DictConfig,
a logger in place of src.utils.RankedLogger) and a loop-based torch_scatter.segment_csr/scatter written here (a
segment mean has one definition; this package's own operators are NOT used, so the vectors do not depend on them).
`src.utils` is registered as a bare package so that its hydra/lightning __init__ does not run.

Two cases:
 * loss: a seeded batch of two scenes with duplicated caption embeddings (so that np.unique(axis=0) merges rows), a
   caption without points, per-voxel inputs `pre` of a 1x1 last layer (W, b), clip_feat = (pre @ W.T + b)[v2p_map]
   as language_module.py:205 forms it; CaptionLoss as configs/model/loss/caption.yaml builds it (weighted_sum,
   use_logit_scale) and three other settings.  Recorded: the inputs, the loss, its gradients with respect to pre,
   W, b and the logit scale, and the text-feature bookkeeping of extract_text_features.
 * head: the per-point features of the reference's own forward pass (make_golden_forward.run) through
   CLIPAlignmentEval.predict with normalize_input False and True: sampled rows of the logits, every label.

Usage:  python -B tests/golden/make_golden_loss.py
"""
import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/mosaic3d"

SETTINGS = {  # name -> CaptionLoss kwargs
    "cfg": dict(reduction="weighted_sum", use_logit_scale=True),       # configs/model/loss/caption.yaml:3-5
    "plain": dict(reduction="weighted_sum", use_logit_scale=False),
    "mean": dict(reduction="mean", use_logit_scale=True),
    "raw": dict(reduction="weighted_sum", use_logit_scale=False, normalize=False),
}
HEAD_CLASSES, HEAD_ROWS = 20, 96


def _install_standins():
    jt = types.ModuleType("jaxtyping")

    class _Ann:
        def __class_getitem__(cls, item):
            return cls

    for n in ("Int", "Float", "Bool", "Shaped"):
        setattr(jt, n, _Ann)
    sys.modules["jaxtyping"] = jt
    oc = types.ModuleType("omegaconf")
    oc.DictConfig = dict
    sys.modules["omegaconf"] = oc

    ts = types.ModuleType("torch_scatter")

    def segment_csr(src, indptr, reduce="sum"):
        rows = []
        for s in range(indptr.numel() - 1):
            seg = src[int(indptr[s]):int(indptr[s + 1])]
            if seg.shape[0] == 0:
                rows.append(torch.zeros(src.shape[1:], dtype=src.dtype))
            else:
                rows.append(seg.mean(0) if reduce == "mean" else seg.sum(0))
        return torch.stack(rows, 0)

    def scatter(src, index, dim=0, out=None, reduce="sum"):
        n = int(index.max()) + 1 if out is None else out.shape[0]
        res = torch.zeros((n,) + src.shape[1:], dtype=src.dtype)
        res.index_add_(0, index, src)
        if reduce == "mean":
            cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
            res = res / cnt.view(-1, *([1] * (src.dim() - 1)))
        return res

    ts.segment_csr, ts.scatter = segment_csr, scatter
    sys.modules["torch_scatter"] = ts

    pk = types.ModuleType("src.utils")
    pk.__path__ = [REF + "/src/utils"]
    pk.RankedLogger = lambda name=__name__, rank_zero_only=False, extra=None: logging.getLogger(name)
    sys.modules["src.utils"] = pk


def loss_inputs():
    """The seeded batch of the loss case (all float64)."""
    rng = np.random.default_rng(20241112)
    n_vox, n_pts, cin, d, pool_rows = 420, 900, 12, 48, 10
    pre = rng.normal(size=(n_vox, cin))
    w = rng.normal(size=(d, cin)) * 0.3
    b = rng.normal(size=(d,)) * 0.2
    v2p = rng.integers(0, n_vox, size=n_pts).astype(np.int64)
    pool = rng.normal(size=(pool_rows, d))
    pool /= np.linalg.norm(pool, axis=1, keepdims=True)
    offset = np.array([0, 500, 900], dtype=np.int64)           # two scenes
    caps_per_scene = [9, 7]
    cap_rows, cap_scene, idx = [], [], []
    for s, nc in enumerate(caps_per_scene):
        lo, hi = int(offset[s]), int(offset[s + 1])
        for c in range(nc):
            cap_rows.append(int(rng.integers(0, pool_rows)))
            cap_scene.append(s)
            k = 0 if (s == 1 and c == 3) else int(rng.integers(1, 160))   # one caption without points
            idx.append(np.sort(rng.choice(hi - lo, size=k, replace=False)).astype(np.int64))  # scene-local indices
    return dict(pre=pre, w=w, b=b, v2p=v2p, pool=pool, offset=offset, cap_rows=np.array(cap_rows, dtype=np.int64),
                cap_scene=np.array(cap_scene, dtype=np.int64), idx=idx)


def run_loss(CL):
    inp = loss_inputs()
    out = {f"in_{k}": v for k, v in inp.items() if k != "idx"}
    out["in_idx_flat"] = np.concatenate(inp["idx"])
    out["in_idx_len"] = np.array([len(i) for i in inp["idx"]], dtype=np.int64)
    # what the collate function hands to the loss (mosaic3d/src/data/utils/collate.py:32-70,123-130): scene-local
    # indices shifted by the scene's first point, flattened; offsets per caption; points per caption
    shifted = [torch.from_numpy(i) + int(inp["offset"][s]) for i, s in zip(inp["idx"], inp["cap_scene"])]
    point_indices = torch.cat(shifted)
    counts = torch.tensor([len(i) for i in inp["idx"]], dtype=torch.int64)
    caption_offsets = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)])
    # num_points_per_caption is the weight of a caption in the weighted sum: the collate passes the counts
    embeddings = [[torch.from_numpy(inp["pool"][r]) for r, s in zip(inp["cap_rows"], inp["cap_scene"]) if s == sc]
                  for sc in range(2)]
    for name, kw in SETTINGS.items():
        pre = torch.from_numpy(inp["pre"]).requires_grad_(True)
        w = torch.from_numpy(inp["w"]).requires_grad_(True)
        b = torch.from_numpy(inp["b"]).requires_grad_(True)
        mod = CL.CaptionLoss(**kw).double()
        feats = pre @ w.T + b
        clip_feat = feats[torch.from_numpy(inp["v2p"])]          # language_module.py:205
        loss = mod.loss(clip_feat, point_indices, caption_offsets, counts, clip_encoder=None, captions=None,
                        embeddings=embeddings)
        leaves = [pre, w, b] + ([mod.logit_scale] if kw.get("use_logit_scale") else [])
        grads = torch.autograd.grad(loss, leaves)
        out[f"{name}_loss"] = np.array([loss.item()])
        out[f"{name}_dpre"] = grads[0].numpy()
        out[f"{name}_dw"] = grads[1].numpy()
        out[f"{name}_db"] = grads[2].numpy()
        if kw.get("use_logit_scale"):
            out[f"{name}_dscale"] = np.array([grads[3].item()])
            out[f"{name}_scale"] = np.array([mod.logit_scale.item()])
    text, per_seg, per_cap = CL.CaptionLoss().extract_text_features(None, None, embeddings)
    out["text_features"] = text.numpy()
    out["labels_per_segment"] = np.asarray(per_seg.numpy()).reshape(-1).astype(np.int64)
    out["labels_per_caption"] = per_cap.numpy().astype(np.int64)
    return out


def run_head(CA, forward_run=None):
    import make_golden_forward as mgf
    from segmentation_amd import scenes
    rec, outs, point, params, batch, cfg = forward_run if forward_run is not None else mgf.run()
    clip_feat = point.sparse_conv_feat.features[point.v2p_map]   # language_module.py:202-207
    emb = torch.from_numpy(scenes.text_embeddings(HEAD_CLASSES, cfg["out_channels"])).double()
    out = {}
    rng = np.random.default_rng(1)
    rows = np.sort(rng.choice(clip_feat.shape[0], size=HEAD_ROWS, replace=False)).astype(np.int64)
    out["head_rows"] = rows
    for tag, norm in (("raw", False), ("norm", True)):
        ev = CA.CLIPAlignmentEval(normalize_input=norm)
        ev.set_target_embedding(emb)
        ev.emb_target = emb   # set_target_embedding casts to float32; the golden is taken in float64
        logit = ev.predict(clip_feat, return_logit=True)
        out[f"head_{tag}_logits"] = logit[torch.from_numpy(rows)].numpy()
        out[f"head_{tag}_pred"] = ev.predict(clip_feat).numpy().astype(np.int16)
        out[f"head_{tag}_max"] = logit.max(1)[1].numpy().astype(np.int16)   # language_module.py:328
        top2 = logit.topk(2, dim=1).values
        out[f"head_{tag}_margin"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32)
    out["head_n_points"] = np.array([clip_feat.shape[0]], dtype=np.int64)
    return out


def reference_loss_modules():
    """(caption_loss, clip_alignment_loss) imported untouched from /root/reference; sys.modules is restored
    afterwards (the modules stay usable through the returned objects)."""
    old_flag, sys.dont_write_bytecode = sys.dont_write_bytecode, True
    saved = dict(sys.modules)
    _install_standins()
    sys.path.insert(0, REF)
    try:
        from src.models.losses import caption_loss as CL
        from src.models.losses import clip_alignment_loss as CA
    finally:
        sys.path.remove(REF)
        sys.dont_write_bytecode = old_flag
        for k in list(sys.modules):
            if k not in saved:
                del sys.modules[k]
        for k, v in saved.items():
            sys.modules[k] = v
    return CL, CA


def main():
    CL, CA = reference_loss_modules()
    out = run_loss(CL)
    out.update(run_head(CA))
    path = os.path.join(HERE, "reference_loss.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: float(out[f"{k}_loss"][0]) for k in SETTINGS})


if __name__ == "__main__":
    main()
