"""Training step of train_spunet_multidata_ppt on the HIP kernels.

Mirrors DenseLanguageLitModule.training_step (mosaic3d/src/models/lightning_modules/language_module.py:209-277),
CaptionLoss.loss and CaptionLossBase.extract_text_features (mosaic3d/src/models/losses/caption_loss.py:195-235,
:25-49), the optimiser / scheduler of configs/optim/sgd.yaml:2-8 and configs/scheduler/onecyclelr.yaml:2-11 and the
data-parallel gradient exchange of configs/trainer/ddp.yaml (one process per GPU, `torch.distributed` over RCCL,
local batch-norm statistics).  The sparse convolutions run forward and backward on the kernels of the C ABI
(sparse._SparseConvFn); normalisation, activation and the loss are torch ops on the same device.
"""
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F



_ROW_HASH_WEIGHTS = {}


def _unique_rows_on_device(text: torch.Tensor):
    """np.unique(text, axis=0, return_index=True, return_inverse=True) for a [S, D] float32 tensor on the GPU.

    torch.unique(dim=0) sorts whole rows in one workgroup (8.5 ms for 1200 x 768).  Here equal rows are grouped by a
    64-bit hash of their bit patterns, the group representatives (first occurrences) are put into lexicographic
    order by stable sorts on their four leading columns, and the result is verified: every row must equal its
    representative and no two neighbouring representatives may tie on those columns.  A failed check (a hash
    collision, or rows that agree on four columns) falls back to torch.unique(dim=0)."""
    s, d = text.shape
    lead = min(4, d)
    key = (d, text.device)
    weights = _ROW_HASH_WEIGHTS.get(key)
    if weights is None:
        g = torch.Generator().manual_seed(0x5EED)
        weights = (torch.randint(-(1 << 62), 1 << 62, (d,), generator=g, dtype=torch.int64) | 1).to(text.device)
        _ROW_HASH_WEIGHTS[key] = weights
    bits = (text.detach().float() + 0.0).contiguous().view(torch.int32).to(torch.int64)  # -0.0 -> +0.0
    _, inv = torch.unique((bits * weights).sum(1), return_inverse=True)
    u = int(inv.max()) + 1
    first = torch.full((u,), s, dtype=torch.int64, device=text.device)
    first.scatter_reduce_(0, inv, torch.arange(s, device=text.device), reduce="amin")
    reps = text[first]
    order = torch.arange(u, device=text.device)
    for c in range(lead - 1, -1, -1):
        order = order[torch.sort(reps[order, c], stable=True).indices]
    reps, first = reps[order], first[order]
    rank = torch.empty_like(order)
    rank[order] = torch.arange(u, device=text.device)
    inverse = rank[inv]
    ok = (reps[inverse] == text).all()
    if u > 1:
        ok = ok & (reps[1:, :lead] != reps[:-1, :lead]).any(1).all()
    if not bool(ok):
        uniq, inverse = torch.unique(text.detach(), dim=0, return_inverse=True)
        first = torch.full((uniq.shape[0],), s, dtype=torch.int64, device=text.device)
        first.scatter_reduce_(0, inverse, torch.arange(s, device=text.device), reduce="amin")
    return first, inverse


def unique_text_features(embeddings: Sequence[Sequence[torch.Tensor]]):
    """extract_text_features with precomputed embeddings (caption_loss.py:30-39): unique rows of the stacked
    caption embeddings (np.unique along axis 0) -> (text_features [U,D], labels_per_segment [S], labels_per_caption [U])."""
    text = torch.stack([item for sub in embeddings for item in sub], 0)
    if text.is_cuda:  # same result (rows in lexicographic order, first occurrences) without leaving the device
        first, inverse = _unique_rows_on_device(text)
        return text[first], inverse, first
    device = text.device
    _, first, inverse = np.unique(text.detach().cpu().numpy(), return_index=True, return_inverse=True, axis=0)
    first = torch.from_numpy(first).to(device)
    return text[first], torch.from_numpy(np.asarray(inverse).reshape(-1)).to(device), first


def caption_loss(point_features, text_features, labels_per_segment, point_indices, caption_offsets,
                 num_points_per_caption, normalize: bool = True, logit_scale: Optional[torch.Tensor] = None,
                 reduction: str = "weighted_sum", row_map: Optional[torch.Tensor] = None):
    """CaptionLoss.loss after the text features are known (caption_loss.py:212-235).

    row_map: when the per-point features are a row gather of fewer distinct rows (clip_feat = features[v2p_map],
    language_module.py:205), pass the distinct rows as `point_features` and the gather index as `row_map`:
    normalisation, logits and log-softmax are row-wise and commute with the gather, so they run once per voxel
    instead of once per point, and the [N_points, D] tensor is never formed."""
    device = point_features.device
    if normalize:
        point_features = F.normalize(point_features, dim=-1)
    logits = point_features @ text_features.T.to(device)
    if logit_scale is not None:
        logits = logit_scale.exp() * logits
    scores = F.log_softmax(logits, dim=-1)
    # The reference gathers scores[point_indices] ([L, U]), averages it per caption and lets nll_loss pick the
    # caption's own column.  Only that column is ever read, so the same number comes from one scalar per
    # (point, caption) pair: loss_c = -mean_{p in c} scores[row(p), label_c].  The [L, U] tensor, its segment mean
    # and their backward passes are never formed.
    if reduction not in ("mean", "weighted_sum"):
        raise ValueError(f"unknown reduction {reduction!r}")
    rows = point_indices.to(device)
    if row_map is not None:
        rows = row_map[rows]
    offsets = caption_offsets.to(device)
    counts = offsets[1:] - offsets[:-1]
    n_seg = counts.numel()
    labels = labels_per_segment.to(device)
    seg = torch.repeat_interleave(torch.arange(n_seg, device=device), counts, output_size=rows.numel())
    picked = scores.reshape(-1).gather(0, rows * scores.shape[1] + labels[seg])
    # F.nll_loss leaves out targets equal to its ignore_index (-100); a caption without points has mean 0
    if reduction == "mean":
        coef = 1.0 / (counts.clamp(min=1).to(scores.dtype) * n_seg)
    else:
        w = num_points_per_caption.to(device)
        coef = w.to(scores.dtype) / (counts.clamp(min=1).to(scores.dtype) * w.sum())
    return -(picked * coef[seg]).sum()


def flatten_caption_indices(batched_indices: Sequence[Sequence[torch.Tensor]], batch_offsets: torch.Tensor):
    """convert_list_list_tensor_to_tensor (mosaic3d/src/data/utils/collate.py:32-70 call site :123-130): per-scene lists of
    per-caption point index tensors -> (point_indices [L] shifted by the scene's first point, caption_offsets [S+1],
    num_points_per_caption [S])."""
    idx, counts = [], []
    for b, scene in enumerate(batched_indices):
        for t in scene:
            idx.append(t.long() + int(batch_offsets[b]))
            counts.append(t.numel())
    counts = torch.tensor(counts, dtype=torch.int64)
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)])
    return torch.cat(idx), offsets, counts


def allreduce_gradients(params, dist=None, bucket_bytes: int = 64 << 20):
    """Average the gradients over the data-parallel ranks: parameters are packed into flat buckets (64 MB by
    default: few, large ring all-reduces suit the point-to-point xGMI links) and all-reduced; no-op without a
    process group."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    grads = [p.grad for p in params if p.grad is not None]
    buckets, cur, size = [], [], 0
    for g in grads:
        cur.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        dist.all_reduce(flat)
        flat /= world
        o = 0
        for g in b:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
    return len(buckets)


class TrainStep:
    """One optimisation step: forward (kernel maps + SpUNet in train mode), caption loss, backward, gradient
    all-reduce, SGD(nesterov) + OneCycleLR."""

    def __init__(self, model: torch.nn.Module, total_steps: int, lr: float = 5e-2, momentum: float = 0.9,
                 weight_decay: float = 1e-4, pct_start: float = 0.05, div_factor: float = 10.0,
                 final_div_factor: float = 1000.0, dist=None, use_logit_scale: bool = False):
        self.model = model
        self.dist = dist
        self.logit_scale = None
        params = list(model.parameters())
        if use_logit_scale:
            self.logit_scale = torch.nn.Parameter(torch.ones([], device=params[0].device) * float(np.log(1 / 0.07)))
            params.append(self.logit_scale)
        self.params = params
        self.opt = torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=True)
        self.sched = torch.optim.lr_scheduler.OneCycleLR(self.opt, max_lr=lr, total_steps=total_steps, pct_start=pct_start,
                                                         anneal_strategy="cos", div_factor=div_factor,
                                                         final_div_factor=final_div_factor)

    def loss(self, batch: Dict):
        out = self.model(batch)
        feats = out["sparse_conv_feat"].features if isinstance(out, dict) else out.sparse_conv_feat.features
        v2p = out["v2p_map"] if isinstance(out, dict) else out.v2p_map
        cd = batch["caption_data"]
        text, seg_labels, _ = unique_text_features(cd["embedding"])
        # clip_feat = feats[v2p] (language_module.py:205) is applied inside the loss as a row map
        return caption_loss(feats, text, seg_labels, cd["point_indices"], cd["caption_offsets"],
                            cd["num_points_per_caption"], logit_scale=self.logit_scale, row_map=v2p)

    def __call__(self, batch: Dict) -> float:
        self.model.train()
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss(batch)
        loss.backward()
        allreduce_gradients(self.params, self.dist)
        self.opt.step()
        self.sched.step()
        return float(loss.detach())
