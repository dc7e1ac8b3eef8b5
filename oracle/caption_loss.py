"""CPU restatement of the training loss of train_spunet_multidata_ppt.  Test infrastructure only.

Follows /root/reference/mosaic3d/src/models/losses/caption_loss.py:25-49 (extract_text_features with precomputed
`embedding` rows: stack, np.unique along axis 0 with first-occurrence indices and inverse) and :195-235
(CaptionLoss.loss: normalise, logits against the unique text rows, optional exp(logit_scale), log-softmax, gather of
the rows named by point_indices, mean per caption, NLL of the caption's own text row, point-count-weighted mean),
on clip_feat = features[v2p_map] as /root/reference/mosaic3d/src/models/lightning_modules/language_module.py:202-207
forms it.  Pinned: tests/golden/reference_loss.npz holds the outputs of the reference's own classes on a seeded batch
(tests/golden/make_golden_loss.py); tests/test_oracle_loss_golden.py checks this file against them.
"""
import numpy as np
import torch
import torch.nn.functional as F


def extract_text_features(embeddings):
    """caption_loss.py:31-39.  embeddings: list (scenes) of lists (captions) of [D] tensors.
    Returns (text_features [U, D], labels_per_segment [S], labels_per_caption [U])."""
    text = torch.stack([item for sub in embeddings for item in sub], 0)
    _, per_caption, per_segment = np.unique(text.detach().cpu().numpy(), return_index=True, return_inverse=True, axis=0)
    per_caption = torch.from_numpy(per_caption)
    return text[per_caption], torch.from_numpy(np.asarray(per_segment).reshape(-1)), per_caption


def segment_mean(rows, offsets):
    """torch_scatter.segment_csr(rows, offsets, reduce="mean"): the mean of rows[offsets[s]:offsets[s+1]] per
    segment, zero for an empty one (caption_loss.py:223)."""
    out = []
    for s in range(offsets.numel() - 1):
        seg = rows[int(offsets[s]):int(offsets[s + 1])]
        out.append(seg.mean(0) if seg.shape[0] else torch.zeros(rows.shape[1:], dtype=rows.dtype))
    return torch.stack(out, 0)


def caption_loss(point_features, embeddings, point_indices, caption_offsets, num_points_per_caption,
                 normalize=True, logit_scale=None, reduction="weighted_sum"):
    """CaptionLoss.loss (caption_loss.py:195-235).  logit_scale: None (use_logit_scale False) or the scalar whose
    exponential multiplies the logits (:190-191,218-219)."""
    text, per_segment, _ = extract_text_features(embeddings)
    if normalize:
        point_features = F.normalize(point_features, dim=-1)
    logits = point_features @ text.T.to(point_features.dtype)
    if logit_scale is not None:
        logits = logit_scale.exp() * logits
    scores = F.log_softmax(logits, dim=-1)
    reduced = segment_mean(scores[point_indices], caption_offsets)
    loss = F.nll_loss(reduced, per_segment, reduction="none")
    if reduction == "mean":
        return loss.mean()
    if reduction == "weighted_sum":
        w = num_points_per_caption.to(loss.dtype)
        return (loss * w).sum() / w.sum()
    raise ValueError(f"Unknown reduce type: {reduction}")


def training_loss(voxel_features, v2p_map, embeddings, point_indices, caption_offsets, num_points_per_caption, **kw):
    """The loss of a training step from per-voxel features: clip_feat = features[v2p_map]
    (language_module.py:205,234-237), then caption_loss."""
    return caption_loss(voxel_features[torch.as_tensor(v2p_map, dtype=torch.int64)], embeddings, point_indices,
                        caption_offsets, num_points_per_caption, **kw)
