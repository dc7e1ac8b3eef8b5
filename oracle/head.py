"""CPU restatement of the open-vocabulary head.  Test infrastructure only.

Follows /root/reference/mosaic3d/src/models/lightning_modules/language_module.py:202-207
(clip_feat = features[v2p_map]), /root/reference/mosaic3d/src/models/losses/clip_alignment_loss.py:126-147
(optional L2 normalise, feat @ emb_target.T) and language_module.py:328 (logits.max(1)[1]).
"""
import torch


def head_logits(voxel_feats, v2p_map, emb_target, normalize_input=False):
    clip_feat = voxel_feats[torch.as_tensor(v2p_map, dtype=torch.int64)]
    if normalize_input:
        clip_feat = torch.nn.functional.normalize(clip_feat, p=2, dim=1)
    return clip_feat @ emb_target.float().t()


def head_predict(voxel_feats, v2p_map, emb_target, normalize_input=False):
    return head_logits(voxel_feats, v2p_map, emb_target, normalize_input).max(1)[1]
