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
    text, col_mask = _pad_text_rows(text_features.to(device))
    logits = point_features @ text.T
    return _loss_from_logits(logits, labels_per_segment, point_indices, caption_offsets, num_points_per_caption,
                             logit_scale, reduction, row_map, col_mask)


def _pad_text_rows(text):
    """On the GPU the number of distinct text rows U becomes the row length of the [N, U] logits and a GEMM dimension;
    an odd U (511 as readily as 512: it is whatever np.unique leaves) puts every row of the logits off the 16-byte
    grid and the vendor GEMM and softmax kernels on their slow paths (measured: 26 instead of 20 ms of backward at
    U = 511).  U is padded to a multiple of 32 with zero rows; their logits are set to -inf by the returned mask, so
    they add exp(-inf) = 0 to the softmax sums and receive no gradient: same loss, same gradients."""
    u = text.shape[0]
    if not text.is_cuda or u % 32 == 0:
        return text, None
    pad = (-u) % 32
    mask = torch.zeros(u + pad, dtype=text.dtype, device=text.device)
    mask[u:] = float("-inf")
    return torch.cat([text, text.new_zeros((pad, text.shape[1]))], 0), mask


class _TallMatmul(torch.autograd.Function):
    """x [N, C] @ m [C, U] for N in the hundreds of thousands and small C, U.  The gradient of m reduces over N;
    as one GEMM it has C*U/tile-size workgroups (a few dozen) and runs on a fraction of the chip, so it is formed as
    a batched product over row ranges and summed in range order (deterministic)."""

    RANGES = 64

    @staticmethod
    def forward(ctx, x, m):
        ctx.save_for_backward(x, m)
        return x @ m

    @staticmethod
    def backward(ctx, g):
        x, m = ctx.saved_tensors
        dx = dm = None
        if ctx.needs_input_grad[0]:
            dx = g @ m.T
        if ctx.needs_input_grad[1]:
            n, r = x.shape[0], _TallMatmul.RANGES
            if n < 64 * r:
                dm = x.T @ g
            else:
                main = n // r * r
                g = g.contiguous()
                dm = torch.bmm(x[:main].view(r, main // r, -1).transpose(1, 2), g[:main].view(r, main // r, -1)).sum(0)
                if main < n:
                    dm = dm + x[main:].T @ g[main:]
        return dx, dm


def caption_loss_folded(pre_features, weight, bias, text_features, labels_per_segment, point_indices, caption_offsets,
                        num_points_per_caption, normalize: bool = True, logit_scale: Optional[torch.Tensor] = None,
                        reduction: str = "weighted_sum", row_map: Optional[torch.Tensor] = None):
    """caption_loss of features = pre_features @ weight.T + bias (the network's last, 1x1x1 convolution:
    spconv_unet_v1m3_pdnorm.py:318-320) without forming them.  With C_in = 96 inputs, D = 768 outputs and U texts,
        features @ text.T      = pre @ (weight.T @ text.T) + bias @ text.T            [N, U] from a [C_in, U] matrix
        |features|^2 per row   = rowsum((pre @ G) * pre) + 2 pre @ (weight.T @ bias) + |bias|^2,  G = weight.T @ weight
    so the [N, D] tensor, its normalisation and the two [N, D] x [D, U] products of the loss (forward and backward)
    become [N, C_in]-sized work; autograd carries the gradient to weight and bias through the small matrices."""
    device = pre_features.device
    w = weight.reshape(weight.shape[0], -1)  # [D, C_in]
    text, col_mask = _pad_text_rows(text_features.to(device))
    pre_features = pre_features.contiguous()
    logits = _TallMatmul.apply(pre_features, w.T @ text.T)
    if bias is not None:
        logits = logits + bias @ text.T
    if normalize:
        quad = _TallMatmul.apply(pre_features, w.T @ w)
        if bias is not None:
            # the linear term rides on the quadratic one (a matrix-vector product over [N, C_in] as a call of its
            # own runs far below memory speed in the BLAS library)
            sq = ((quad + 2.0 * (bias @ w)) * pre_features).sum(1) + bias @ bias
        else:
            sq = (quad * pre_features).sum(1)
        # F.normalize: x / max(|x|, 1e-12)
        logits = logits * (1.0 / sq.clamp_min(0.0).sqrt().clamp_min(1e-12))[:, None]
    return _loss_from_logits(logits, labels_per_segment, point_indices, caption_offsets, num_points_per_caption,
                             logit_scale, reduction, row_map, col_mask)


def _loss_from_logits(logits, labels_per_segment, point_indices, caption_offsets, num_points_per_caption, logit_scale,
                      reduction, row_map, col_mask=None):
    device = logits.device
    if logit_scale is not None:
        logits = logit_scale.exp() * logits
    if col_mask is not None:   # padding columns of _pad_text_rows: out of the softmax
        logits = logits + col_mask
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


def mix3d_offset(offset):
    """The Mix3D step of the reference, literally: cat(offset[1:-1:2], offset[-1:]) (language_module.py:212-214 in the
    training step, collate.py:154-157 in the collate function).  On offsets WITHOUT a leading zero, [a, b, c, d]
    becomes [b, d]: neighbouring samples merged in pairs.  The reference applies the line to the offsets its own
    collate builds, which carry a leading zero (collate.py:96-102): [0, a, b, c, d] becomes [a, c, d], i.e. the samples
    [0, a), [a, c), [c, d) once offset2bincount has put the zero back (misc.py:8-13), and [0, a, b] becomes [a, b]
    (nothing merged).  This function keeps that behaviour for both conventions, so a batch gives the result the
    reference gives for it."""
    offset = torch.as_tensor(offset)
    return torch.cat([offset[1:-1:2], offset[-1:]], dim=0)


class GradientReducer:
    """Data-parallel gradient averaging (configs/trainer/ddp.yaml:4-10: one process per GPU, local batch-norm
    statistics), bucketed and overlapped with the backward pass.

    The parameters are laid out once, in reverse registration order (roughly the order in which backward finishes
    them), over flat float32 buckets of about `bucket_bytes`.  Every parameter carries a post-accumulate hook; when
    the last parameter of a bucket has its gradient, the bucket is packed and its all-reduce is started
    asynchronously (on RCCL that runs on the communicator's own stream, behind the kernels already enqueued, and
    overlaps the rest of backward; few, large ring all-reduces suit the point-to-point xGMI links).  finish() starts
    whatever is left, waits, and writes the averages back.

    Ranks may use different parameters in a step (with decoupled PDBatchNorm only the norms of the batch's own
    condition receive a gradient, and every rank draws its own dataset): the layout never depends on which gradients
    exist; a missing gradient contributes zeros, and a parameter receives the averaged gradient iff at least one rank
    had one (a flag per parameter travels in one small MAX all-reduce) -- otherwise its .grad stays None, as on a
    single GPU."""

    def __init__(self, params, dist, bucket_bytes: int = 32 << 20, overlap: bool = True, force: bool = False):
        """force: run the whole exchange (packing, hooks, asynchronous all-reduce, write-back) even in a process
        group of one rank, where it changes nothing -- the way to exercise the path on a single GPU."""
        self.dist = dist
        self.params = [p for p in params if p.requires_grad]
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist is not None and dist.is_initialized())
        self.overlap = overlap
        order = list(reversed(range(len(self.params))))
        self.bucket_of, self.slices, sizes = {}, {}, []
        size = 0
        for i in order:
            n = self.params[i].numel()
            if not sizes or (size > 0 and (size + n) * 4 > bucket_bytes):
                sizes.append(0)
                size = 0
            bidx = len(sizes) - 1
            self.bucket_of[i] = bidx
            self.slices[i] = (size, size + n)
            size += n
            sizes[bidx] = size
        dev = self.params[0].device if self.params else "cpu"
        self.flat = [torch.zeros(n, dtype=torch.float32, device=dev) for n in sizes]
        self.members = [[i for i in order if self.bucket_of[i] == b] for b in range(len(sizes))]
        self.flags = torch.zeros(len(self.params), dtype=torch.float32, device=dev)
        self._pending = self._works = self._launched = None
        self._next = 0
        self._handles = []
        if self.active and overlap:
            for i, p in enumerate(self.params):
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i):
        def hook(_p):
            if self._pending is None:   # not between begin() and finish(): gradients are left alone
                return
            self._pending[self.bucket_of[i]].discard(i)
            self._launch_ready()
        return hook

    def _launch_ready(self):
        # buckets go out strictly in index order, so that every rank issues the same sequence of collectives
        # whatever order its gradients arrive in
        while self._next < len(self.members) and not self._pending[self._next]:
            self._launch(self._next)
            self._next += 1

    def begin(self, unused=()):
        """Call before backward().  `unused`: parameters (objects) known not to receive a gradient in this step on
        this rank (e.g. the norms of the other dataset conditions): their buckets need not wait for them."""
        if not self.active:
            return
        self._pending = [set(m) for m in self.members]
        if unused:
            skip = {id(p) for p in unused}
            for i, p in enumerate(self.params):
                if id(p) in skip:
                    self._pending[self.bucket_of[i]].discard(i)
        self._launched = [False] * len(self.members)
        self._works = []
        self._next = 0

    def _launch(self, b):
        flat = self.flat[b]
        have = [i for i in self.members[b] if self.params[i].grad is not None]
        if len(have) != len(self.members[b]):
            flat.zero_()
        if have:
            views = [flat[self.slices[i][0]:self.slices[i][1]].view_as(self.params[i]) for i in have]
            torch._foreach_copy_(views, [self.params[i].grad for i in have])
        self._launched[b] = True
        self._works.append(self.dist.all_reduce(flat, async_op=True))

    def finish(self):
        """Call after backward(): averaged gradients are in place when it returns.  Returns the bucket count."""
        if not self.active:
            return 0
        if self._pending is None:
            self.begin()
        for b in range(self._next, len(self.members)):
            self._launch(b)
        self._next = len(self.members)
        self.flags.copy_(torch.tensor([0.0 if p.grad is None else 1.0 for p in self.params]))
        self.dist.all_reduce(self.flags, op=self.dist.ReduceOp.MAX)
        for w in self._works:
            w.wait()
        used = self.flags.tolist()   # one host read-back per step
        dst, src = [], []
        for i, p in enumerate(self.params):
            if used[i] == 0.0:
                continue
            if p.grad is None:
                p.grad = torch.empty_like(p)
            lo, hi = self.slices[i]
            dst.append(p.grad)
            src.append(self.flat[self.bucket_of[i]][lo:hi].view_as(p))
        for f in self.flat:
            f.div_(self.world)
        if dst:
            torch._foreach_copy_(dst, src)
        self._pending = None
        return len(self.members)

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def allreduce_gradients(params, dist=None, bucket_bytes: int = 64 << 20):
    """Average the gradients over the data-parallel ranks after backward has finished (no overlap): a
    GradientReducer used once.  No-op without a process group.  Returns the number of buckets."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    red = GradientReducer(list(params), dist, bucket_bytes, overlap=False)
    red.begin()
    return red.finish()


class TrainStep:
    """One optimisation step: forward (kernel maps + SpUNet in train mode), caption loss, backward, gradient
    all-reduce, SGD(nesterov) + OneCycleLR."""

    def __init__(self, model: torch.nn.Module, total_steps: int, lr: float = 5e-2, momentum: float = 0.9,
                 weight_decay: float = 1e-4, pct_start: float = 0.05, div_factor: float = 10.0,
                 final_div_factor: float = 1000.0, dist=None, use_logit_scale: bool = False, mix_prob: float = 0.0,
                 seed: int = 0, overlap_allreduce: bool = True, fold_head: bool = True, force_reducer: bool = False):
        self.model = model
        # fold_head: the loss is computed from the input of the backbone's last 1x1x1 convolution
        # (caption_loss_folded); needs a backbone with `final` and `skip_final` (spunet.SpUNet)
        backbone = getattr(model, "backbone", model)
        self._fold_backbone = backbone if fold_head and hasattr(backbone, "skip_final") and \
            getattr(getattr(backbone, "final", None), "conv1x1", False) else None
        self.dist = dist
        # language_module.py:212-214: with probability mix_prob neighbouring samples of the batch are merged in pairs
        self.mix_prob = mix_prob
        self._rng = np.random.default_rng(seed)
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
        self.reducer = None
        if dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force_reducer):
            self.reducer = GradientReducer(self.params, dist, overlap=overlap_allreduce, force=force_reducer)

    def _idle_params(self, batch):
        """Parameters of the per-dataset norms of the conditions this batch does not carry (PDBatchNorm picks
        bns[conditions.index(condition)], spconv_unet_v1m3_pdnorm.py:57-62): they get no gradient in this step."""
        conditions = getattr(self.model, "conditions", None)
        cond = batch.get("condition")
        if not conditions or not cond or cond[0] not in conditions:
            return ()
        ci = list(conditions).index(cond[0])
        cache = self.__dict__.setdefault("_idle_cache", {})
        if ci not in cache:
            keep = f".bns.{ci}."
            cache[ci] = [p for n, p in self.model.named_parameters() if ".bns." in n and keep not in n]
        return cache[ci]

    def loss(self, batch: Dict):
        cd = batch["caption_data"]
        bb = self._fold_backbone
        if bb is not None:
            bb.skip_final = True
            try:
                out = self.model(batch)
            finally:
                bb.skip_final = False
            text, seg_labels, _ = unique_text_features(cd["embedding"])
            return caption_loss_folded(out["sparse_conv_feat"].features, bb.final.weight, bb.final.bias, text,
                                       seg_labels, cd["point_indices"], cd["caption_offsets"],
                                       cd["num_points_per_caption"], logit_scale=self.logit_scale,
                                       row_map=out["v2p_map"])
        out = self.model(batch)
        feats = out["sparse_conv_feat"].features if isinstance(out, dict) else out.sparse_conv_feat.features
        v2p = out["v2p_map"] if isinstance(out, dict) else out.v2p_map
        text, seg_labels, _ = unique_text_features(cd["embedding"])
        # clip_feat = feats[v2p] (language_module.py:205) is applied inside the loss as a row map
        return caption_loss(feats, text, seg_labels, cd["point_indices"], cd["caption_offsets"],
                            cd["num_points_per_caption"], logit_scale=self.logit_scale, row_map=v2p)

    def __call__(self, batch: Dict) -> float:
        self.model.train()
        self.opt.zero_grad(set_to_none=True)
        if self.mix_prob > 0 and self._rng.random() < self.mix_prob:
            batch = dict(batch, offset=mix3d_offset(batch["offset"]))
        loss = self.loss(batch)
        if self.reducer is not None:
            self.reducer.begin(unused=self._idle_params(batch))
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.opt.step()
        self.sched.step()
        return float(loss.detach())
