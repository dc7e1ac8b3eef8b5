"""SpUNet (v1m3 PDNorm) + PPT wrapper built on segmentation_amd.sparse, with the reference's
parameter names, so that a Mosaic3D checkpoint's `net.*` state dict loads unchanged.

Layer list, channel plan and forward order follow
mosaic3d/src/models/networks/spunet/spconv_unet_v1m3_pdnorm.py:224-414 and
mosaic3d/src/models/networks/ppt/model.py:7-22; voxelisation follows Point.sparsify
(mosaic3d/src/models/utils/structure.py:197-251) but runs on the GPU (ops.voxelize).

This module is the layer-by-layer (unfused) path; segmentation_amd.engine runs the same network
with folded norms and fused epilogues."""
import os
from collections import OrderedDict
from typing import Dict, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, sparse

CONDITIONS = ("ScanNet", "ARKitScenes", "Matterport3D", "Structured3D", "ScanNetPP")
SPUNET34C = dict(in_channels=3, out_channels=768, base_channels=32, context_channels=256,
                 channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2),
                 conditions=CONDITIONS)
SPUNET18A = dict(SPUNET34C, channels=(32, 64, 128, 256, 128, 128, 96, 96), layers=(2, 2, 2, 2, 2, 2, 2, 2))


class PDBatchNorm(nn.Module):
    """Per-dataset BatchNorm1d + context modulation (v1m3:19-68)."""

    def __init__(self, num_features, context_channels=256, eps=1e-3, momentum=0.01,
                 conditions=CONDITIONS, decouple=True, adaptive=False, affine=True):
        super().__init__()
        self.conditions = tuple(conditions)
        self.decouple, self.adaptive, self.affine = decouple, adaptive, affine
        mk = lambda: nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine)
        if decouple:
            self.bns = nn.ModuleList([mk() for _ in self.conditions])
        else:
            self.bn = mk()
        if adaptive:
            self.modulation = nn.Sequential(nn.SiLU(), nn.Linear(context_channels, 2 * num_features, bias=True))

    def pick(self, condition):
        if not self.decouple:
            return self.bn
        if condition not in self.conditions:
            raise AssertionError(f"unknown condition {condition!r}")
        return self.bns[self.conditions.index(condition)]

    def forward(self, feat, condition=None, context=None, residual=None, relu=False):
        """The reference's forward, plus what its callers apply to the result (v1m3:121-140): `residual` is added
        and a ReLU applied when asked for, inside the normalisation's own elementwise pass where that runs on the
        library's kernels."""
        bn = self.pick(condition)
        if not self.adaptive:
            return _batch_norm_with(bn, feat, bn.weight, bn.bias, residual, relu)
        pre = self.__dict__.pop("_affine", None)  # set for one call by SpUNet.forward (all norms in one product)
        if pre is not None and feat.dim() == 2:
            return _batch_norm_with(bn, feat, pre[0], pre[1], residual, relu)
        assert context is not None
        shift, scale = self.modulation(context).chunk(2, dim=1)
        if bn.affine and shift.shape[0] == 1 and feat.dim() == 2:
            # One context row for the whole batch (ppt/model.py:15-22): bn(x)*(1+scale)+shift is the same batch
            # norm with weight gamma*(1+scale) and bias beta*(1+scale)+shift, so the two [N, C] elementwise passes
            # (and the three of their backward) become operations on [C] vectors.
            gain = 1.0 + scale[0]
            return _batch_norm_with(bn, feat, bn.weight * gain, bn.bias * gain + shift[0], residual, relu)
        return _finish(bn(feat) * (1.0 + scale) + shift, residual, relu)


def _finish(y, residual, relu):
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


_BN_KERNEL_ROWS = int(os.environ.get("M3D_BN_KERNEL_ROWS", "20000"))
_COUNTS = {}


class _BatchNormTrainFn(torch.autograd.Function):
    """y = relu?(batch-statistics normalisation of x [n, c] + residual) on the library's kernels: the column
    reductions (m3d_bn_stats, m3d_bn_act_backward_reduce) and one elementwise pass each way (m3d_bn_act_fwd,
    m3d_bn_act_backward); the ReLU's mask is read back from the saved output.  Returns (y, mean, biased
    variance); only y is differentiable."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, residual, relu):
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        mean, var = ops.bn_stats(x)
        y = ops.bn_act_fwd(x, mean, var, eps, weight, bias, residual, relu)
        ctx.save_for_backward(x, weight, mean, var, y if relu else None)
        ctx.eps, ctx.has_weight, ctx.has_bias = eps, weight is not None, bias is not None
        ctx.want_dres = residual is not None and ctx.needs_input_grad[4]
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        x, weight, mean, var, y = ctx.saved_tensors
        dx, dres, sum_g, sum_g_xmu = ops.bn_act_backward(x, dy.contiguous(), y, mean, var, ctx.eps, weight,
                                                         ctx.want_dres)
        dweight = sum_g_xmu * torch.rsqrt(var + ctx.eps) if ctx.has_weight else None
        return dx, dweight, sum_g if ctx.has_bias else None, None, dres, None


def _batch_norm_with(bn, x, weight, bias, residual=None, relu=False):
    """nn.BatchNorm1d.forward of `bn` with another affine pair: same statistics, running-average update and
    batch counter; then + residual and ReLU if asked for."""
    factor = 0.0 if bn.momentum is None else bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            factor = 1.0 / float(bn.num_batches_tracked)
    use_batch_stats = bn.training or (bn.running_mean is None and bn.running_var is None)
    keep_running = not bn.training or bn.track_running_stats
    n = x.shape[0]
    if (use_batch_stats and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] % 4 == 0
            and x.shape[1] <= 1024 and _BN_KERNEL_ROWS and n >= _BN_KERNEL_ROWS):
        y, mean, var = _BatchNormTrainFn.apply(x, weight, bias, bn.eps, residual, relu)
        if keep_running and bn.running_mean is not None:
            with torch.no_grad():
                bn.running_mean.lerp_(mean, factor)
                bn.running_var.lerp_(var * (n / (n - 1.0)), factor)
        return y
    y = F.batch_norm(x, bn.running_mean if keep_running else None, bn.running_var if keep_running else None,
                     weight, bias, use_batch_stats, factor, bn.eps)
    return _finish(y, residual, relu)


class ConvNormAct(nn.Module):
    """conv -> PDBatchNorm -> ReLU; the shape of conv_input / down[s] / up[s] (v1m3:143-221)."""

    def __init__(self, conv, norm):
        super().__init__()
        self.conv = conv
        self.bn = norm
        self.relu = nn.ReLU()

    def forward(self, x):
        x, condition, context = x
        out = self.conv(x)
        return out.replace_feature(self.bn(out.features, condition, context, relu=True))


class BasicBlock(sparse.SparseModule):
    """Residual block of two SubM k3 convolutions (v1m3:71-140)."""

    def __init__(self, in_channels, embed_channels, norm_fn, indice_key):
        super().__init__()
        self.in_channels, self.embed_channels = in_channels, embed_channels
        if in_channels != embed_channels:
            self.proj_conv = sparse.SparseConv3d(in_channels, embed_channels, kernel_size=1, bias=False)
            self.proj_norm = norm_fn(embed_channels)
        self.conv1 = sparse.SubMConv3d(in_channels, embed_channels, kernel_size=3, padding=1, bias=False,
                                       indice_key=indice_key)
        self.bn1 = norm_fn(embed_channels)
        self.relu = nn.ReLU()
        self.conv2 = sparse.SubMConv3d(embed_channels, embed_channels, kernel_size=3, padding=1, bias=False,
                                       indice_key=indice_key)
        self.bn2 = norm_fn(embed_channels)

    def forward(self, x):
        x, condition, context = x
        out = self.conv1(x)
        out = out.replace_feature(self.bn1(out.features, condition, context, relu=True))
        out = self.conv2(out)
        if self.in_channels == self.embed_channels:
            res = x.features
        else:
            res = self.proj_norm(self.proj_conv(x).features, condition, context)
        return out.replace_feature(self.bn2(out.features, condition, context, residual=res, relu=True)), \
            condition, context


class SpUNet(nn.Module):
    def __init__(self, in_channels=3, out_channels=768, num_classes=0, base_channels=32, context_channels=256,
                 channels=SPUNET34C["channels"], layers=SPUNET34C["layers"], conditions=CONDITIONS,
                 zero_init=True, norm_decouple=True, norm_adaptive=True, norm_affine=True,
                 hash_method="fnv", pooling_method="mean", pooling_seed=0):
        super().__init__()
        # voxelizer options of spconv_unet_v1m1_base.SpUNetBase (v1m1:177-178, :335-339); v1m3 always runs the defaults
        self.hash_method, self.pooling_method, self.pooling_seed = hash_method, pooling_method, pooling_seed
        # True: forward stops in front of `final` and returns its input as sparse_conv_feat (for callers that fold
        # the last, 1x1x1 convolution into what follows: train.caption_loss_folded)
        self.skip_final = False
        assert len(layers) % 2 == 0 and len(layers) == len(channels)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.base_channels, self.channels, self.layers = base_channels, tuple(channels), tuple(layers)
        self.num_stages = len(layers) // 2
        self.conditions = tuple(conditions)
        self.zero_init = zero_init
        norm = lambda c: PDBatchNorm(c, context_channels, 1e-3, 0.01, self.conditions, norm_decouple,
                                     norm_adaptive, norm_affine)
        ch = self.channels
        self.conv_input = ConvNormAct(
            sparse.SubMConv3d(in_channels, base_channels, kernel_size=3, padding=1, bias=False, indice_key="stem"),
            norm(base_channels))
        self.down, self.up, self.enc, self.dec = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        enc_c, dec_c = base_channels, ch[-1]
        for s in range(self.num_stages):
            self.down.append(ConvNormAct(
                sparse.SparseConv3d(enc_c, ch[s], kernel_size=2, stride=2, bias=False, indice_key=f"spconv{s + 1}"),
                norm(ch[s])))
            self.enc.append(sparse.SparseSequential(OrderedDict(
                (f"block{i}", BasicBlock(ch[s], ch[s], norm, f"subm{s + 1}")) for i in range(layers[s]))))
            self.up.append(ConvNormAct(
                sparse.SparseInverseConv3d(ch[len(ch) - s - 2], dec_c, kernel_size=2, bias=False,
                                           indice_key=f"spconv{s + 1}"),
                norm(dec_c)))
            self.dec.append(sparse.SparseSequential(OrderedDict(
                (f"block{i}", BasicBlock(dec_c + enc_c if i == 0 else dec_c, dec_c, norm, f"subm{s}"))
                for i in range(layers[len(ch) - s - 1]))))
            enc_c, dec_c = ch[s], ch[len(ch) - s - 2]
        final_out = num_classes if num_classes > 0 else out_channels
        self.final = sparse.SparseConv3d(ch[-1], final_out, kernel_size=1, bias=True)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        """v1m3:362-378."""
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, sparse.SubMConv3d):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm1d):
            if m.affine:
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, PDBatchNorm):
            if self.zero_init and m.adaptive:
                nn.init.constant_(m.modulation[-1].weight, 0)
                nn.init.constant_(m.modulation[-1].bias, 0)

    def sparsify(self, input_dict, pad=128):
        """Point.sparsify (structure.py:197-251).  A supplied "grid_coord" (GridSample(return_grid_coord=True)) is
        voxelized as it stands instead of coord / grid_size, and a supplied "sparse_shape" replaces max + pad
        (structure.py:213-224)."""
        kw = dict(pad=pad, hash_method=self.hash_method, pooling_method=self.pooling_method, seed=self.pooling_seed)
        if "grid_coord" in input_dict:
            gc = input_dict["grid_coord"]
            if int(gc.min()) != 0 or not bool((gc.min(0)[0] == 0).all()):
                # the voxelizer shifts by the batch minimum; the reference keeps a supplied grid_coord as it is
                raise ValueError("grid_coord must start at 0 on every axis (as GridSample produces it)")
            vox = ops.voxelize(gc.float(), input_dict["feat"], input_dict["offset"], 1.0, **kw)
        else:
            vox = ops.voxelize(input_dict["coord"], input_dict["feat"], input_dict["offset"],
                               input_dict["grid_size"], **kw)
        if "sparse_shape" in input_dict:
            vox["spatial_shape"] = [int(v) for v in input_dict["sparse_shape"]]
        x = sparse.SparseConvTensor(vox["voxel_feats"], vox["voxel_coords"], vox["spatial_shape"],
                                    vox["batch_size"])
        return x, vox

    def _fold_modulations(self, condition, context, dtype):
        """With one context row per batch every PDBatchNorm of the network reduces to a batch norm with weight
        gamma*(1+scale) and bias beta*(1+scale)+shift (see PDBatchNorm.forward).  All of them are formed here from
        two matrix-vector products over the concatenated modulation weights, instead of 59 products and five
        [C]-vector kernels per norm; each norm picks its slice up on its next call."""
        norms = self.__dict__.get("_pd_norms")
        if norms is None:
            norms = [m for m in self.modules() if isinstance(m, PDBatchNorm) and m.adaptive and m.affine]
            self.__dict__["_pd_norms"] = norms
        if context is None or context.shape[0] != 1 or not norms or os.environ.get("M3D_FOLD_MODULATIONS", "1") == "0":
            return
        sizes = [m.modulation[1].out_features // 2 for m in norms]
        lin = [m.modulation[1] for m in norms]
        act = F.silu(context.to(dtype))
        # one product over the whole weights ([shift rows; scale rows] per norm, v1m3:66), then two gathers that
        # collect all shifts and all scales: slicing every weight instead would cost a zero fill and a copy per slice
        # in the backward pass
        mod = F.linear(act, torch.cat([l.weight for l in lin]), torch.cat([l.bias for l in lin]))[0]
        picks = self.__dict__.get("_mod_picks")
        if picks is None or picks[0].device != mod.device:
            shift_idx, scale_idx, base = [], [], 0
            for c in sizes:
                shift_idx.append(torch.arange(base, base + c))
                scale_idx.append(torch.arange(base + c, base + 2 * c))
                base += 2 * c
            picks = (torch.cat(shift_idx).to(mod.device), torch.cat(scale_idx).to(mod.device))
            self.__dict__["_mod_picks"] = picks
        shift, scale = mod.index_select(0, picks[0]), mod.index_select(0, picks[1])
        bns = [m.pick(condition) for m in norms]
        gain = 1.0 + scale
        weight = torch.cat([b.weight for b in bns]) * gain
        bias = torch.cat([b.bias for b in bns]) * gain + shift
        for m, w, b in zip(norms, weight.split(sizes), bias.split(sizes)):
            m.__dict__["_affine"] = (w, b)

    def forward(self, input_dict):
        """Returns dict(sparse_conv_feat, feat [N0, out], v2p_map [N_pts]) -- the fields of the
        reference's Point that the head reads (language_module.py:202-207)."""
        x, vox = self.sparsify(input_dict)
        condition = input_dict["condition"][0]
        context = input_dict.get("context")
        try:
            self._fold_modulations(condition, context, x.features.dtype)
            return self._forward_net(x, vox, condition, context)
        finally:
            # the folded affine pairs are for this call only: a forward that raised half-way must not leave any
            # behind for the next call (or for a direct call of a sub-module)
            for m in self.__dict__.get("_pd_norms") or ():
                m.__dict__.pop("_affine", None)

    def _forward_net(self, x, vox, condition, context):
        x = self.conv_input([x, condition, context])
        if context is not None:
            context = context.to(x.features.dtype)
        skips = [x]
        for s in range(self.num_stages):
            x = self.down[s]([x, condition, context])
            x, _, _ = self.enc[s]([x, condition, context])
            skips.append(x)
        x = skips.pop(-1)
        for s in reversed(range(self.num_stages)):
            x = self.up[s]([x, condition, context])
            skip = skips.pop(-1)
            x = x.replace_feature(torch.cat((x.features, skip.features), dim=1))
            x, _, _ = self.dec[s]([x, condition, context])
        if not self.skip_final:
            x = self.final(x)
        return dict(sparse_conv_feat=x, feat=x.features, v2p_map=vox["v2p_map"], voxel=vox)


class PPT(nn.Module):
    """Context embedding per dataset condition in front of the backbone (ppt/model.py:7-22)."""

    def __init__(self, backbone: nn.Module, conditions: Sequence[str] = CONDITIONS, context_channels=256):
        super().__init__()
        self.backbone = backbone
        self.conditions = tuple(conditions)
        self.embedding_table = nn.Embedding(len(self.conditions), context_channels)

    def forward(self, batch_dict: Dict):
        condition = batch_dict["condition"][0]
        assert condition in self.conditions
        idx = torch.tensor([self.conditions.index(condition)], device=batch_dict["coord"].device)
        batch_dict["context"] = self.embedding_table(idx)
        return self.backbone(batch_dict)


def load_oracle_params(model: PPT, params: Dict[str, torch.Tensor]):
    """Load a flat parameter dict keyed like the reference backbone's state_dict (+ 'embedding_table.weight')."""
    sd = {}
    for k, v in params.items():
        sd[k if k.startswith("embedding_table") else "backbone." + k] = v
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if not m.endswith("num_batches_tracked")]
    if missing or unexpected:
        raise KeyError(f"parameter mismatch: missing={missing[:5]} unexpected={unexpected[:5]}")
    return model
