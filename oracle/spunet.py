"""CPU restatement of the PPT SpUNet (v1m3 PDNorm) forward, unfused, layer by layer.
Test infrastructure only.

Follows /root/reference/mosaic3d/src/models/networks/spunet/spconv_unet_v1m3_pdnorm.py:
PDBatchNorm :19-68, BasicBlock :71-140, SPConvDown :143-170, SPConvUp :173-199,
SPConvPatchEmbedding :202-221, SpUNetBase.__init__ :224-360 (layer list / parameter names),
_init_weights :362-378, forward :380-414; and /root/reference/mosaic3d/src/models/networks/ppt/model.py:15-22
(context = embedding_table[conditions.index(condition)]).

Parameters are a flat dict keyed exactly like the reference module's state_dict
("conv_input.conv.weight", "enc.0.block0.bn1.bns.2.running_var", ...), tensors in the
reference layouts (conv weights [Cout, k0, k1, k2, Cin]).
"""
import math

import numpy as np
import torch

from .rulebook import subm_rulebook, strided_rulebook
from .sparse_conv import subm_conv, strided_conv, inverse_conv, conv1x1
from .voxelize import sparsify

CONDITIONS = ("ScanNet", "ARKitScenes", "Matterport3D", "Structured3D", "ScanNetPP")

SPUNET34C = dict(in_channels=3, out_channels=768, base_channels=32, context_channels=256,
                 channels=(32, 64, 128, 256, 256, 128, 96, 96), layers=(2, 3, 4, 6, 2, 2, 2, 2),
                 conditions=CONDITIONS)
SPUNET18A = dict(in_channels=3, out_channels=768, base_channels=32, context_channels=256,
                 channels=(32, 64, 128, 256, 128, 128, 96, 96), layers=(2, 2, 2, 2, 2, 2, 2, 2),
                 conditions=CONDITIONS)


# ----------------------------------------------------------------------------- parameters
def _trunc_normal(shape, std, gen):
    t = torch.empty(shape)
    torch.nn.init.trunc_normal_(t, std=std, generator=gen)
    return t


def _kaiming_uniform(shape, fan_in, gen):
    bound = 1.0 / math.sqrt(fan_in)
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def make_spunet_params(cfg=SPUNET34C, seed=0, trained_like=False):
    """Random parameters with the reference's shapes and names.

    trained_like=False follows the reference init rules (_init_weights, v1m3:362-378):
    SubMConv3d/Linear trunc_normal(0.02), BN weight 1 / bias 0 / mean 0 / var 1, modulation
    zero-initialised; other convs keep the class default (kaiming-uniform, a=sqrt(5)).
    trained_like=True randomises BN statistics/affine and the modulation so that every term
    of the PDNorm arithmetic is exercised, and scales conv weights to keep activations O(1)."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    nc = len(cfg["conditions"])
    ctx = cfg["context_channels"]

    def norm(prefix, c):
        for i in range(nc):
            if trained_like:
                P[f"{prefix}.bns.{i}.weight"] = 1.0 + 0.2 * torch.randn(c, generator=g)
                P[f"{prefix}.bns.{i}.bias"] = 0.1 * torch.randn(c, generator=g)
                P[f"{prefix}.bns.{i}.running_mean"] = 0.1 * torch.randn(c, generator=g)
                P[f"{prefix}.bns.{i}.running_var"] = 0.5 + torch.rand(c, generator=g)
            else:
                P[f"{prefix}.bns.{i}.weight"] = torch.ones(c)
                P[f"{prefix}.bns.{i}.bias"] = torch.zeros(c)
                P[f"{prefix}.bns.{i}.running_mean"] = torch.zeros(c)
                P[f"{prefix}.bns.{i}.running_var"] = torch.ones(c)
        if trained_like:
            P[f"{prefix}.modulation.1.weight"] = _trunc_normal((2 * c, ctx), 0.02, g)
            P[f"{prefix}.modulation.1.bias"] = 0.05 * torch.randn(2 * c, generator=g)
        else:
            P[f"{prefix}.modulation.1.weight"] = torch.zeros(2 * c, ctx)
            P[f"{prefix}.modulation.1.bias"] = torch.zeros(2 * c)

    def subm(name, cin, cout, k=3):
        if trained_like:
            std = math.sqrt(2.0 / (cin * 9.0))   # ~9 of 27 neighbours are active on a surface
            P[name] = torch.randn((cout, k, k, k, cin), generator=g) * std
        else:
            P[name] = _trunc_normal((cout, k, k, k, cin), 0.02, g)

    def other(name, cin, cout, k):
        fan = cin * k ** 3
        if trained_like:
            fan = cin * (1 if k == 1 else 3)
            P[name] = torch.randn((cout, k, k, k, cin), generator=g) * math.sqrt(2.0 / fan)
        else:
            P[name] = _kaiming_uniform((cout, k, k, k, cin), fan, g)

    def block(prefix, cin, cout):
        if cin != cout:
            other(f"{prefix}.proj_conv.weight", cin, cout, 1)
            norm(f"{prefix}.proj_norm", cout)
        subm(f"{prefix}.conv1.weight", cin, cout)
        norm(f"{prefix}.bn1", cout)
        subm(f"{prefix}.conv2.weight", cout, cout)
        norm(f"{prefix}.bn2", cout)

    ch, layers, base = cfg["channels"], cfg["layers"], cfg["base_channels"]
    ns = len(layers) // 2
    subm("conv_input.conv.weight", cfg["in_channels"], base)
    norm("conv_input.bn", base)
    enc_c, dec_c = base, ch[-1]
    for s in range(ns):
        other(f"down.{s}.conv.weight", enc_c, ch[s], 2)
        norm(f"down.{s}.bn", ch[s])
        for i in range(layers[s]):
            block(f"enc.{s}.block{i}", ch[s], ch[s])
        other(f"up.{s}.conv.weight", ch[len(ch) - s - 2], dec_c, 2)
        norm(f"up.{s}.bn", dec_c)
        for i in range(layers[len(ch) - s - 1]):
            block(f"dec.{s}.block{i}", dec_c + enc_c if i == 0 else dec_c, dec_c)
        enc_c, dec_c = ch[s], ch[len(ch) - s - 2]
    other("final.weight", ch[-1], cfg["out_channels"], 1)
    P["final.bias"] = _kaiming_uniform((cfg["out_channels"],), ch[-1], g)
    P["embedding_table.weight"] = torch.randn(nc, ctx, generator=g)
    return P


# ----------------------------------------------------------------------------- arithmetic
def pdnorm(P, prefix, x, cond_idx, context, eps=1e-3):
    """PDBatchNorm.forward in eval mode (v1m3:57-68): BatchNorm1d(eps=1e-3) of the condition,
    then x * (1 + scale) + shift with (shift, scale) = Linear(SiLU(context)).chunk(2, 1)."""
    w = P[f"{prefix}.bns.{cond_idx}.weight"]
    b = P[f"{prefix}.bns.{cond_idx}.bias"]
    m = P[f"{prefix}.bns.{cond_idx}.running_mean"]
    v = P[f"{prefix}.bns.{cond_idx}.running_var"]
    x = torch.nn.functional.batch_norm(x, m, v, w, b, False, 0.0, eps)
    mw = P.get(f"{prefix}.modulation.1.weight")
    if mw is not None:
        mod = torch.nn.functional.linear(torch.nn.functional.silu(context), mw,
                                         P[f"{prefix}.modulation.1.bias"])
        shift, scale = mod.chunk(2, dim=1)
        x = x * (1.0 + scale) + shift
    return x


def fold_pdnorm(P, prefix, cond_idx, context, eps=1e-3):
    """Per-channel (a, b) with pdnorm(x) == a * x + b in exact arithmetic (float64 fold)."""
    w = P[f"{prefix}.bns.{cond_idx}.weight"].double()
    b = P[f"{prefix}.bns.{cond_idx}.bias"].double()
    m = P[f"{prefix}.bns.{cond_idx}.running_mean"].double()
    v = P[f"{prefix}.bns.{cond_idx}.running_var"].double()
    a = w / torch.sqrt(v + eps)
    c = b - m * a
    mw = P.get(f"{prefix}.modulation.1.weight")
    if mw is not None:
        mod = torch.nn.functional.linear(torch.nn.functional.silu(context.double()), mw.double(),
                                         P[f"{prefix}.modulation.1.bias"].double())
        shift, scale = mod.chunk(2, dim=1)
        a = a * (1.0 + scale[0])
        c = c * (1.0 + scale[0]) + shift[0]
    return a.float(), c.float()


def _block(P, prefix, x, pair, cond_idx, context):
    """BasicBlock.forward (v1m3:121-140)."""
    residual = x
    out = subm_conv(x, P[f"{prefix}.conv1.weight"], pair)
    out = torch.relu(pdnorm(P, f"{prefix}.bn1", out, cond_idx, context))
    out = subm_conv(out, P[f"{prefix}.conv2.weight"], pair)
    out = pdnorm(P, f"{prefix}.bn2", out, cond_idx, context)
    if f"{prefix}.proj_conv.weight" in P:
        residual = pdnorm(P, f"{prefix}.proj_norm", conv1x1(residual, P[f"{prefix}.proj_conv.weight"]),
                          cond_idx, context)
    return torch.relu(out + residual)


def build_rulebooks(voxel_coords, spatial_shape, num_stages=4):
    """All kernel maps one forward needs: subm[l] for levels 0..num_stages and the strided maps
    between consecutive levels.  "stem" and "subm0" share level 0's SubM map."""
    levels = [dict(indices=np.asarray(voxel_coords), shape=list(spatial_shape))]
    for s in range(num_stages):
        cur = levels[-1]
        oi, oshape, pf, pb = strided_rulebook(cur["indices"], cur["shape"])
        cur["pair_fwd"], cur["pair_bwd"] = pf, pb
        levels.append(dict(indices=oi, shape=oshape))
    for lv in levels:
        lv["subm"] = subm_rulebook(lv["indices"], lv["shape"])
    return levels


def spunet_forward(P, cfg, voxel_feats, voxel_coords, spatial_shape, condition="ScanNet",
                   levels=None, return_intermediates=False, dtype=torch.float32):
    """SpUNetBase.forward after sparsify (v1m3:384-407) with PPT's context (ppt/model.py:15-22).
    Returns per-voxel features [N0, out_channels] (and optionally a dict of layer outputs).
    dtype=torch.float64 (with float64 parameters) gives the yardstick that f32 runs are measured against."""
    ch, layers = cfg["channels"], cfg["layers"]
    ns = len(layers) // 2
    ci = list(cfg["conditions"]).index(condition)
    context = P["embedding_table.weight"][ci:ci + 1]
    if levels is None:
        levels = build_rulebooks(voxel_coords, spatial_shape, ns)
    inter = {}
    x = torch.as_tensor(voxel_feats).to(dtype)
    x = subm_conv(x, P["conv_input.conv.weight"], levels[0]["subm"])
    x = torch.relu(pdnorm(P, "conv_input.bn", x, ci, context))
    inter["conv_input"] = x
    skips = [x]
    for s in range(ns):
        x = strided_conv(x, P[f"down.{s}.conv.weight"], levels[s]["pair_fwd"])
        x = torch.relu(pdnorm(P, f"down.{s}.bn", x, ci, context))
        inter[f"down.{s}"] = x
        for i in range(layers[s]):
            x = _block(P, f"enc.{s}.block{i}", x, levels[s + 1]["subm"], ci, context)
        inter[f"enc.{s}"] = x
        skips.append(x)
    x = skips.pop(-1)
    for s in reversed(range(ns)):
        x = inverse_conv(x, P[f"up.{s}.conv.weight"], levels[s]["pair_bwd"])
        x = torch.relu(pdnorm(P, f"up.{s}.bn", x, ci, context))
        inter[f"up.{s}"] = x
        skip = skips.pop(-1)
        x = torch.cat((x, skip), dim=1)
        for i in range(layers[len(ch) - s - 1]):
            x = _block(P, f"dec.{s}.block{i}", x, levels[s]["subm"], ci, context)
        inter[f"dec.{s}"] = x
    x = conv1x1(x, P["final.weight"], P.get("final.bias"))
    inter["final"] = x
    return (x, inter) if return_intermediates else x


def spunet_from_points(P, cfg, coord, feat, offset, grid_size, condition="ScanNet", **kw):
    """Whole reference path from raw points: sparsify (pad=128, v1m3:382) + forward."""
    sp = sparsify(coord, feat, offset, grid_size, pad=128)
    out = spunet_forward(P, cfg, torch.from_numpy(sp["voxel_feats"]), sp["voxel_coords"],
                         sp["spatial_shape"], condition, **kw)
    return sp, out
