"""Host-side mirror of the `spconv.pytorch` surface Mosaic3D's model code uses, backed by the
gfx950 kernels in libm3d_hip.so.

Surface (exhaustive for the reference's src/): SparseConvTensor, SubMConv3d, SparseConv3d,
SparseInverseConv3d, SparseSequential, SparseModule, ConvAlgo, modules.is_spconv_module,
core.ImplicitGemmIndiceData -- see segmentation_amd.compat for the module aliases.
Reference call sites: mosaic3d/src/models/utils/structure.py:243-248, :304-347;
mosaic3d/src/models/networks/spunet/spconv_unet_v1m3_pdnorm.py:71-221.

Conventions kept from spconv 2.x so that existing checkpoints load: conv weights are
[Cout, k0, k1, k2, Cin]; SubMConv3d / SparseConv3d / SparseInverseConv3d are siblings (the
reference's _init_weights uses isinstance(m, SubMConv3d), v1m3:362-378)."""
import enum
import os
import math
from collections import OrderedDict
from typing import List, Optional, Sequence, Union

import torch
import torch.nn as nn

from . import ops


# How the sparse-convolution modules form their f32 products: "f32" (exact, v_mfma_f32_32x32x2_f32),
# "bf16x6" (3-term bf16 split, f32-level accuracy), "bf16x3" (2-term split, ~1e-5 relative), "bf16", or "f16x3"
# (f16 pairs, f32-level accuracy inside the f16 range; inference only: gradients underflow that range unnoticed).
_PRECISION = "f32"


def set_precision(precision: str):
    global _PRECISION
    if precision not in ops.PRECISION:
        raise ValueError(f"unknown precision {precision!r}; choose from {sorted(ops.PRECISION)}")
    _PRECISION = precision


def get_precision() -> str:
    return _PRECISION


class ConvAlgo(enum.Enum):
    Native = 0
    MaskImplicitGemm = 1
    MaskSplitImplicitGemm = 2


class ImplicitGemmIndiceData:
    """Cached kernel map of one `indice_key` (what spconv keeps in SparseConvTensor.indice_dict).

    pair_fwd [K, N_out] / pair_bwd [K, N_in] follow the canonical definitions in include/m3d.h;
    for a submanifold map pair_bwd is None and N_out == N_in.  Tile schedules are built lazily,
    once per map, and shared by every convolution that uses the key."""

    def __init__(self, *, pair_fwd, pair_bwd, out_indices, indices, out_spatial_shape, spatial_shape,
                 ksize, stride, padding, dilation, is_subm, batch_size=1, algo=ConvAlgo.MaskImplicitGemm):
        self.pair_fwd = pair_fwd
        self.pair_bwd = pair_bwd
        self.out_indices = out_indices
        self.indices = indices
        self.out_spatial_shape = list(out_spatial_shape)
        self.spatial_shape = list(spatial_shape)
        self.ksize, self.stride, self.padding, self.dilation = ksize, stride, padding, dilation
        self.is_subm = is_subm
        self.batch_size = batch_size
        self.algo = algo
        self._sched_fwd = None
        self._sched_bwd = None

    @property
    def sched_fwd(self):
        if self._sched_fwd is None:
            n_out = self.out_indices.shape[0]
            # submanifold maps walk the voxels cell by cell, grouped by neighbour pattern (3x3x3) or in
            # Morton order (other kernel sizes); strided outputs are already in row-major spatial order
            mode = (3 if self.pair_fwd.shape[0] == 27 else 1) if self.is_subm else 0
            self._sched_fwd = ops.build_schedule(self.pair_fwd, n_out, self.out_indices if mode else None, mode,
                                                 self.out_spatial_shape, self.batch_size, tile_order=mode == 3)
        return self._sched_fwd

    @property
    def sched_bwd(self):
        if self._sched_bwd is None:
            if self.pair_bwd is None:
                raise RuntimeError("submanifold kernel maps have no inverse schedule")
            n = self.indices.shape[0]
            morton = ops.build_schedule(self.pair_bwd, n, self.indices, 1, self.spatial_shape, self.batch_size)
            self._sched_bwd = ops.build_schedule(self.pair_bwd, n, None, 2, base_order=morton.out_row)
        return self._sched_bwd


class SparseConvTensor:
    """features [N, C] + indices [N, 4] int32 (batch, d0, d1, d2) + spatial_shape + batch_size."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None,
                 indice_dict=None, benchmark=False, **_ignored):
        self._features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {} if indice_dict is None else indice_dict
        self.grid = grid
        self.voxel_num = voxel_num
        self.benchmark = benchmark

    @property
    def features(self):
        return self._features

    @features.setter
    def features(self, value):
        self._features = value

    def replace_feature(self, feature):
        """Shallow copy sharing indices and the kernel-map cache (spconv semantics)."""
        t = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.grid,
                             self.voxel_num, self.indice_dict, self.benchmark)
        return t

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    @property
    def spatial_size(self):
        return int(torch.tensor(self.spatial_shape).prod())

    def dense(self, channels_first=True):
        """Densify (debug/test helper; small tensors only)."""
        c = self.features.shape[1]
        out = torch.zeros((self.batch_size, *self.spatial_shape, c), dtype=self.features.dtype,
                          device=self.features.device)
        idx = self.indices.long()
        out[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = self.features
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out


class SparseModule(nn.Module):
    """Marker base class: containers pass SparseConvTensor (or the reference's
    [tensor, condition, context] lists) straight to subclasses of it."""


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, subm=False, output_padding=0, transposed=False, inverse=False,
                 indice_key=None, algo: Optional[ConvAlgo] = None, fp32_accum=None, record_voxel_count=False,
                 act_type=None, act_alpha=0, act_beta=0, large_kernel_fast_algo=False, name=None):
        super().__init__()
        if ndim != 3:
            raise NotImplementedError("only 3-D sparse convolutions are implemented")
        if groups != 1 or transposed:
            raise NotImplementedError("grouped / transposed sparse convolutions are not implemented")
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.padding = _triple(padding)
        self.dilation = _triple(dilation)
        self.subm = subm
        self.inverse = inverse
        self.indice_key = indice_key
        self.algo = algo
        self.conv1x1 = all(k == 1 for k in self.kernel_size) and all(s == 1 for s in self.stride)
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self._packed = None
        self._packed_key = None
        self.reset_parameters()

    def reset_parameters(self):
        # spconv's class default: kaiming-uniform with a = sqrt(5) on fan_in = Cin * prod(kernel)
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, bias={self.bias is not None}, subm={self.subm}, "
                f"inverse={self.inverse}, indice_key={self.indice_key}")

    # -- weights in the kernel's layout, re-packed only when the parameter changes
    def packed_weight(self):
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device, _PRECISION)
        if self._packed_key != key:
            self._packed = ops.pack_weight(self.weight, _PRECISION)
            self._packed_key = key
        return self._packed

    # -- kernel maps
    def _get_indice_data(self, x: SparseConvTensor) -> ImplicitGemmIndiceData:
        data = x.find_indice_pair(self.indice_key)
        if self.inverse:
            if data is None:
                raise ValueError(f"SparseInverseConv3d: no cached kernel map for indice_key={self.indice_key!r}")
            if data.is_subm or data.ksize != self.kernel_size:
                raise ValueError("SparseInverseConv3d: cached kernel map does not match this layer")
            if data.out_indices.shape[0] != x.indices.shape[0]:
                raise ValueError("SparseInverseConv3d: input does not match the cached kernel map's output")
            return data
        if data is not None:
            if data.ksize != self.kernel_size or data.is_subm != self.subm:
                raise ValueError(f"indice_key {self.indice_key!r} is cached with a different convolution geometry")
            if data.indices.shape[0] != x.indices.shape[0]:
                raise ValueError(f"indice_key {self.indice_key!r}: cached kernel map has a different voxel count")
            return data
        if self.subm:
            if self.kernel_size != [3, 3, 3] or self.dilation != [1, 1, 1]:
                raise NotImplementedError("SubMConv3d supports kernel_size 3 (and 1), dilation 1")
            pair = ops.subm_rulebook(x.indices, x.spatial_shape, x.batch_size)
            data = ImplicitGemmIndiceData(pair_fwd=pair, pair_bwd=None, out_indices=x.indices, indices=x.indices,
                                          out_spatial_shape=x.spatial_shape, spatial_shape=x.spatial_shape,
                                          ksize=self.kernel_size, stride=self.stride, padding=self.padding,
                                          dilation=self.dilation, is_subm=True, batch_size=x.batch_size)
        else:
            if (self.kernel_size != [2, 2, 2] or self.stride != [2, 2, 2] or self.padding != [0, 0, 0]
                    or self.dilation != [1, 1, 1]):
                raise NotImplementedError("SparseConv3d supports kernel_size 2, stride 2, padding 0")
            oi, oshape, pf, pb = ops.strided_rulebook(x.indices, x.spatial_shape, x.batch_size)
            data = ImplicitGemmIndiceData(pair_fwd=pf, pair_bwd=pb, out_indices=oi, indices=x.indices,
                                          out_spatial_shape=oshape, spatial_shape=x.spatial_shape,
                                          ksize=self.kernel_size, stride=self.stride, padding=self.padding,
                                          dilation=self.dilation, is_subm=False, batch_size=x.batch_size)
        if self.indice_key is not None:
            x.indice_dict[self.indice_key] = data
        return data

    def forward(self, x: SparseConvTensor):
        if not isinstance(x, SparseConvTensor):
            raise TypeError(f"expected SparseConvTensor, got {type(x).__name__}")
        feats = x.features
        if feats.shape[1] != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {feats.shape[1]}")
        if torch.is_grad_enabled() and (feats.requires_grad or self.weight.requires_grad
                                        or (self.bias is not None and self.bias.requires_grad)):
            return self._forward_train(x)
        in_dtype = feats.dtype
        feats = feats.float().contiguous()
        w = self.packed_weight()
        bias = None if self.bias is None else self.bias.detach().float().contiguous()
        if self.conv1x1:
            out = ops.conv_fwd(feats, w, None, feats.shape[0], shift=bias)
            res = x.replace_feature(out.to(in_dtype))
            return res
        data = self._get_indice_data(x)
        if self.inverse:
            out = ops.conv_fwd(feats, w, data.sched_bwd, data.indices.shape[0], shift=bias)
            res = SparseConvTensor(out.to(in_dtype), data.indices, data.spatial_shape, x.batch_size, x.grid,
                                   x.voxel_num, x.indice_dict, x.benchmark)
        else:
            out = ops.conv_fwd(feats, w, data.sched_fwd, data.out_indices.shape[0], shift=bias)
            res = SparseConvTensor(out.to(in_dtype), data.out_indices, data.out_spatial_shape, x.batch_size,
                                   x.grid, x.voxel_num, x.indice_dict, x.benchmark)
        return res


    def _forward_train(self, x: SparseConvTensor):
        """Differentiable path (exact f32): the same kernels under a torch.autograd.Function."""
        feats = x.features.float()
        if self.conv1x1:
            out = _SparseConvFn.apply(feats, self.weight, self.bias, None, None, feats.shape[0], feats.shape[0], False)
            return x.replace_feature(out)
        data = self._get_indice_data(x)
        if self.inverse:
            n_out, n_in = data.indices.shape[0], data.out_indices.shape[0]
            out = _SparseConvFn.apply(feats, self.weight, self.bias, data.sched_bwd, data.sched_fwd, n_out, n_in, False)
            return SparseConvTensor(out, data.indices, data.spatial_shape, x.batch_size, x.grid, x.voxel_num,
                                    x.indice_dict, x.benchmark)
        n_out, n_in = data.out_indices.shape[0], data.indices.shape[0]
        back = data.sched_fwd if data.is_subm else data.sched_bwd
        out = _SparseConvFn.apply(feats, self.weight, self.bias, data.sched_fwd, back, n_out, n_in, data.is_subm)
        return SparseConvTensor(out, data.out_indices, data.out_spatial_shape, x.batch_size, x.grid, x.voxel_num,
                                x.indice_dict, x.benchmark)


_SPLIT_FIRST_ROWS = int(os.environ.get("M3D_SPLIT_FIRST_ROWS", "1500"))


def _maybe_split(x, w):
    """Large f32 activations of a bf16x3 convolution are split into the two-term format once, so that the
    convolution runs on the LDS-DMA kernel (which needs both channel counts to be multiples of 32); small ones are
    split inside the register-staged kernel."""
    if (w.precision == "bf16x3" and w.cin % 32 == 0 and w.cout % 32 == 0 and x.shape[1] == w.cin and _SPLIT_FIRST_ROWS
            and x.shape[0] >= _SPLIT_FIRST_ROWS):
        return ops.split_f32(x, 2)
    return x


class _SparseConvFn(torch.autograd.Function):
    """out = conv(feats, weight) + bias through m3d_conv_fwd_f32; the input gradient is the forward kernel on the
    transposed weights over the transposed kernel map (for a submanifold map: the same schedule with the kernel
    offsets reversed), the weight gradient is m3d_conv_bwd_weight_f32 / _bf16x3.  All three form their products as
    set_precision() says (f32 activations are split inside the kernels; "bf16x6" keeps the exact f32 weight
    gradient)."""

    @staticmethod
    def forward(ctx, feats, weight, bias, sched_fwd, sched_back, n_out, n_in, flip):
        if _PRECISION == "f16x3":
            raise ValueError('precision "f16x3" is for inference: gradients are far smaller than the f16 range '
                             'resolves and the range check only sees overflow; train with "bf16x3", "bf16x6" or "f32"')
        feats = feats.contiguous()
        w = ops.pack_weight(weight.detach(), _PRECISION)
        out = ops.conv_fwd(_maybe_split(feats, w), w, sched_fwd, n_out,
                           shift=None if bias is None else bias.detach().float().contiguous())
        ctx.save_for_backward(feats, weight)
        ctx.meta = (sched_fwd, sched_back, n_out, n_in, flip, bias is not None, _PRECISION)
        return out

    @staticmethod
    def backward(ctx, dout):
        feats, weight = ctx.saved_tensors
        sched_fwd, sched_back, n_out, n_in, flip, has_bias, precision = ctx.meta
        dout = dout.contiguous().float()
        cout, cin = weight.shape[0], weight.shape[-1]
        k = weight.numel() // (cout * cin)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = ops.pack_weight(weight, precision, transposed=True, flip=flip)  # W'[ci][k'][co], one pass
            dx = ops.conv_fwd(_maybe_split(dout, wt), wt, sched_back, n_in)
        if ctx.needs_input_grad[1]:
            dw = ops.conv_bwd_weight(feats, dout, sched_fwd, k,
                                     precision="bf16x3" if precision == "bf16x3" else "f32").reshape(weight.shape)
        if has_bias and ctx.needs_input_grad[2]:
            db = dout.sum(0)
        return dx, dw, db, None, None, None, None, None


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, algo=None, fp32_accum=None, large_kernel_fast_algo=False, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         subm=True, indice_key=indice_key, algo=algo, name=name)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, algo=None, fp32_accum=None, record_voxel_count=False,
                 large_kernel_fast_algo=False, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         subm=False, indice_key=indice_key, algo=algo, name=name)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True, algo=None,
                 fp32_accum=None, large_kernel_fast_algo=False, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                         indice_key=indice_key, algo=algo, name=name)


class SparseSequential(SparseModule):
    """Sequential container: sparse modules receive the running value as is (a SparseConvTensor or
    the reference's [tensor, condition, context] list, v1m3:290-308,396); plain nn.Modules are
    applied to `.features`."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError(f"index {idx} is out of range")
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        for module in self._modules.values():
            if is_spconv_module(module):
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    input = input.replace_feature(module(input.features))
            else:
                input = module(input)
        return input


# ----------------------------------------------------------------------------- torch_scatter subset
def _on_kernels(src):
    """[rows, c] f32 on the GPU goes to the HIP kernels (m3d_scatter_rows / m3d_segment_csr); anything else (CPU
    tensors of the reference's own CPU code paths and of the CPU tests, other ranks and dtypes) to torch operators."""
    return src.is_cuda and src.dtype == torch.float32 and src.dim() == 2


class _ScatterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, n_out, mean, out0):
        from . import _lib, ops
        lib = _lib.load()
        src = src.contiguous()
        index = index.contiguous().long()
        ops._need_cuda(src, index)
        out = torch.zeros((n_out, src.shape[1]), dtype=torch.float32, device=src.device) if out0 is None else out0
        count = torch.empty((n_out,), dtype=torch.int32, device=src.device) if mean else None
        _lib.check(lib.m3d_scatter_rows(ops._ptr(src), ops._ptr(index), src.shape[0], src.shape[1], n_out, int(mean),
                                        ops._ptr(out), ops._ptr(count), ops._stream()), "m3d_scatter_rows")
        ctx.save_for_backward(index, count)
        return out

    @staticmethod
    def backward(ctx, g):
        index, count = ctx.saved_tensors
        gs = g[index]
        if count is not None:
            gs = gs / count.clamp(min=1)[index].to(g.dtype).unsqueeze(1)
        return gs, None, None, None, None


class _SegmentCsrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, indptr, mean):
        from . import _lib, ops
        lib = _lib.load()
        src = src.contiguous()
        indptr = indptr.to(src.device).contiguous().long()
        ops._need_cuda(src, indptr)
        n_seg = indptr.numel() - 1
        out = torch.empty((n_seg, src.shape[1]), dtype=torch.float32, device=src.device)
        _lib.check(lib.m3d_segment_csr(ops._ptr(src), src.shape[1], ops._ptr(indptr), n_seg, int(mean), ops._ptr(out),
                                       ops._stream()), "m3d_segment_csr")
        ctx.save_for_backward(indptr)
        ctx.mean, ctx.rows = mean, src.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        (indptr,) = ctx.saved_tensors
        counts = indptr[1:] - indptr[:-1]
        if ctx.mean:
            g = g / counts.clamp(min=1).to(g.dtype).unsqueeze(1)
        seg = torch.repeat_interleave(torch.arange(counts.numel(), device=g.device), counts)
        gs = torch.zeros((ctx.rows, g.shape[1]), dtype=g.dtype, device=g.device)
        lo, hi = int(indptr[0]), int(indptr[-1])
        gs[lo:hi] = g[seg]
        return gs, None, None


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    """torch_scatter.scatter for reduce in {sum, mean} along dim 0 (structure.py:58, v1m3:410)."""
    if dim != 0:
        raise NotImplementedError("scatter: only dim=0 is implemented")
    if reduce not in ("sum", "add", "mean"):
        raise NotImplementedError(f"scatter: reduce={reduce!r} is not implemented")
    if _on_kernels(src) and (out is None or (out.is_contiguous() and out.dtype == torch.float32)):
        n = out.shape[0] if out is not None else (int(index.max()) + 1 if dim_size is None else dim_size)
        return _ScatterFn.apply(src, index, n, reduce == "mean", out)
    n = int(index.max()) + 1 if dim_size is None else dim_size
    if out is None:
        out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    out.index_add_(0, index, src)
    if reduce == "mean":
        cnt = torch.bincount(index, minlength=out.shape[0]).clamp_(min=1).to(src.dtype)
        out = out / cnt.view(-1, *([1] * (src.dim() - 1)))
    return out


def segment_csr(src, indptr, out=None, reduce="sum"):
    """torch_scatter.segment_csr for reduce in {sum, mean} (caption_loss.py:223)."""
    if reduce not in ("sum", "add", "mean"):
        raise NotImplementedError(f"segment_csr: reduce={reduce!r} is not implemented")
    if _on_kernels(src) and out is None:
        return _SegmentCsrFn.apply(src, indptr, reduce == "mean")
    counts = indptr[1:] - indptr[:-1]
    seg = torch.repeat_interleave(torch.arange(counts.numel(), device=src.device), counts)
    res = torch.zeros((counts.numel(),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    res.index_add_(0, seg, src[int(indptr[0]):int(indptr[-1])])
    if reduce == "mean":
        res = res / counts.clamp(min=1).to(src.dtype).view(-1, *([1] * (src.dim() - 1)))
    return res
