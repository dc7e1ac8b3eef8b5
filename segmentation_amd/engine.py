"""Fused inference engine for the PPT SpUNet hot path:

    voxelize -> kernel maps -> SpUNet forward (folded PDNorm, fused epilogues, concat-free decoder)
             -> CLIP-text logits -> per-voxel argmax -> per-point labels

Same inputs and outputs as the reference's DenseLanguageLitModule.forward + CLIPAlignmentEval.predict
(mosaic3d/src/models/lightning_modules/language_module.py:197-207,314-328,
 mosaic3d/src/models/losses/clip_alignment_loss.py:138-147), same network as
spconv_unet_v1m3_pdnorm.py:380-414 in eval mode.  What is restructured for MI355X:

* PDBatchNorm (eval) + context modulation is folded to one per-channel (scale, shift) and applied in
  the convolution epilogue together with the residual add and ReLU (v1m3:57-68,121-140);
* torch.cat((x, skip)) (v1m3:404) is never materialised: the convolution reads the two tensors as
  two K-slabs;
* the head runs per voxel and the labels are gathered through v2p_map (argmax commutes with the
  row gather); the [N_pts, 768] clip_feat tensor is only produced on request.
"""
from typing import Dict, Optional, Sequence

import collections
import os
import time
import torch

import ctypes

from . import _lib, ops, sparse


def _fold_norm(sd: Dict[str, torch.Tensor], prefix: str, cond_idx: int, context: Optional[torch.Tensor],
               eps: float = 1e-3):
    """(scale, shift) such that PDBatchNorm_eval(x) == scale * x + shift, folded in float64."""
    key = f"{prefix}.bns.{cond_idx}" if f"{prefix}.bns.{cond_idx}.running_mean" in sd else f"{prefix}.bn"
    mean = sd[f"{key}.running_mean"].double()
    var = sd[f"{key}.running_var"].double()
    w = sd[f"{key}.weight"].double() if f"{key}.weight" in sd else torch.ones_like(mean)
    b = sd[f"{key}.bias"].double() if f"{key}.bias" in sd else torch.zeros_like(mean)
    a = w / torch.sqrt(var + eps)
    c = b - mean * a
    mw = sd.get(f"{prefix}.modulation.1.weight")
    if mw is not None:
        mod = torch.nn.functional.linear(torch.nn.functional.silu(context.double()), mw.double(),
                                         sd[f"{prefix}.modulation.1.bias"].double())
        shift, scale = mod.chunk(2, dim=1)
        a = a * (1.0 + scale[0])
        c = c * (1.0 + scale[0]) + shift[0]
    return a.float().contiguous(), c.float().contiguous()


_PIPE_STREAMS = {}  # device index -> [side stream, extra network streams ...] of predict_many


def _tensors_of(obj, seen=None, cuda_only=True):
    """Every (CUDA) tensor reachable from obj through lists, tuples, dicts and the attributes of plain objects,
    `__slots__` classes included (each tensor once)."""
    seen = set() if seen is None else seen
    if torch.is_tensor(obj):
        if (obj.is_cuda or not cuda_only) and id(obj) not in seen:
            seen.add(id(obj))
            yield obj
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v, seen, cuda_only)
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v, seen, cuda_only)
    elif (hasattr(obj, "__dict__") or hasattr(obj, "__slots__")) and id(obj) not in seen:
        seen.add(id(obj))
        names = list(getattr(obj, "__dict__", {})) + [n for c in type(obj).__mro__ for n in getattr(c, "__slots__", ())]
        for name in names:
            if hasattr(obj, name):
                yield from _tensors_of(getattr(obj, name), seen, cuda_only)


class _Conv:
    __slots__ = ("w", "scale", "shift", "cout")

    def __init__(self, w, scale, shift):
        self.w, self.scale, self.shift = w, scale, shift
        self.cout = w.cout


class _Level:
    """Kernel maps of one resolution level."""
    __slots__ = ("indices", "shape", "n", "subm", "down", "up", "pair_subm", "pair_fwd", "pair_bwd",
                 "next_indices", "next_shape", "n_next")


class _SymLevel:
    """A level of the traced network: schedules are (kind, output level) tags, row counts are not known."""
    __slots__ = ("subm", "down", "up", "n", "n_next")

    def __init__(self, l):
        self.subm, self.down, self.up, self.n, self.n_next = ("subm", l), ("down", l + 1), ("up", l), None, None


class _Trace:
    """Layer list of the network as m3d_net_create takes it (include/m3d.h), recorded by running the engine's own
    forward code on value ids instead of tensors."""
    OPS = dict(conv=0, split=1, head=2, argmax=3, gather=4)
    SCHED = {None: 0, "subm": 1, "down": 2, "up": 3}

    def __init__(self, in_channels):
        self.values = [(0, in_channels, 0)]   # (level, channels, format); value 0 = the voxel features
        self.layers = []
        self.keep = []                        # tensors the layer records point into

    def new(self, level, channels, fmt):
        self.values.append((level, channels, fmt))
        return len(self.values) - 1

    def add(self, op, in0, in1=None, residual=None, out_f32=None, out_split=None, sched=None, k=1, cout=0, c0=0, c1=0,
            relu=False, precision=0, n_classes=0, classes_padded=0, weight=None, scale=None, shift=None, text=None):
        i = lambda v: -1 if v is None else int(v)
        p = lambda t: None if t is None else t.data_ptr()
        self.keep += [t for t in (weight, scale, shift, text) if t is not None]
        self.layers.append(_lib.NetLayer(self.OPS[op], i(in0), i(in1), i(residual), i(out_f32), i(out_split),
                                         self.SCHED[sched], k, cout, c0, c1, int(bool(relu)), precision, n_classes,
                                         classes_padded, p(weight), p(scale), p(shift), p(text)))


class _NetPlan:
    """The engine's network as one library call per scene (m3d_net_run)."""

    def __init__(self, trace, n_levels, classes_padded):
        self.lib = _lib.load()
        self.keep = trace.keep
        self.n_levels, self.classes_padded = n_levels, classes_padded
        layers = (_lib.NetLayer * len(trace.layers))(*trace.layers)
        values = (_lib.NetValue * len(trace.values))(*[_lib.NetValue(*v) for v in trace.values])
        handle = ctypes.c_void_p()
        _lib.check(self.lib.m3d_net_create(layers, len(trace.layers), values, len(trace.values), n_levels,
                                           ctypes.byref(handle)), "m3d_net_create")
        self.handle, self.n_layers = handle, len(trace.layers)
        self.rows = (ctypes.c_int64 * n_levels)()
        self.maps = (_lib.NetMaps * n_levels)()

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.m3d_net_destroy(self.handle)
            self.handle = None

    @staticmethod
    def _fill(dst, sched):
        if isinstance(sched, ops.ArenaSchedule):
            dst.nbr, dst.out_row, dst.tile_mask = sched.ptr(sched.o_nbr), sched.ptr(sched.o_row), sched.ptr(sched.o_mask)
            dst.tile_order = sched.ptr(sched.o_order)
        else:
            dst.nbr, dst.out_row, dst.tile_mask = sched.nbr.data_ptr(), sched.out_row.data_ptr(), sched.tile_mask.data_ptr()
            dst.tile_order = None if sched.tile_order is None else sched.tile_order.data_ptr()
        dst.n, dst.npad, dst.k = sched.n, sched.npad, sched.k

    def run(self, voxel_feats, levels, v2p_map, n_points, want_logits):
        assert len(levels) == self.n_levels
        for l, lv in enumerate(levels):
            self.rows[l] = lv.n
            self._fill(self.maps[l].subm, lv.subm)
            if lv.down is not None:
                self._fill(self.maps[l].down, lv.down)
                self._fill(self.maps[l].up, lv.up)
        dev = voxel_feats.device
        n0 = levels[0].n
        arena = torch.empty((self.lib.m3d_net_arena_bytes(self.handle, self.rows),), dtype=torch.uint8, device=dev)
        vlabel = torch.empty((n0,), dtype=torch.int32, device=dev)
        plabel = torch.empty((n_points,), dtype=torch.int64, device=dev)
        logits = torch.empty((n0, self.classes_padded), dtype=torch.float32, device=dev) if want_logits else None
        ops._need_cuda(voxel_feats, v2p_map)
        rc = self.lib.m3d_net_run(self.handle, self.maps, self.rows, ops._ptr(voxel_feats), ops._ptr(arena), arena.numel(),
                                  ops._ptr(v2p_map), n_points, ops._ptr(vlabel), ops._ptr(plabel), ops._ptr(logits),
                                  self.classes_padded, ops._stream())
        _lib.check(rc, "m3d_net_run")
        return vlabel, plabel, logits


class SpUNetEngine:
    def __init__(self, model_or_state: object, cfg: Dict, condition: str = "ScanNet", device="cuda",
                 precision: str = "f32", label_head: Optional[str] = None):
        """model_or_state: a PPT module (segmentation_amd.spunet.PPT or the reference's) or its state_dict
        ('backbone.*' + 'embedding_table.weight').  cfg: channels / layers / conditions of the backbone.
        label_head: None, or "bf16x3" / "f16x3" / "bf16x6": the precision of the fused head (last layer + text logits +
        argmax) when a call asks for labels only -- the logits then feed nothing but the argmax, and two-term products
        give the same labels as the f32-level ones at half the cost.  Features and logits, when asked for, are always
        computed at the engine's precision."""
        self.device = torch.device(device)
        if label_head is not None and label_head not in ("bf16x3", "f16x3", "bf16x6"):
            raise ValueError("label_head must be None, 'bf16x3', 'f16x3' or 'bf16x6'")
        self.label_head = label_head
        with torch.cuda.device(self.device):
            self._init(model_or_state, cfg, condition, precision)

    def _init(self, model_or_state, cfg, condition, precision):
        sd = model_or_state.state_dict() if hasattr(model_or_state, "state_dict") else dict(model_or_state)
        sd = {k: v.detach().to("cpu") for k, v in sd.items()}
        self._state = sd        # kept for the wider-range twin that predict_checked falls back to
        self._twin = None
        self.pipe_log = None  # list: predict_many appends its host times per scene (diagnostics)
        self._emb = None
        self.cfg = cfg
        self.precision = precision
        # schedule mode of the submanifold maps per level (the last entry repeats): rows regrouped by neighbour
        # pattern inside cells of 256 voxels per side at level 0 and over the whole level at levels 1 and 2, voxel
        # order below.  With the 128-row tiles launched heaviest first, larger cells (purer patterns, fewer active
        # offsets per tile) beat the spatial locality of small ones: cells of 64 / 32 / all measured 2.8 % slower.
        self.sched_modes = [int(m) for m in os.environ.get("M3D_SCHED_MODES", "2051,3843,3843,0").split(",")]
        self._terms = ops.TERMS[precision]
        # levels whose activations travel as plain f32 rows only: the convolutions gather the f32 rows and form the
        # terms after the fragment read -- the same terms, products and bits as on rows split beforehand.  With three
        # terms a level-0 tensor is then written once at 4 bytes per element instead of 6 (+ 4 where a residual needs
        # the f32 copy as well) and gathered at 4: the 30-scene sequence 5.25 -> 5.09 ms per scene with level 0 alone,
        # the same with level 1 added, slower from level 2 on.  With two terms the split rows are 4 bytes per element
        # already and the plain rows only add the forming of the terms (f16x3 3.38 -> 3.62 ms): off.
        default = "0" if self._terms == 3 else ""
        self.act_f32 = set(int(v) for v in os.environ.get("M3D_ACT_F32_LEVELS", default).split(",") if v != "") \
            if self._terms in (2, 3, 4) else set()
        # with a split precision `final` also writes an sf32 copy of the features for the logits GEMM
        self.head_precision = precision
        self.fuse_head = os.environ.get("M3D_FUSE_HEAD", "1") != "0"
        # 128-row tiles of the submanifold convolutions are launched heaviest first (m3d_schedule_tile_order)
        self.tile_order = os.environ.get("M3D_TILE_ORDER", "1") != "0"
        self.channels, self.layers = tuple(cfg["channels"]), tuple(cfg["layers"])
        self.num_stages = len(self.layers) // 2
        self.condition = condition
        ci = list(cfg["conditions"]).index(condition)
        ctx = sd["embedding_table.weight"][ci:ci + 1] if "embedding_table.weight" in sd else None
        bsd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
        self._sd = bsd

        def conv(wname, nname=None, bias=None):
            w = ops.pack_weight(bsd[wname], precision, self.device)
            if nname is not None:
                a, c = _fold_norm(bsd, nname, ci, ctx)
            else:
                a = None
                c = bsd[bias].float().contiguous() if bias is not None and bias in bsd else None
            return self._make_conv(w, a, c)

        def block(prefix):
            d = dict(conv1=conv(f"{prefix}.conv1.weight", f"{prefix}.bn1"),
                     conv2=conv(f"{prefix}.conv2.weight", f"{prefix}.bn2"))
            if f"{prefix}.proj_conv.weight" in bsd:
                d["proj"] = conv(f"{prefix}.proj_conv.weight", f"{prefix}.proj_norm")
            return d

        ch, layers = self.channels, self.layers
        self.stem = conv("conv_input.conv.weight", "conv_input.bn")
        self.down = [conv(f"down.{s}.conv.weight", f"down.{s}.bn") for s in range(self.num_stages)]
        self.up = [conv(f"up.{s}.conv.weight", f"up.{s}.bn") for s in range(self.num_stages)]
        self.enc = [[block(f"enc.{s}.block{i}") for i in range(layers[s])] for s in range(self.num_stages)]
        self.dec = [[block(f"dec.{s}.block{i}") for i in range(layers[len(ch) - s - 1])]
                    for s in range(self.num_stages)]
        self.final = conv("final.weight", None, "final.bias")
        self.final_head = None   # `final` in the format of the fused head where that differs from the backbone's
        # the network as one library call per scene (m3d_net_run) instead of one call per layer from here: same
        # kernels and results, a fraction of the host time.  M3D_NET_PLAN=0 keeps the layer-by-layer path.
        self.use_plan = os.environ.get("M3D_NET_PLAN", "1") != "0"
        # kernel maps and schedules of all levels through m3d_levels_begin / m3d_levels_finish (two library calls)
        self.use_level_chain = os.environ.get("M3D_LEVEL_CHAIN", "1") != "0"
        self._trace = None
        self._plan = None
        self.text = None
        self.n_classes = 0
        self.last_stats = {}

    def _make_conv(self, w, scale, shift):
        """A layer record; a weight packed with a power-of-two factor ("f16x3") has the inverse folded into the
        epilogue's scale vector here, once, so that the per-call path never forms it."""
        if scale is not None:
            scale = scale.to(self.device)
        if w.exp:
            scale = (torch.ones(w.cout, device=self.device) if scale is None else scale) * w.unscale
            w.exp = 0
        return _Conv(w, None if scale is None else scale.float().contiguous(),
                     None if shift is None else shift.to(self.device))

    def f16_range_exceeded(self, clear=True):
        """Precision "f16x3" only: True if an activation left the f16 range (|x| > 65504) in any call on this device
        since the status was last cleared; such results are invalid and should be redone with "bf16x6".  Waits for
        the GPU."""
        with torch.cuda.device(self.device):
            return ops.f16_status(clear)

    # ------------------------------------------------------------------ head
    def set_text_embeddings(self, emb: torch.Tensor):
        """emb [C, D] (rows are unit-norm CLIP text features, clip_alignment_loss.py:123-124)."""
        with torch.cuda.device(self.device):
            self._set_text_embeddings(emb)

    def _set_text_embeddings(self, emb: torch.Tensor):
        emb = emb.detach().float().to(self.device)
        self._emb = emb
        self._plan = None
        if self._twin is not None:
            self._twin.set_text_embeddings(emb)
        c, d = emb.shape
        if d != self.final.cout:
            raise ValueError(f"text dim {d} does not match the backbone output {self.final.cout}")
        cpad = 32 if c <= 32 else 64 if c <= 64 else 96 if c <= 96 else (c + 127) // 128 * 128
        w = torch.zeros((cpad, 1, d), dtype=torch.float32, device=self.device)
        w[:c, 0] = emb
        self.text = self._make_conv(ops.pack_weight(w, self.head_precision), None, None)
        self.n_classes = c
        # one-kernel head (last 1x1 layer + logits + argmax): 2- and 3-term split precisions, supported shapes only
        # (plain bf16 runs the two-term head on the two-term split of its last block's f32 output, with `final` packed
        # a second time with two terms: the 768-d features then never travel, and the head is the more accurate one)
        self.text_fused = None
        self._head_terms = ops.TERMS[self.label_head] if self.label_head else (2 if self._terms == 1 else self._terms)
        if self._head_terms in (2, 3, 4) and self.fuse_head and ops.head_fused_supported(self.final.w.cin, c):
            self.text_fused = ops.head_pack_text(emb, cin=self.final.w.cin, terms=self._head_terms)
            if self._head_terms != self._terms and self.final_head is None:
                w2 = ops.pack_weight(self._sd["final.weight"], ops.PRECISION_OF_TERMS[self._head_terms], self.device)
                b2 = self._sd["final.bias"].float().contiguous() if "final.bias" in self._sd else None
                self.final_head = self._make_conv(w2, None, b2)
        # folded head: logits = x96 . (E Wf)^T + E bf   (no nonlinearity between `final` and the head)
        wf = self._sd["final.weight"].reshape(self.final.cout, -1).double().to(self.device)
        wfold = torch.zeros((cpad, 1, wf.shape[1]), dtype=torch.float32, device=self.device)
        wfold[:c, 0] = (emb.double() @ wf).float()
        bfold = torch.zeros((cpad,), dtype=torch.float32, device=self.device)
        if "final.bias" in self._sd:
            bfold[:c] = (emb.double() @ self._sd["final.bias"].double().to(self.device)).float()
        self.text_folded = self._make_conv(ops.pack_weight(wfold, self.precision), None, bfold.contiguous())

    # ------------------------------------------------------------------ kernel maps
    def _subm_schedule(self, l, pair, n, indices, shape, batch_size):
        mode = self.sched_modes[min(l, len(self.sched_modes) - 1)]
        return ops.build_schedule(pair, n, indices, mode, shape, batch_size, tile_order=self.tile_order)

    def build_levels(self, voxel_coords, spatial_shape, batch_size=1):
        """Kernel maps + tile schedules of every resolution level.  The chain of strided maps runs first, with the
        row counts handed from level to level on the device and read back once (behind the level-0 map work); the
        submanifold maps and all schedules are enqueued without any further host read-back, so the host runs ahead
        of the GPU into the network."""
        return self.build_levels_begin(voxel_coords, spatial_shape, batch_size)()

    def build_levels_begin(self, voxel_coords, spatial_shape, batch_size=1):
        """First half of build_levels: enqueues the chain of strided maps and the level-0 map and schedule, and
        returns the function that does the rest (it waits for the level counts, nothing else)."""
        if self.use_level_chain:
            # all maps and schedules through two library calls (same bits as the level-by-level calls below)
            modes = [self.sched_modes[min(l, len(self.sched_modes) - 1)] for l in range(self.num_stages + 1)]
            done = ops.level_chain_begin(voxel_coords, spatial_shape, batch_size, self.num_stages, modes, self.tile_order)

            def finish():
                levels = []
                for d in done():
                    lv = _Level()
                    lv.indices, lv.shape, lv.n, lv.pair_subm, lv.subm = d["indices"], d["shape"], d["n"], d["pair_subm"], d["subm"]
                    lv.pair_fwd, lv.pair_bwd = d.get("pair_fwd"), d.get("pair_bwd")
                    lv.next_indices, lv.next_shape, lv.n_next = d.get("next_indices"), d.get("next_shape"), d.get("n_next", 0)
                    lv.down, lv.up = d.get("down"), d.get("up")
                    levels.append(lv)
                return levels

            return finish
        chain_done = ops.strided_chain(voxel_coords, spatial_shape, batch_size, self.num_stages, defer=True)
        # level 0 needs nothing from the chain: its map and schedule keep the GPU busy while the counts travel
        pair0 = ops.subm_rulebook(voxel_coords, list(spatial_shape), batch_size)
        sched0 = self._subm_schedule(0, pair0, voxel_coords.shape[0], voxel_coords, list(spatial_shape), batch_size)
        return lambda: self._build_levels_finish(voxel_coords, spatial_shape, batch_size, chain_done, pair0, sched0)

    def _build_levels_finish(self, voxel_coords, spatial_shape, batch_size, chain_done, pair0, sched0):
        levels = []
        chain = chain_done()
        idx, shape = voxel_coords, list(spatial_shape)
        for l in range(self.num_stages + 1):
            lv = _Level()
            lv.indices, lv.shape, lv.n = idx, shape, idx.shape[0]
            lv.down = lv.up = lv.pair_fwd = lv.pair_bwd = None
            lv.next_indices = lv.next_shape = None
            lv.n_next = 0
            if l < self.num_stages:
                oi, oshape, pf, pb = chain[l]
                lv.pair_fwd, lv.pair_bwd = pf, pb
                lv.next_indices, lv.next_shape, lv.n_next = oi, oshape, oi.shape[0]
            levels.append(lv)
            idx, shape = lv.next_indices, lv.next_shape
        for l, lv in enumerate(levels):
            if l == 0:
                lv.pair_subm, lv.subm = pair0, sched0
            else:
                lv.pair_subm = ops.subm_rulebook(lv.indices, lv.shape, batch_size)
                lv.subm = self._subm_schedule(l, lv.pair_subm, lv.n, lv.indices, lv.shape, batch_size)
            if lv.pair_fwd is not None:
                lv.down = ops.build_schedule(lv.pair_fwd, lv.n_next, None, 0)
                lv.up = ops.build_schedule(lv.pair_bwd, lv.n, None, 2, base_order=lv.subm.out_row)
        return levels

    # ------------------------------------------------------------------ network
    # Activations travel as (f32 tensor or None, SplitTensor or None).  With precision "f32" only the f32
    # member exists; with the split precisions convolutions consume the sf32 member through the LDS-DMA
    # kernel and an f32 copy is written only where a residual or the caller needs it.
    def _conv(self, c, x, sched, n, x2=None, residual=None, relu=False, need_f32=True, need_sf=True):
        T = self._terms
        if self._trace is not None:
            return self._conv_traced(c, x, sched, x2, residual, relu, need_f32, need_sf)
        if T == 0:
            out = ops.conv_fwd(x[0], c.w, sched, n, in1=None if x2 is None else x2[0], scale=c.scale,
                               shift=c.shift, residual=residual, relu=relu)
            return (out, None)
        if x[1] is not None and c.w.precision != "f32":
            r = ops.conv_fwd(x[1], c.w, sched, n, in1=None if x2 is None else x2[1], scale=c.scale, shift=c.shift,
                             residual=residual, relu=relu, want_f32=need_f32, split_terms=T if need_sf else 0)
        else:  # f32 input (the 3-channel stem runs on the direct kernel; levels kept as plain f32 rows)
            r = ops.conv_fwd(x[0], c.w, sched, n, in1=None if x2 is None else x2[0], scale=c.scale, shift=c.shift,
                             residual=residual, relu=relu, want_f32=need_f32, split_terms=T if need_sf else 0)
        return r if need_sf else (r, None)

    def _conv_traced(self, c, x, sched, x2, residual, relu, need_f32, need_sf):
        """_conv on value ids: records the layer (same choice of inputs and outputs as the code above)."""
        T, tr = self._terms, self._trace
        if T == 0:
            in0, in1, want_f32, split = x[0], None if x2 is None else x2[0], True, 0
        elif x[1] is not None and c.w.precision != "f32":
            in0, in1, want_f32, split = x[1], None if x2 is None else x2[1], need_f32, (T if need_sf else 0)
        else:
            in0, in1, want_f32, split = x[0], None if x2 is None else x2[0], need_f32, (T if need_sf else 0)
        level = tr.values[in0][0] if sched is None else sched[1]
        out_f32 = tr.new(level, c.cout, 0) if want_f32 else None
        out_sf = tr.new(level, c.cout, split) if split else None
        tr.add("conv", in0, in1, residual, out_f32, out_sf, None if sched is None else sched[0], c.w.k, c.cout,
               tr.values[in0][1], 0 if in1 is None else tr.values[in1][1], relu, ops.PRECISION[c.w.precision],
               weight=c.w.data, scale=c.scale, shift=c.shift)
        return (out_f32, out_sf)

    def _build_plan(self):
        """Trace backbone + head (the default call: labels, optionally logits) into a _NetPlan."""
        nl = self.num_stages + 1
        tr = _Trace(self.stem.w.cin)
        levels = [_SymLevel(l) for l in range(nl)]
        self._trace = tr
        try:
            if self.text_fused is not None:
                fin = self.final_head if self.final_head is not None else self.final
                if self._head_terms != self._terms:
                    x = self.backbone(0, levels, last_f32=True)
                    xs = tr.new(0, tr.values[x[0]][1], self._head_terms)
                    tr.add("split", x[0], out_split=xs)
                else:
                    xs = self.backbone(0, levels)[1]
                tr.add("head", xs, cout=self.text_fused.cmid, n_classes=self.text_fused.n_classes,
                       classes_padded=self.text_fused.classes_padded, weight=fin.w.data, scale=fin.scale, shift=fin.shift,
                       text=self.text_fused.data)
                cpad = self.text_fused.classes_padded
            else:
                x = self.backbone(0, levels)
                pair = self._conv(self.final, x, None, None, need_f32=self._terms == 0, need_sf=self._terms != 0)
                src = (None, pair[1]) if self._terms else (pair[0], None)
                logits = self._conv(self.text, src, None, None, need_f32=True, need_sf=False)[0]
                tr.add("argmax", logits, n_classes=self.n_classes)
                cpad = self.text.cout
            tr.add("gather", 0)
        finally:
            self._trace = None
        return _NetPlan(tr, nl, cpad)

    def _block(self, blk, x, sched, n, x2=None, last_needs_f32=True, plain=False, last_needs_sf=True):
        """BasicBlock with fused epilogues; (x, x2) are the two K-slabs of a concatenated input.  plain: the level's
        activations are f32 rows only (last_needs_sf: the block's output is wanted in the split format all the same)."""
        if "proj" in blk:
            res = self._conv(blk["proj"], x, None, n, x2=x2, need_f32=True, need_sf=False)[0]
        else:
            res = x[0]
        t = self._conv(blk["conv1"], x, sched, n, x2=x2, relu=True, need_f32=self._terms == 0 or plain, need_sf=not plain)
        return self._conv(blk["conv2"], t, sched, n, residual=res, relu=True,
                          need_f32=last_needs_f32 or self._terms == 0, need_sf=last_needs_sf or not plain)

    def backbone(self, voxel_feats, levels, keep=None, last_f32=False):
        """SpUNetBase.forward after sparsify (v1m3:388-407).  Returns the activation pair of the last decoder
        block ([N0, channels[-1]])."""
        ns = self.num_stages
        l0 = levels[0]
        f32_only = self._terms == 0
        plain = [l in self.act_f32 for l in range(ns + 1)]
        x = self._conv(self.stem, (voxel_feats, None), l0.subm, l0.n, relu=True,
                       need_f32=f32_only or keep is not None or plain[0], need_sf=not plain[0])
        if keep is not None:
            keep["conv_input"] = x[0]
        skips = [x]
        for s in range(ns):
            x = self._conv(self.down[s], x, levels[s].down, levels[s].n_next, relu=True, need_f32=True,
                           need_sf=not plain[s + 1])
            if keep is not None:
                keep[f"down.{s}"] = x[0]
            for blk in self.enc[s]:
                x = self._block(blk, x, levels[s + 1].subm, levels[s + 1].n, plain=plain[s + 1], last_needs_sf=False)
            if keep is not None:
                keep[f"enc.{s}"] = x[0]
            skips.append(x)
        x = skips.pop(-1)
        for s in reversed(range(ns)):
            x = self._conv(self.up[s], x, levels[s].up, levels[s].n, relu=True,
                           need_f32=f32_only or keep is not None or plain[s], need_sf=not plain[s])
            if keep is not None:
                keep[f"up.{s}"] = x[0]
            skip = skips.pop(-1)
            for i, blk in enumerate(self.dec[s]):
                # the network's last block feeds `final` / the head, which read the split copy only
                last = s == 0 and i == len(self.dec[s]) - 1 and keep is None
                # (its output in the split format also where the level's activations are plain f32 rows)
                x = self._block(blk, x, levels[s].subm, levels[s].n, x2=skip if i == 0 else None,
                                last_needs_f32=last_f32 or not last, plain=plain[s], last_needs_sf=last)
            if keep is not None:
                keep[f"dec.{s}"] = x[0]
        return x

    def _voxelize_and_maps(self, coord, feat, offset, grid_size):
        # (Building the maps on a side stream overlapped with the encoder was tried and measured slower:
        #  the map kernels are latency-bound on their host read-backs, not on GPU time.)
        vox = ops.voxelize(coord, feat, offset, grid_size, pad=128)
        return vox, self.build_levels(vox["voxel_coords"], vox["spatial_shape"], vox["batch_size"])

    def forward_features(self, coord, feat, offset, grid_size, keep=None, want_split=False, want_f32=True):
        """Whole path up to the per-voxel open-vocabulary features.  Returns (features [N0, D], vox dict); with
        want_split the first member is the pair (f32 features or None, split features)."""
        with torch.cuda.device(self.device):   # kernels launch on the current device's stream
            return self._forward_features(coord, feat, offset, grid_size, keep, want_split, want_f32)

    def _forward_features(self, coord, feat, offset, grid_size, keep=None, want_split=False, want_f32=True):
        vox, levels = self._voxelize_and_maps(coord, feat, offset, grid_size)
        x = self.backbone(vox["voxel_feats"], levels, keep)
        out_pair = self._conv(self.final, x, None, levels[0].n, need_f32=want_f32 or not want_split, need_sf=want_split)
        out = out_pair[0]
        if keep is not None:
            keep["final"] = out
            keep["levels"] = [levels[l] for l in range(len(levels))]
        self.last_stats = dict(n_points=int(coord.shape[0]), n_voxels=[levels[l].n for l in range(len(levels))])
        if want_split:
            return out_pair, vox
        return out, vox

    @torch.no_grad()
    def predict(self, coord, feat, offset, grid_size, fold_head=False, return_logits=False, return_features=False):
        """Per-point labels (int64), like logits.max(1)[1] on clip_feat @ emb_target.T.  return_logits adds the
        per-voxel logits of the features as they are (CLIPAlignmentEval with normalize_input=False, as
        configs/model/loss/caption.yaml:7 sets it; normalising the rows first scales every row of logits by a
        positive number and leaves the labels unchanged).  return_features adds the per-voxel 768-d features of the
        same pass as "voxel_feat" (clip_feat of language_module.py:205 is voxel_feat[v2p_map], the record
        language_module.py:349-356 saves); the last layer then writes them out instead of keeping them on chip."""
        if self.text is None:
            raise RuntimeError("call set_text_embeddings() first")
        if return_features and fold_head:
            raise ValueError("fold_head never forms the features; it cannot return them")
        with torch.cuda.device(self.device):
            return self._predict(coord, feat, offset, grid_size, fold_head, return_logits, return_features)

    def predict_checked(self, coord, feat, offset, grid_size, **kw):
        """predict() with the range check of precision "f16x3" applied: the call waits for its result, reads the
        device's f16 status and, if an activation left the f16 range, computes the scene again with an engine of
        precision "bf16x6" built from the same parameters (kept for later calls).  Other precisions: plain predict().
        The returned dict carries "precision", the precision that produced it."""
        if self.precision != "f16x3":
            return dict(self.predict(coord, feat, offset, grid_size, **kw), precision=self.precision)
        self.f16_range_exceeded(clear=True)  # leftovers of earlier, unchecked calls are not this call's business
        out = self.predict(coord, feat, offset, grid_size, **kw)
        if not self.f16_range_exceeded(clear=True):
            return dict(out, precision="f16x3")
        return dict(self._bf16x6_twin().predict(coord, feat, offset, grid_size, **kw), precision="bf16x6")

    def _bf16x6_twin(self):
        if self._twin is None:
            self._twin = SpUNetEngine(self._state, self.cfg, self.condition, self.device, precision="bf16x6")
            if self._emb is not None:
                self._twin.set_text_embeddings(self._emb)
        return self._twin

    def predict_many(self, scenes, fold_head=False, return_logits=False, return_features=False, net_streams=1,
                     check_range=False):
        """predict() for a sequence of scenes, software-pipelined over two streams: while the network and the head
        of scene i run on the current stream, the voxelizer and the map builders of scene i + 1 run on a side stream.
        That first stage is a chain of ~170 small launches with two host read-backs and leaves most of the GPU idle;
        beside the convolutions of the previous scene it costs little.  One host thread does both: it enqueues the
        network of scene i (no read-back inside), then builds the maps of scene i + 1, whose read-backs wait for the
        side stream only.  Results are those of predict(), scene by scene, in order.
        net_streams=k deals the networks of consecutive scenes round-robin to the current stream and k - 1 more, so
        that the latency-bound small levels of one scene run beside the large convolutions of another.
        check_range (precision "f16x3" only): read the f16 status word after the last scene and, if an activation
        left the f16 range anywhere, compute the whole sequence again with the bf16x6 twin (as predict_checked does
        for one scene).
        scenes: iterable of (coord, feat, offset, grid_size); returns a list."""
        if self.text is None:
            raise RuntimeError("call set_text_embeddings() first")
        if return_features and fold_head:
            raise ValueError("fold_head never forms the features; it cannot return them")
        if check_range and self.precision == "f16x3":
            scenes = list(scenes)
            self.f16_range_exceeded(clear=True)
            outs = self.predict_many(scenes, fold_head, return_logits, return_features, net_streams)
            if not self.f16_range_exceeded(clear=True):
                return outs
            return self._bf16x6_twin().predict_many(scenes, fold_head, return_logits, return_features, net_streams)
        with torch.cuda.device(self.device), torch.no_grad():
            main = torch.cuda.current_stream()
            # one set of streams per device, shared by all engines: a process has few hardware queues (4 by default),
            # and streams beyond them share a queue with another stream and stop overlapping with it
            pool = _PIPE_STREAMS.setdefault(self.device.index if self.device.index is not None
                                            else torch.cuda.current_device(), [])
            while len(pool) < net_streams:
                pool.append(torch.cuda.Stream(device=self.device))
            side = pool[0]
            side.wait_stream(main)  # the inputs are ready where the caller's stream stands now
            nets = [main]
            for extra in pool[1:max(1, net_streams)]:
                extra.wait_stream(main)
                nets.append(extra)

            # The first stage of a scene in three host phases, all enqueued on the side stream: A the voxelizer, B (needs
            # the voxel count) the chain of strided maps and the level-0 map, C (needs the level counts) the other maps;
            # each phase's read-back waits for its own small copy only.  The host interleaves them with the network
            # of an earlier scene, so that a read-back finds its result already there instead of idling:
            #   A(i+2)  network(i)  B(i+2)  C(i+1)
            class _Scene:
                __slots__ = ("scene", "n_points", "phase", "fin", "vox", "levels", "ready")

            pending = {}   # scene number -> _Scene
            # a list exists before the call (the wait above covers it); anything else may build each scene when it is
            # asked for, on the caller's stream, behind the networks already enqueued there
            lazy = not isinstance(scenes, (list, tuple))
            it = iter(scenes)
            fetched = [0]

            def advance(j, phase):
                """Run the first-stage phases of scene j up to `phase` (1 = A, 2 = B, 3 = C); False past the end."""
                while fetched[0] <= j:
                    sc = next(it, None)
                    if sc is None:
                        return False
                    if lazy:
                        side.wait_stream(main)   # the scene's tensors are ready where the caller's stream stands now
                    for t in sc[:3]:
                        # the voxelizer reads the inputs on the side stream: their memory must not be handed out again
                        # (to a network on the caller's stream, say) before it has finished, even if the caller and
                        # this function drop them right away
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(side)
                    st = _Scene()
                    st.scene, st.n_points, st.phase, st.fin, st.vox, st.levels, st.ready = sc, 0, 0, None, None, None, None
                    pending[fetched[0]] = st
                    fetched[0] += 1
                st = pending.get(j)
                if st is None:
                    return False
                with torch.cuda.stream(side):
                    if st.phase < 1 <= phase:
                        coord, feat, offset, grid_size = st.scene
                        st.scene, st.n_points = None, int(coord.shape[0])
                        st.fin = ops.voxelize(coord, feat, offset, grid_size, pad=128, defer=True)
                        st.phase = 1
                    if st.phase < 2 <= phase:
                        st.vox = st.fin()
                        st.fin = self.build_levels_begin(st.vox["voxel_coords"], st.vox["spatial_shape"],
                                                         st.vox["batch_size"])
                        st.phase = 2
                    if st.phase < 3 <= phase:
                        st.levels = st.fin()
                        st.fin = None
                        st.ready = torch.cuda.Event()
                        st.ready.record(side)
                        st.phase = 3
                return True

            outs, inflight = [], collections.deque()
            try:
                i = 0
                advance(0, 3)
                advance(1, 2)
                while i in pending:
                    advance(i + 2, 1)
                    advance(i, 3)
                    st = pending.pop(i)
                    n_points, vox, levels, ready = st.n_points, st.vox, st.levels, st.ready
                    net = nets[len(outs) % len(nets)]
                    net.wait_event(ready)
                    t_a = time.perf_counter()
                    with torch.cuda.stream(net):
                        if self.pipe_log is not None:
                            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            ev0.record()
                        out = self._predict(None, None, None, None, fold_head, return_logits, return_features,
                                            maps=(vox, levels, n_points))
                        if self.pipe_log is not None:
                            ev1.record()
                    t_b = time.perf_counter()
                    # the caller will use (and free) the results on its own stream; some were allocated on the side
                    # stream (v2p_map) or on a second network stream
                    for v in out.values():
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(main)
                    outs.append(out)
                    # memory handed out on the side stream must outlive its use on the network's stream: every map
                    # tensor is marked as used there, so the allocator hands its block out again only after the work
                    # enqueued up to here has finished (no host wait: a wait here is time in which the next network
                    # is not being enqueued)
                    marked = 0
                    for t in _tensors_of((vox, levels)):
                        t.record_stream(net)
                        marked += 1
                    # at least the coordinates, the submanifold map and its schedule of every level
                    assert marked >= 5 * len(levels), "map tensors of a scene not found: they would be released early"
                    # bounded run-ahead: the host may be several scenes ahead of the GPU (their maps are held until the
                    # GPU is done with them), never more than eight
                    done = torch.cuda.Event()
                    done.record(net)
                    inflight.append(done)
                    if len(inflight) > 8:
                        inflight.popleft().synchronize()
                    advance(i + 2, 2)
                    advance(i + 1, 3)
                    i += 1
                    if self.pipe_log is not None:  # host seconds: network enqueue, first stage of the next scene
                        self.pipe_log.append((t_b - t_a, time.perf_counter() - t_b, ev0, ev1))
            finally:
                side.synchronize()
                for net in nets[1:]:
                    main.wait_stream(net)
            return outs

    def _predict(self, coord, feat, offset, grid_size, fold_head, return_logits, return_features=False, maps=None):
        vlabel = None
        vfeat = None
        if maps is None:
            vox, levels = self._voxelize_and_maps(coord, feat, offset, grid_size)
            n_points = int(coord.shape[0])
        else:
            vox, levels, n_points = maps
        self.last_stats = dict(n_points=n_points, n_voxels=[levels[l].n for l in range(len(levels))])
        if self.use_plan and not fold_head and not return_features and ops.PROFILE is None \
                and not (return_logits and self.label_head is not None):
            # the whole network and the head as one library call
            if self._plan is None:
                self._plan = self._build_plan()
            vlabel, plabel, logits = self._plan.run(vox["voxel_feats"], levels, vox["v2p_map"], n_points, return_logits)
            out = dict(pred=plabel, voxel_pred=vlabel, v2p_map=vox["v2p_map"], n_voxels=vox["n_voxels"])
            if return_logits:
                out["voxel_logits"] = logits[:, :self.n_classes]
            return out
        # the fused head serves a call unless it asks for the features, or for logits while the head's precision is a
        # labels-only choice (label_head)
        if self.text_fused is not None and not fold_head and not return_features and \
                not (return_logits and self.label_head is not None):
            # `final`, the logits and the argmax in one kernel: the CLIP features stay on chip
            fin = self.final_head if self.final_head is not None else self.final
            if self._head_terms != self._terms:
                # the head's operand format differs from the backbone's: split the f32 copy of the last block's output
                x = self.backbone(vox["voxel_feats"], levels, last_f32=True)
                xs = ops.split_f32(x[0], self._head_terms)
            else:
                xs = self.backbone(vox["voxel_feats"], levels)[1]
            vlabel, logits = ops.head_fused(xs, fin.w, self.text_fused, scale=fin.scale, shift=fin.shift,
                                            want_logits=return_logits)
        elif fold_head:
            x = self.backbone(vox["voxel_feats"], levels)
            t = self.text_folded
            src = x[1] if self._terms else x[0]
            # (a weight packed with a power-of-two factor carries the inverse in t.scale: precision "f16x3")
            logits = ops.conv_fwd(src, t.w, None, levels[0].n, scale=t.scale, shift=t.shift, flop_cout=self.n_classes)
        else:
            # the head consumes the split features directly; the f32 copy is only written when asked for
            x = self.backbone(vox["voxel_feats"], levels)
            pair = self._conv(self.final, x, None, levels[0].n, need_f32=return_features or self._terms == 0,
                              need_sf=self._terms != 0)
            src = pair[1] if self._terms else pair[0]
            vfeat = pair[0]
            logits = ops.conv_fwd(src, self.text.w, None, levels[0].n, scale=self.text.scale, flop_cout=self.n_classes)
        if vlabel is None:
            vlabel = ops.argmax_rows(logits, self.n_classes)
        plabel = ops.gather_labels(vlabel, vox["v2p_map"])
        out = dict(pred=plabel, voxel_pred=vlabel, v2p_map=vox["v2p_map"], n_voxels=vox["n_voxels"])
        if return_logits:
            out["voxel_logits"] = logits[:, :self.n_classes]
        if return_features:
            out["voxel_feat"] = vfeat
        return out
