"""Tensor-level wrappers of the C ABI: device pointers in, torch tensors out.

PyTorch supplies device memory, the current HIP stream and nothing else; every operator below
runs a hand-written gfx950 kernel from libm3d_hip.so."""
import ctypes
import os
import threading
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

HASH = {"fnv": 0, "ravel": 1}
GRID = {"div": 0, "recip": 1, "floor": 2}

_workspaces = {}

# When set to a list, conv_fwd appends (kind, algorithmic_flops, start_event, end_event) per launch;
# used by bench.py to time the convolution kernel with HIP events on the launch stream.
PROFILE = None


_OWNER_THREAD = {}   # device index -> ident of the first host thread that launched on it
_ALLOW_THREADS = os.environ.get("M3D_ALLOW_THREADS", "0") == "1"


def _check_thread(dev: int):
    """One host thread per GPU drives the operators: workspaces are shared per (device, stream), read-back buffers
    and the allocator's stream bookkeeping (record_stream of map tensors in SpUNetEngine.predict_many) assume that
    the thread that enqueues a producer also enqueues its consumers.  The first thread to launch on a device owns
    it; a launch from another thread raises unless M3D_ALLOW_THREADS=1 (the caller then orders streams, workspaces
    and tensor lifetimes across its threads itself).  The autograd engine's backward thread is exempt: it runs
    while the owner waits in backward(), on the stream the forward ran on."""
    me = threading.get_ident()
    owner = _OWNER_THREAD.setdefault(dev, me)
    if owner != me and not _ALLOW_THREADS and torch._C._current_graph_task_id() == -1:
        raise RuntimeError(f"segmentation_amd operators on cuda:{dev} are driven by one host thread (the first that "
                           "launched there); this call comes from another thread.  Use one process per GPU and one "
                           "thread per process, or set M3D_ALLOW_THREADS=1 and synchronise the threads yourself")


def _raw_stream() -> int:
    """Handle of the current stream of the current device.  (torch.cuda.current_stream() builds a Stream object and
    re-validates the device on every call: a quarter of the host time of a scene before this replaced it.)"""
    dev = torch._C._cuda_getDevice()
    _check_thread(dev)
    return torch._C._cuda_getCurrentRawStream(dev)


def _stream():
    return ctypes.c_void_p(_raw_stream())


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _workspace(nbytes: int, device) -> torch.Tensor:
    """One growing scratch buffer per (device, stream); kernels on a stream run in order, so
    consecutive calls may share it."""
    key = ((device if isinstance(device, torch.device) else torch.device(device)).index, _raw_stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _need_cuda(*ts):
    """Every operator launches on the current stream of the CURRENT device: tensors must live there."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("segmentation_amd operators need tensors on an AMD GPU (cuda device); "
                               "there is no CPU path")
        if t.device.index != torch._C._cuda_getDevice():
            raise RuntimeError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                               "run under `with torch.cuda.device(tensor.device):` (one process per GPU is the "
                               "intended deployment)")


def _shape3(shape: Sequence[int]):
    arr = (ctypes.c_int64 * 3)(*[int(s) for s in shape])
    return arr


# ----------------------------------------------------------------------------- voxelize
_PINNED_META = {}


def voxelize(coord, feat, offset, grid_size, pad=128, hash_method="fnv", grid_mode="div",
             return_keys=False, return_grid_coord=False, pooling_method="mean", seed=0, defer=False):
    """Point.sparsify + mean_pooling / random_pooling on the GPU (reference: structure.py:197-251, :34-110).

    pooling_method="random" keeps the voxel set, order and v2p_map and takes every voxel's features from one of its
    points, drawn uniformly as a function of (seed, point index); the chosen points are returned as "choice".

    offset: cumulative counts, with or without the leading 0 (misc.py:8-13), tensor or list.
    Returns a dict with voxel_coords [N0,4] i32, voxel_feats [N0,C] f32, v2p_map [N] i64,
    spatial_shape (list of 3), batch_size, optional keys [N0] (uint64 bit patterns in int64).
    defer=True: the kernels are enqueued, the output sizes travel to pinned host memory asynchronously, and a
    function is returned that waits for that copy alone and yields the dict (mean pooling only): what the caller
    enqueues in between keeps the host busy while the voxelizer runs."""
    lib = _lib.load()
    _need_cuda(coord, feat)
    coord = coord.contiguous().float()
    feat = feat.contiguous().float()
    n, c = feat.shape
    assert coord.shape == (n, 3)
    off = torch.as_tensor(offset).detach().to("cpu", torch.int64).flatten()
    if off.numel() == 0:
        raise ValueError("offset must not be empty")
    if int(off[0]) != 0:
        off = torch.cat([torch.zeros(1, dtype=torch.int64), off])
    if int(off[-1]) != n:
        raise ValueError(f"offset[-1]={int(off[-1])} does not match the number of points {n}")
    counts = off[1:] - off[:-1]
    if (counts < 0).any():
        raise ValueError("offset must be non-decreasing")
    if n == 0:
        raise ValueError("voxelize needs at least one point")
    b = counts.numel()
    batch_size = int(torch.nonzero(counts > 0).max()) + 1
    dev = coord.device
    # through pinned memory, asynchronously: a pageable copy would hold the host until the stream has caught up
    off_pin = torch.empty(off.shape, dtype=torch.int64, pin_memory=True)
    off_pin.copy_(off)
    off_d = off_pin.to(dev, non_blocking=True)
    voxel_coords = torch.empty((n, 4), dtype=torch.int32, device=dev)
    voxel_feats = torch.empty((n, c), dtype=torch.float32, device=dev)
    v2p = torch.empty((n,), dtype=torch.int64, device=dev)
    keys = torch.empty((n,), dtype=torch.int64, device=dev) if return_keys else None
    gc = torch.empty((n, 3), dtype=torch.int32, device=dev) if return_grid_coord else None
    meta = torch.empty((16,), dtype=torch.int64, device=dev)
    nb = lib.m3d_voxelize_workspace_bytes(n, c)
    ws = _workspace(nb, dev)
    rc = lib.m3d_voxelize(_ptr(coord), _ptr(feat), _ptr(off_d), n, c, b, float(grid_size), int(pad),
                          HASH[hash_method], GRID[grid_mode], _ptr(voxel_coords), _ptr(voxel_feats),
                          _ptr(v2p), _ptr(keys), _ptr(gc), _ptr(meta), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "m3d_voxelize")

    def result(m):
        if m[12]:
            raise ValueError("voxelize: the grid coordinates span 2^31 cells or more along an axis "
                             f"(extent {[int(v) for v in m[7:10]]}): non-finite or out-of-range point coordinates")
        n0 = m[0]
        out = dict(voxel_coords=voxel_coords[:n0], voxel_feats=voxel_feats[:n0], v2p_map=v2p,
                   spatial_shape=[int(m[1]), int(m[2]), int(m[3])], batch_size=batch_size, n_voxels=n0,
                   grid_min=[int(m[4]), int(m[5]), int(m[6])])
        if return_keys:
            out["keys"] = keys[:n0]
        if return_grid_coord:
            out["grid_coord"] = gc
        return out

    if defer:
        if pooling_method != "mean":
            raise ValueError("voxelize(defer=True) supports mean pooling only")
        # one pinned buffer per call in flight: the caller may start the next scene before finishing this one
        slot = _PINNED_META.setdefault(dev.index, [])
        pinned = slot.pop() if slot else torch.empty((16,), dtype=torch.int64, pin_memory=True)
        pinned.copy_(meta, non_blocking=True)
        landed = torch.cuda.Event()
        landed.record()

        def finish():
            landed.synchronize()
            m = pinned.tolist()
            slot.append(pinned)
            return result(m)

        return finish
    out = result(meta.cpu().tolist())  # the one host sync of the voxelizer: output sizes are data dependent
    n0 = out["n_voxels"]
    if pooling_method == "random":
        choice = torch.empty((n0,), dtype=torch.int64, device=dev)
        ws2 = _workspace(lib.m3d_voxel_pick_random_workspace_bytes(n0), dev)
        rc = lib.m3d_voxel_pick_random(_ptr(v2p), _ptr(feat), n, c, n0, int(seed) & (2 ** 64 - 1), _ptr(choice),
                                       _ptr(voxel_feats), _ptr(ws2), ws2.numel(), _stream())
        _lib.check(rc, "m3d_voxel_pick_random")
        out["choice"] = choice
    elif pooling_method != "mean":
        raise ValueError(f"Unknown pooling method: {pooling_method}")
    return out


# ----------------------------------------------------------------------------- kernel maps
def subm_rulebook(indices, spatial_shape, batch_size=0):
    """SubM k3 kernel map: pair [27, N] i32 (reference call sites: v1m3:98-117,205-212).  batch_size (every batch
    index is below it) lets grids of up to 2^28 cells use the occupancy-bitmap lookup; 0 = unknown (hash table)."""
    lib = _lib.load()
    _need_cuda(indices)
    indices = indices.contiguous()
    assert indices.dtype == torch.int32 and indices.dim() == 2 and indices.shape[1] == 4
    n = indices.shape[0]
    pair = torch.empty((27, n), dtype=torch.int32, device=indices.device)
    if n == 0:
        return pair
    ws = _workspace(lib.m3d_subm_rulebook_workspace_bytes(n, _shape3(spatial_shape), int(batch_size)), indices.device)
    rc = lib.m3d_subm_rulebook(_ptr(indices), n, _shape3(spatial_shape), int(batch_size), _ptr(pair), _ptr(ws),
                               ws.numel(), _stream())
    _lib.check(rc, "m3d_subm_rulebook")
    return pair


_PINNED_COUNTS = {}


def _level_bounds(n0, spatial_shape, batch_size, n_levels):
    """Upper bound of the rows of every level before the counts are known: the rows of level 0, and the cells of the
    level's own grid (shape halved per level) when the batch size is known."""
    shape, out = [int(v) for v in spatial_shape], []
    for l in range(n_levels):
        cells = int(batch_size) * shape[0] * shape[1] * shape[2] if batch_size and batch_size > 0 else n0
        out.append(n0 if l == 0 else max(1, min(n0, cells)))
        shape = [d // 2 for d in shape]
    return out


def strided_chain(indices, spatial_shape, batch_size, depth, defer=False):
    """`depth` successive k2 s2 p0 kernel maps (v1m3:154-161), each on the previous one's outputs.  The
    output count of a level stays on the device and feeds the next level, so the whole chain is enqueued
    without a host read-back; the counts are read once at the end.  Returns a list of
    (out_indices [M,4], out_shape, pair_fwd [8,M] (a strided view), pair_bwd [8,N]).
    With defer=True the counts travel to pinned host memory asynchronously and a function is returned that
    waits for that copy alone and yields the list: work enqueued in between keeps the GPU busy meanwhile."""
    lib = _lib.load()
    _need_cuda(indices)
    indices = indices.contiguous()
    assert indices.dtype == torch.int32 and indices.dim() == 2 and indices.shape[1] == 4
    n = indices.shape[0]
    if n == 0:
        raise ValueError("strided_rulebook needs at least one voxel")
    dev = indices.device
    counts = torch.empty((depth,), dtype=torch.int64, device=dev)
    bufs, shape, cur = [], [int(s) for s in spatial_shape], indices
    cap = _level_bounds(n, spatial_shape, batch_size, depth + 1)
    for l in range(depth):
        out_idx = torch.empty((cap[l + 1], 4), dtype=torch.int32, device=dev)
        pf = torch.empty((8, cap[l]), dtype=torch.int32, device=dev)
        pb = torch.empty((8, cap[l]), dtype=torch.int32, device=dev)
        ws = _workspace(lib.m3d_strided_rulebook_workspace_bytes(cap[l], _shape3(shape), int(batch_size)), dev)
        d_n_rows = counts[l - 1:l].data_ptr() if l else None
        rc = lib.m3d_strided_rulebook(_ptr(cur), cap[l], d_n_rows, _shape3(shape), int(batch_size), _ptr(out_idx), _ptr(pf),
                                      _ptr(pb), counts[l:l + 1].data_ptr(), _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "m3d_strided_rulebook")
        shape = [s // 2 for s in shape]
        bufs.append((out_idx, list(shape), pf, pb))
        cur = out_idx
    def slices(m):
        out, n_in = [], n
        for l, (out_idx, oshape, pf, pb) in enumerate(bufs):
            if m[l] == 0:
                raise ValueError(f"strided_rulebook: level {l} has no output voxel")
            out.append((out_idx[:m[l]], oshape, pf[:, :m[l]], pb[:, :n_in]))
            n_in = m[l]
        return out

    if not defer:
        return slices(counts.cpu().tolist())  # the one host sync of the chain
    slot = _PINNED_COUNTS.setdefault((dev.index, depth), [])  # one pinned buffer per call in flight
    pinned = slot.pop() if slot else torch.empty((depth,), dtype=torch.int64, pin_memory=True)
    pinned.copy_(counts, non_blocking=True)
    landed = torch.cuda.Event()
    landed.record()

    def finish():
        landed.synchronize()
        m = pinned.tolist()
        slot.append(pinned)
        return slices(m)

    return finish


def strided_rulebook(indices, spatial_shape, batch_size):
    """k2 s2 p0 kernel map (v1m3:154-161).  Returns out_indices [M,4], out_shape, pair_fwd [8,M]
    (a strided view), pair_bwd [8,N]."""
    return strided_chain(indices, spatial_shape, batch_size, 1)[0]


@dataclass
class Schedule:
    """Tile schedule of a kernel map (see include/m3d.h, m3d_build_schedule)."""
    nbr: torch.Tensor        # [k, npad] i32
    out_row: torch.Tensor    # [npad] i32
    tile_mask: torch.Tensor  # [npad/32] i32 (bit k = offset k active in the tile)
    n: int
    npad: int
    k: int
    tile_order: object = None  # optional [npad/128] i32: 128-row tiles, heaviest first (m3d_schedule_tile_order)


class ArenaSchedule:
    """A Schedule whose arrays are slices of one int32 buffer (the level-chain entry points write all maps and
    schedules of a scene into two such buffers): the device addresses are at hand without creating a tensor per
    array; the tensors are made on request."""
    __slots__ = ("buf", "o_nbr", "o_row", "o_mask", "o_order", "n", "npad", "k", "_npairs")

    def __init__(self, buf, o_nbr, o_row, o_mask, o_order, n, npad, k):
        self.buf, self.o_nbr, self.o_row, self.o_mask, self.o_order = buf, o_nbr, o_row, o_mask, o_order
        self.n, self.npad, self.k, self._npairs = n, npad, k, None

    def ptr(self, off):
        return None if off < 0 else self.buf.data_ptr() + 4 * off

    @property
    def nbr(self):
        return self.buf[self.o_nbr:self.o_nbr + self.k * self.npad].view(self.k, self.npad)

    @property
    def out_row(self):
        return self.buf[self.o_row:self.o_row + self.npad]

    @property
    def tile_mask(self):
        return self.buf[self.o_mask:self.o_mask + self.npad // 32]

    @property
    def tile_order(self):
        return None if self.o_order < 0 else self.buf[self.o_order:self.o_order + self.npad // 128]


class _Carver:
    """Offsets (in int32 elements, 256-byte aligned) of consecutive arrays inside one buffer."""

    def __init__(self):
        self.off = 0

    def take(self, count):
        o = self.off
        self.off += (int(count) + 63) // 64 * 64
        return o


def _fork(streams):
    """The given streams continue from where the current stream stands."""
    cur = torch.cuda.current_stream()
    for st in streams:
        st.wait_stream(cur)


def _join(streams):
    """The current stream continues only after everything enqueued on the given streams."""
    cur = torch.cuda.current_stream()
    for st in streams:
        cur.wait_stream(st)


def level_chain_begin(indices, spatial_shape, batch_size, depth, subm_modes, tile_order=True, aux_streams=()):
    """Kernel maps and schedules of all levels through m3d_levels_begin / m3d_levels_finish (two library calls per
    scene).  Enqueues the chain of strided maps and level 0's submanifold map + schedule, starts the copy of the
    level counts to pinned memory and returns finish(): it waits for that copy alone, enqueues the rest and returns
    a list of per-level dicts (indices, shape, n, pair_subm, subm; and, except on the last level, pair_fwd, pair_bwd,
    next_indices, next_shape, n_next, down, up) -- the same maps and schedules, bit for bit, as subm_rulebook /
    strided_chain / build_schedule give level by level.
    aux_streams: streams the independent parts may run on beside the current one (the chain of strided maps and
    level 0's map in the first call, the levels in the second); everything is joined back into the current stream
    before finish() returns.  (Measured on the benchmark scene with three extra streams: the maps stage takes 0.96 ms
    instead of 0.68 ms -- the cross-stream dependencies cost more than the overlap of these small launches gains --
    so the engine does not use it.)"""
    lib = _lib.load()
    _need_cuda(indices)
    indices = indices.contiguous()
    assert indices.dtype == torch.int32 and indices.dim() == 2 and indices.shape[1] == 4
    n0, dev, nl = indices.shape[0], indices.device, depth + 1
    if n0 == 0:
        raise ValueError("level_chain needs at least one voxel")
    npad0 = lib.m3d_schedule_npad(n0)
    cap = _level_bounds(n0, spatial_shape, batch_size, nl)   # rows a level can have before its count is known
    c = _Carver()
    o_next = [c.take(4 * cap[l + 1]) for l in range(depth)]
    o_pf = [c.take(8 * cap[l]) for l in range(depth)]
    o_pb = [c.take(8 * cap[l]) for l in range(depth)]
    o_pair0, o_nbr0, o_row0, o_mask0 = c.take(27 * n0), c.take(27 * npad0), c.take(npad0), c.take(npad0 // 32)
    o_ord0 = c.take(npad0 // 128) if tile_order else -1
    buf = torch.empty((c.off,), dtype=torch.int32, device=dev)
    base = buf.data_ptr()
    counts = torch.empty((max(depth, 1),), dtype=torch.int64, device=dev)
    L = (_lib.Level * nl)()
    shape = [int(v) for v in spatial_shape]
    shapes = []
    for l in range(nl):
        v = L[l]
        shapes.append(list(shape))
        v.indices = indices.data_ptr() if l == 0 else base + 4 * o_next[l - 1]
        v.n = n0 if l == 0 else 0
        v.shape[0], v.shape[1], v.shape[2] = shape
        v.subm_mode = int(subm_modes[min(l, len(subm_modes) - 1)])
        if l < depth:
            v.next_indices, v.pair_fwd, v.pair_bwd = base + 4 * o_next[l], base + 4 * o_pf[l], base + 4 * o_pb[l]
            v.ld = cap[l]
        shape = [d // 2 for d in shape]
    v = L[0]
    v.pair_subm = base + 4 * o_pair0
    v.subm.nbr, v.subm.out_row, v.subm.tile_mask = base + 4 * o_nbr0, base + 4 * o_row0, base + 4 * o_mask0
    v.subm.tile_order = None if o_ord0 < 0 else base + 4 * o_ord0
    aux = list(aux_streams)
    ws_bytes = lib.m3d_levels_workspace_bytes(L, nl, int(batch_size))
    if aux and depth:
        _fork(aux[:1])
        with torch.cuda.stream(aux[0]):   # level 0's map and schedule beside the chain
            ws = _workspace(ws_bytes, dev)
            _lib.check(lib.m3d_levels_begin(L, nl, int(batch_size), 2, None, _ptr(ws), ws.numel(), _stream()),
                       "m3d_levels_begin")
        ws = _workspace(ws_bytes, dev)
        _lib.check(lib.m3d_levels_begin(L, nl, int(batch_size), 1, _ptr(counts), _ptr(ws), ws.numel(), _stream()),
                   "m3d_levels_begin")
    else:
        ws = _workspace(ws_bytes, dev)
        rc = lib.m3d_levels_begin(L, nl, int(batch_size), 3, _ptr(counts), _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "m3d_levels_begin")
    slot = _PINNED_COUNTS.setdefault((dev.index, max(depth, 1)), [])  # one pinned buffer per call in flight
    pinned = slot.pop() if slot else torch.empty((max(depth, 1),), dtype=torch.int64, pin_memory=True)
    if depth:
        pinned.copy_(counts, non_blocking=True)
    landed = torch.cuda.Event()
    landed.record()

    def finish():
        landed.synchronize()
        m = pinned.tolist() if depth else []
        slot.append(pinned)
        ns = [n0] + [int(x) for x in m[:depth]]
        for l in range(1, nl):
            if ns[l] <= 0:
                raise ValueError(f"strided_rulebook: level {l - 1} has no output voxel")
        npad = [lib.m3d_schedule_npad(n) for n in ns]
        c2 = _Carver()
        o_pair, o_sub, o_dn, o_up = {}, {0: (o_nbr0, o_row0, o_mask0, o_ord0)}, {}, {}
        for l in range(nl):
            if l >= 1:
                o_pair[l] = c2.take(27 * ns[l])
                o_sub[l] = (c2.take(27 * npad[l]), c2.take(npad[l]), c2.take(npad[l] // 32),
                            c2.take(npad[l] // 128) if tile_order else -1)
            if l < depth:
                o_dn[l] = (c2.take(8 * npad[l + 1]), c2.take(npad[l + 1]), c2.take(npad[l + 1] // 32))
                o_up[l] = (c2.take(8 * npad[l]), c2.take(npad[l]), c2.take(npad[l] // 32))
        buf2 = torch.empty((max(c2.off, 64),), dtype=torch.int32, device=dev)
        b2 = buf2.data_ptr()
        for l in range(nl):
            v = L[l]
            v.n = ns[l]
            if l >= 1:
                v.pair_subm = b2 + 4 * o_pair[l]
                on, orow, om, oo = o_sub[l]
                v.subm.nbr, v.subm.out_row, v.subm.tile_mask = b2 + 4 * on, b2 + 4 * orow, b2 + 4 * om
                v.subm.tile_order = None if oo < 0 else b2 + 4 * oo
            if l < depth:
                on, orow, om = o_dn[l]
                v.down.nbr, v.down.out_row, v.down.tile_mask = b2 + 4 * on, b2 + 4 * orow, b2 + 4 * om
                on, orow, om = o_up[l]
                v.up.nbr, v.up.out_row, v.up.tile_mask = b2 + 4 * on, b2 + 4 * orow, b2 + 4 * om
        ws_bytes2 = lib.m3d_levels_workspace_bytes(L, nl, int(batch_size))
        if aux and depth:
            # level l on stream l mod (1 + len(aux)); level 0's part follows its map on aux[0]
            lanes = [aux[0]] + [None] + aux[1:]           # lane 0 = aux[0] (level 0), lane 1 = the current stream
            masks = [0] * len(lanes)
            masks[0] |= 1
            for l in range(1, nl):
                masks[1 + (l - 1) % (len(lanes) - 1)] |= 1 << l
            _fork(aux)   # (aux[0] too: level 0's strided maps come from the chain on the current stream)
            for st, mask in zip(lanes, masks):
                if not mask:
                    continue
                if st is None:
                    ws2 = _workspace(ws_bytes2, dev)
                    _lib.check(lib.m3d_levels_finish(L, nl, int(batch_size), mask, _ptr(ws2), ws2.numel(), _stream()),
                               "m3d_levels_finish")
                else:
                    with torch.cuda.stream(st):
                        ws2 = _workspace(ws_bytes2, dev)
                        _lib.check(lib.m3d_levels_finish(L, nl, int(batch_size), mask, _ptr(ws2), ws2.numel(), _stream()),
                                   "m3d_levels_finish")
            _join(aux)
        else:
            ws2 = _workspace(ws_bytes2, dev)
            rc = lib.m3d_levels_finish(L, nl, int(batch_size), (1 << nl) - 1, _ptr(ws2), ws2.numel(), _stream())
            _lib.check(rc, "m3d_levels_finish")
        out = []
        for l in range(nl):
            n = ns[l]
            d = dict(shape=shapes[l], n=n,
                     indices=indices if l == 0 else buf[o_next[l - 1]:o_next[l - 1] + 4 * n].view(n, 4))
            if l == 0:
                d["pair_subm"] = buf[o_pair0:o_pair0 + 27 * n].view(27, n)
                d["subm"] = ArenaSchedule(buf, o_nbr0, o_row0, o_mask0, o_ord0, n, npad[0], 27)
            else:
                d["pair_subm"] = buf2[o_pair[l]:o_pair[l] + 27 * n].view(27, n)
                d["subm"] = ArenaSchedule(buf2, *o_sub[l], n, npad[l], 27)
            if l < depth:
                m_next = ns[l + 1]
                d["pair_fwd"] = buf[o_pf[l]:o_pf[l] + 8 * cap[l]].view(8, cap[l])[:, :m_next]
                d["pair_bwd"] = buf[o_pb[l]:o_pb[l] + 8 * cap[l]].view(8, cap[l])[:, :n]
                d["next_indices"] = buf[o_next[l]:o_next[l] + 4 * m_next].view(m_next, 4)
                d["next_shape"], d["n_next"] = shapes[l + 1], m_next
                d["down"] = ArenaSchedule(buf2, *o_dn[l], -1, m_next, npad[l + 1], 8)
                d["up"] = ArenaSchedule(buf2, *o_up[l], -1, n, npad[l], 8)
            out.append(d)
        return out

    return finish


def build_schedule(pair, n, indices=None, mode=0, spatial_shape=None, batch_size=1, base_order=None,
                   tile_order=False):
    lib = _lib.load()
    _need_cuda(pair)
    assert pair.dtype == torch.int32 and pair.dim() == 2
    k = pair.shape[0]
    ld = pair.stride(0)
    assert pair.stride(1) == 1 and ld >= n
    dev = pair.device
    npad = lib.m3d_schedule_npad(n)
    nbr = torch.empty((k, npad), dtype=torch.int32, device=dev)
    out_row = torch.empty((npad,), dtype=torch.int32, device=dev)
    mask = torch.empty((npad // 32,), dtype=torch.int32, device=dev)
    if indices is not None:
        indices = indices.contiguous()
        assert indices.dtype == torch.int32 and indices.shape == (n, 4)
    ws = _workspace(lib.m3d_schedule_workspace_bytes(n), dev)
    if base_order is not None:
        assert base_order.dtype == torch.int32 and base_order.is_contiguous() and base_order.numel() >= n
    shp = _shape3(spatial_shape) if spatial_shape is not None else None
    rc = lib.m3d_build_schedule(_ptr(pair), k, ld, n, _ptr(indices), shp, int(batch_size), int(mode),
                                _ptr(base_order), _ptr(nbr), _ptr(out_row), _ptr(mask), _ptr(ws), ws.numel(),
                                _stream())
    _lib.check(rc, "m3d_build_schedule")
    sched = Schedule(nbr, out_row, mask, n, npad, k)
    if tile_order:
        sched.tile_order = torch.empty((npad // 128,), dtype=torch.int32, device=dev)
        rc = lib.m3d_schedule_tile_order(_ptr(mask), npad, _ptr(sched.tile_order), _stream())
        _lib.check(rc, "m3d_schedule_tile_order")
    return sched


# ----------------------------------------------------------------------------- convolution
PRECISION = {"f32": 0, "bf16x3": 1, "bf16x6": 2, "bf16": 3, "f16x3": 4}
# split-format code each precision consumes: the number of bf16 terms, or 4 = the f16 pair (M3D_FMT_F16_PAIR)
TERMS = {"f32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3, "f16x3": 4}
PRECISION_OF_TERMS = {1: "bf16", 2: "bf16x3", 3: "bf16x6", 4: "f16x3"}
FMT_F16_PAIR = 4


def f16_status(clear=True):
    """True if a kernel on the current device wrote a value beyond the f16 range into the f16 pair format since the
    status was last cleared (m3d_f16_status; waits for the current stream).  Results of precision "f16x3" computed
    while it was set are not to be trusted."""
    import ctypes
    lib = _lib.load()
    v = ctypes.c_int(0)
    rc = lib.m3d_f16_status(ctypes.byref(v), int(bool(clear)), _stream())
    _lib.check(rc, "m3d_f16_status")
    return bool(v.value)


@dataclass
class PackedWeight:
    """A convolution weight in the kernel's layout for one precision (see m3d_conv_pack_weight)."""
    data: torch.Tensor  # opaque bytes on the device
    k: int
    cout: int
    cin: int
    precision: str
    # "f16x3": the weight was multiplied by 2**exp before packing (largest magnitude near 2**14, the whole tensor
    # inside the f16 normal range); conv_fwd and head_fused multiply the result by unscale = 2**-exp again
    exp: int = 0
    unscale_vec: object = None  # [cout] f32 filled with 2**-exp (made on first use by a call without a scale vector)

    @property
    def unscale(self):
        return 2.0 ** -self.exp

    def scale_with_unscale(self, scale):
        """The per-channel scale of the epilogue with the weight's power of two folded in."""
        if self.exp == 0:
            return scale
        if scale is None:
            if self.unscale_vec is None:
                self.unscale_vec = torch.full((self.cout,), self.unscale, dtype=torch.float32, device=self.data.device)
            return self.unscale_vec
        return scale * self.unscale


def pack_weight(weight, precision="f32", device=None, transposed=False, flip=False):
    """Reference layout [Cout, k0, k1, k2, Cin] (or [Cout, K, Cin]) f32 -> PackedWeight on `device`.
    transposed=True packs the weight of the layer's input-gradient convolution instead, W'[ci][k'][co] =
    W[co][K-1-k' if flip else k'][ci] (a PackedWeight with Cin outputs and Cout inputs), in the same single pass."""
    lib = _lib.load()
    w = weight.detach().float()
    cout, cin = w.shape[0], w.shape[-1]
    w = w.reshape(cout, -1, cin).contiguous()
    k = w.shape[1]
    if device is not None:
        w = w.to(device)
    _need_cuda(w)
    prec = PRECISION[precision]
    exp = 0
    if precision == "f16x3" and cin % 32 == 0 and cout % 32 == 0:
        # an exact power-of-two rescale into the upper part of the f16 range (reads the maximum back: packing a
        # weight for this precision synchronises, which is why it is for inference engines, packed once)
        wmax = float(w.abs().max())
        if wmax > 0 and np.isfinite(wmax):
            exp = int(np.clip(14 - int(np.ceil(np.log2(wmax))), -60, 60))
            w = w * (2.0 ** exp)
    if transposed:
        nbytes = lib.m3d_conv_packed_weight_bytes(cin, k, cout, prec)
        out = torch.empty((nbytes,), dtype=torch.uint8, device=w.device)
        rc = lib.m3d_conv_pack_weight_transposed(_ptr(w), cout, k, cin, int(bool(flip)), prec, _ptr(out), _stream())
        _lib.check(rc, "m3d_conv_pack_weight_transposed")
        return PackedWeight(out, k, cin, cout, precision, exp)
    nbytes = lib.m3d_conv_packed_weight_bytes(cout, k, cin, prec)
    out = torch.empty((nbytes,), dtype=torch.uint8, device=w.device)
    rc = lib.m3d_conv_pack_weight(_ptr(w), cout, k, cin, prec, _ptr(out), _stream())
    _lib.check(rc, "m3d_conv_pack_weight")
    return PackedWeight(out, k, cout, cin, precision, exp)


@dataclass
class SplitTensor:
    """An f32 [n, c] tensor in the engine-internal split format "sf32" (see include/m3d.h):
    [n + 1][c/G][terms][G] bf16 (G = 16 for three terms, else 32; one term = plain bf16 rows), row n all zero."""
    data: torch.Tensor  # uint8
    n: int
    c: int
    terms: int

    @property
    def shape(self):
        return (self.n, self.c)


def split_f32(x, terms):
    lib = _lib.load()
    _need_cuda(x)
    x = x.contiguous().float()
    n, c = x.shape
    out = torch.empty((lib.m3d_sf32_bytes(n, c, terms),), dtype=torch.uint8, device=x.device)
    rc = lib.m3d_split_f32(_ptr(x), n, c, terms, _ptr(out), _stream())
    _lib.check(rc, "m3d_split_f32")
    return SplitTensor(out, n, c, terms)


def unsplit(st: SplitTensor):
    """sf32 -> f32 (sum of the terms); for tests and debugging."""
    if st.terms == FMT_F16_PAIR:
        t = st.data.view(torch.float16).view(st.n + 1, st.c // 32, 2, 32).float()
        v = t[:, :, 0] + t[:, :, 1] * (1.0 / 2048.0)
        return v.reshape(st.n + 1, st.c)[:st.n].contiguous()
    g = 16 if st.terms == 3 else 32
    v = st.data.view(torch.bfloat16).view(st.n + 1, st.c // g, st.terms, g).float().sum(2)
    return v.reshape(st.n + 1, st.c)[:st.n]


def conv_fwd(in0, weight_packed: PackedWeight, sched: Optional[Schedule], n_out, in1=None, scale=None, shift=None,
             residual=None, relu=False, out=None, flop_cout=None, want_f32=True, split_terms=0):
    """out[row] = act(scale * sum_k W[k] . cat(in0, in1)[nbr[k][t]] + shift + residual[row]).

    sched=None is the dense 1x1 case (Custom1x1Subm3d, structure.py:336-347).  in0/in1 are f32 tensors or
    SplitTensors (then the LDS-DMA kernel runs; the weight must be packed for the matching precision).
    Returns the f32 output, or (f32 output or None, SplitTensor) when split_terms is 1, 2 or 3."""
    lib = _lib.load()
    sf_in = isinstance(in0, SplitTensor)
    k, cout, cin = weight_packed.k, weight_packed.cout, weight_packed.cin
    c0 = in0.shape[1]
    c1 = 0 if in1 is None else in1.shape[1]
    if c0 + c1 != cin:
        raise ValueError(f"input channels {c0}+{c1} do not match the weight's {cin}")
    if sf_in:
        want = PRECISION_OF_TERMS[in0.terms]
        if weight_packed.precision != want:
            raise ValueError(f"sf32 input with {in0.terms} terms needs a weight packed for {want}")
        if in1 is not None and not (isinstance(in1, SplitTensor) and in1.terms == in0.terms):
            raise ValueError("both inputs must use the same split format")
        dev = in0.data.device
        _need_cuda(in0.data, weight_packed.data)
    else:
        _need_cuda(in0, weight_packed.data)
        assert in0.is_contiguous() and in0.dtype == torch.float32
        assert in1 is None or (in1.is_contiguous() and in1.dtype == torch.float32)
        dev = in0.device
    if split_terms not in (0, 1, 2, 3, 4):
        raise ValueError("split_terms must be 0, 1, 2, 3 or 4 (the f16 pair)")
    if sf_in and split_terms and split_terms != in0.terms:
        raise ValueError("the split output of an sf32 convolution uses the input's number of terms")
    if not want_f32 and split_terms == 0:
        raise ValueError("nothing to compute: want_f32=False and split_terms=0")
    scale = weight_packed.scale_with_unscale(scale)
    if out is None and want_f32:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    if out is not None:
        assert out.dtype == torch.float32 and out.shape[0] == n_out and out.shape[1] >= cout and out.stride(1) == 1
    ldo = out.stride(0) if out is not None else cout
    out_sp = None
    if split_terms:
        out_sp = SplitTensor(torch.empty((lib.m3d_sf32_bytes(n_out, cout, split_terms),), dtype=torch.uint8, device=dev),
                             n_out, cout, split_terms)
    for v in (scale, shift):
        assert v is None or (v.is_contiguous() and v.dtype == torch.float32 and v.numel() == cout)
    if residual is not None:
        assert residual.is_contiguous() and residual.shape == (n_out, cout) and residual.dtype == torch.float32
    if n_out == 0:
        return out if not split_terms else (out, out_sp)
    if sched is None:
        if k != 1:
            raise ValueError("a convolution without a schedule must have one kernel offset")
        if in0.shape[0] < n_out:
            raise ValueError("dense conv: fewer input rows than output rows")
        nbr = orow = mask = None
        npad = lib.m3d_schedule_npad(n_out)
    else:
        if sched.k != k:
            raise ValueError(f"schedule has {sched.k} kernel offsets, weight has {k}")
        if sched.n != n_out:
            raise ValueError(f"schedule covers {sched.n} rows, output has {n_out}")
        nbr, orow, mask, npad = sched.nbr, sched.out_row, sched.tile_mask, sched.npad
    ws = _workspace(lib.m3d_conv_workspace_bytes(n_out, cout, k), dev)
    prof = PROFILE
    if prof is not None:
        if sched is None:
            npairs = n_out
        else:
            if getattr(sched, "_npairs", None) is None:
                sched._npairs = int((sched.nbr >= 0).sum().item())
            npairs = sched._npairs
        mfma = c0 % 32 == 0 and c1 % 32 == 0 and cout % 32 == 0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    osp = None if out_sp is None else out_sp.data
    if sf_in:
        rc = lib.m3d_conv_fwd_sf32(_ptr(in0.data), c0, in0.n, _ptr(None if in1 is None else in1.data), c1,
                                   0 if in1 is None else in1.n, _ptr(weight_packed.data), k, cout, _ptr(nbr),
                                   _ptr(orow), _ptr(mask), n_out, npad,
                                   _ptr(None if sched is None else sched.tile_order),
                                   _ptr(scale), _ptr(shift), _ptr(residual),
                                   int(bool(relu)), _ptr(out), ldo, _ptr(osp), in0.terms,
                                   _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "m3d_conv_fwd_sf32")
    else:
        rc = lib.m3d_conv_fwd_f32(_ptr(in0), c0, in0.shape[0], _ptr(in1), c1, 0 if in1 is None else in1.shape[0],
                                  _ptr(weight_packed.data), k, cout, _ptr(nbr),
                                  _ptr(orow), _ptr(mask), n_out, npad,
                                  _ptr(None if sched is None else sched.tile_order), _ptr(scale), _ptr(shift),
                                  _ptr(residual), int(bool(relu)), _ptr(out), ldo, _ptr(osp), split_terms,
                                  PRECISION[weight_packed.precision], _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "m3d_conv_fwd_f32")
    if prof is not None:
        ev1.record()
        kind = ("dma" if sf_in else "igemm") if mfma else "direct"
        # algorithmic bytes of the launch: activations in and out, the weights and the kernel map, each once
        bpe = 4 if not sf_in else 2 * (2 if in0.terms == FMT_F16_PAIR else in0.terms)
        n_in0 = in0.n if sf_in else in0.shape[0]
        n_in1 = 0 if in1 is None else (in1.n if sf_in else in1.shape[0])
        obytes = (cout * 4 if out is not None else 0) + \
                 (cout * 2 * (2 if split_terms == FMT_F16_PAIR else split_terms) if split_terms else 0)
        abytes = (n_in0 * c0 + n_in1 * c1) * bpe + n_out * obytes + k * cout * cin * bpe + \
                 (0 if sched is None else k * npad * 4) + (n_out * cout * 4 if residual is not None else 0)
        prof.append((kind, 2.0 * npairs * cin * (flop_cout or cout), ev0, ev1, (n_out, cin, cout, k), abytes))
    return out if not split_terms else (out, out_sp)


# ----------------------------------------------------------------------------- head
def conv_bwd_weight(x, dout, sched, k, precision="f32"):
    """Weight gradient of a sparse convolution (m3d_conv_bwd_weight_f32 / _bf16x3): x [n_in, Cin] f32 the layer's
    input, dout [n_out, Cout] f32 the gradient of its output, sched the schedule the forward ran on (None = dense
    1x1), precision "f32" (exact products) or "bf16x3" (two-term bf16 splits, three products).
    Returns dW [Cout, k, Cin] f32.  Channel counts that are not multiples of 32 (the 3-channel stem) are
    zero-padded for the kernel and the padding is dropped from the result."""
    lib = _lib.load()
    _need_cuda(x, dout)
    n, cout = dout.shape
    cin = x.shape[1]
    pin, pout = (-cin) % 32, (-cout) % 32
    xk = torch.nn.functional.pad(x.float(), (0, pin)).contiguous() if pin else x.float().contiguous()
    dk = torch.nn.functional.pad(dout.float(), (0, pout)).contiguous() if pout else dout.float().contiguous()
    ci, co = cin + pin, cout + pout
    dev = x.device
    dw = torch.empty((co, k, ci), dtype=torch.float32, device=dev)
    if n == 0:
        return dw.zero_()[:cout, :, :cin]
    if sched is None:
        if k != 1:
            raise ValueError("a convolution without a schedule must have one kernel offset")
        nbr = orow = mask = None
        npad = lib.m3d_schedule_npad(n)
    else:
        if sched.k != k or sched.n != n:
            raise ValueError("schedule does not match the output gradient")
        nbr, orow, mask, npad = sched.nbr, sched.out_row, sched.tile_mask, sched.npad
    ws = _workspace(lib.m3d_conv_bwd_weight_workspace_bytes(n, ci, co, k), dev)
    if precision not in ("f32", "bf16x3"):
        raise ValueError(f"conv_bwd_weight: precision {precision!r} (f32 or bf16x3)")
    name = "m3d_conv_bwd_weight_" + precision
    rc = getattr(lib, name)(_ptr(xk), ci, _ptr(dk), co, k, _ptr(nbr), _ptr(orow), _ptr(mask), n, npad,
                            _ptr(dw), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, name)
    return dw[:cout, :, :cin] if (pin or pout) else dw


def bn_stats(x):
    """Per-channel mean and biased variance of x [n, c] f32 (m3d_bn_stats); c a multiple of 4."""
    lib = _lib.load()
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    n, c = x.shape
    mean = torch.empty((c,), dtype=torch.float32, device=x.device)
    var = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = _workspace(lib.m3d_bn_workspace_bytes(n, c), x.device)
    rc = lib.m3d_bn_stats(_ptr(x), n, c, _ptr(mean), _ptr(var), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "m3d_bn_stats")
    return mean, var


def bn_backward_reduce(x, dy, mean):
    """sum_dy[c], sum_dy_xmu[c] = column sums of dy and of dy * (x - mean) (m3d_bn_backward_reduce)."""
    lib = _lib.load()
    _need_cuda(x, dy, mean)
    assert x.dtype == dy.dtype == mean.dtype == torch.float32 and x.is_contiguous() and dy.is_contiguous()
    assert x.shape == dy.shape and mean.numel() == x.shape[1]
    n, c = x.shape
    sum_dy = torch.empty((c,), dtype=torch.float32, device=x.device)
    sum_dy_xmu = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = _workspace(lib.m3d_bn_workspace_bytes(n, c), x.device)
    rc = lib.m3d_bn_backward_reduce(_ptr(x), _ptr(dy), _ptr(mean), n, c, _ptr(sum_dy), _ptr(sum_dy_xmu), _ptr(ws),
                                    ws.numel(), _stream())
    _lib.check(rc, "m3d_bn_backward_reduce")
    return sum_dy, sum_dy_xmu


def bn_act_fwd(x, mean, var, eps, weight, bias, residual=None, relu=True):
    """y = relu?(batch_norm(x; mean, biased var, eps, weight, bias) + residual) in one pass (m3d_bn_act_fwd)."""
    lib = _lib.load()
    _need_cuda(x, mean, var, weight, bias, residual)
    n, c = x.shape
    y = torch.empty_like(x)
    rc = lib.m3d_bn_act_fwd(_ptr(x), n, c, _ptr(mean), _ptr(var), float(eps), _ptr(weight), _ptr(bias), _ptr(residual),
                            int(bool(relu)), _ptr(y), _stream())
    _lib.check(rc, "m3d_bn_act_fwd")
    return y


def bn_act_backward(x, dy, y, mean, var, eps, weight, want_dres):
    """Backward of bn_act_fwd: returns (dx, dres or None, sum_g, sum_g_xmu); y = the forward output when a ReLU was
    applied, else None."""
    lib = _lib.load()
    _need_cuda(x, dy, y, mean, var, weight)
    n, c = x.shape
    sum_g = torch.empty((c,), dtype=torch.float32, device=x.device)
    sum_g_xmu = torch.empty((c,), dtype=torch.float32, device=x.device)
    ws = _workspace(lib.m3d_bn_workspace_bytes(n, c), x.device)
    rc = lib.m3d_bn_act_backward_reduce(_ptr(x), _ptr(dy), _ptr(y), _ptr(mean), n, c, _ptr(sum_g), _ptr(sum_g_xmu),
                                        _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "m3d_bn_act_backward_reduce")
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    rc = lib.m3d_bn_act_backward(_ptr(x), _ptr(dy), _ptr(y), n, c, _ptr(mean), _ptr(var), float(eps), _ptr(weight),
                                 _ptr(sum_g), _ptr(sum_g_xmu), _ptr(dx), _ptr(dres), _stream())
    _lib.check(rc, "m3d_bn_act_backward")
    return dx, dres, sum_g, sum_g_xmu


def argmax_rows(logits, n_classes):
    lib = _lib.load()
    _need_cuda(logits)
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    n = logits.shape[0]
    label = torch.empty((n,), dtype=torch.int32, device=logits.device)
    rc = lib.m3d_argmax_rows(_ptr(logits), n, int(n_classes), logits.stride(0), _ptr(label), _stream())
    _lib.check(rc, "m3d_argmax_rows")
    return label


class PackedText:
    """Text embeddings in the layout of the fused head kernel."""

    def __init__(self, data, n_classes, classes_padded, cmid, terms=2):
        self.data, self.n_classes, self.classes_padded, self.cmid, self.terms = data, n_classes, classes_padded, cmid, terms


def head_pack_text(text, cin=96, classes_padded=None, terms=2):
    """text [n_classes, cmid] f32 -> PackedText in the split format of `terms` (2 or 3) terms.  The class count is
    padded to what the fused kernel for `cin` input channels takes: a multiple of 32 for 96 channels, 128 or 256
    otherwise."""
    lib = _lib.load()
    _need_cuda(text)
    text = text.contiguous().float()
    c, d = text.shape
    cp = classes_padded or ((c + 31) // 32 * 32 if cin == 96 else (128 if c <= 128 else 256))
    if c > cp or d % 32 != 0:
        raise ValueError(f"head_pack_text: {c} classes x {d} channels not supported")
    if terms not in (2, 3, 4):
        raise ValueError("head_pack_text: the fused head exists for 2 and 3 bf16 terms and the f16 pair (4)")
    out = torch.empty((lib.m3d_head_text_bytes_terms(cp, d, terms),), dtype=torch.uint8, device=text.device)
    rc = lib.m3d_head_pack_text_terms(_ptr(text), c, d, cp, terms, _ptr(out), _stream())
    _lib.check(rc, "m3d_head_pack_text_terms")
    return PackedText(out, c, cp, d, terms)


def head_fused_supported(cin, n_classes):
    return cin in (32, 64, 96, 128) and n_classes <= 256


def head_fused(x: "SplitTensor", w1: "PackedWeight", text: PackedText, scale=None, shift=None, want_logits=False,
               want_labels=True):
    """Last 1x1 layer + text logits + row argmax in one kernel.  Returns (voxel_label or None, logits or None)."""
    lib = _lib.load()
    assert x.terms in (2, 3, 4) and x.terms == text.terms and w1.precision == PRECISION_OF_TERMS[x.terms]
    scale = w1.scale_with_unscale(scale)
    assert w1.k == 1 and w1.cout == text.cmid and w1.cin == x.c
    n, dev = x.n, x.data.device
    label = torch.empty((n,), dtype=torch.int32, device=dev) if want_labels else None
    logits = torch.empty((n, text.classes_padded), dtype=torch.float32, device=dev) if want_logits else None
    ev = None
    if PROFILE is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    rc = lib.m3d_head_fused_terms(_ptr(x.data), x.c, n, x.terms, _ptr(w1.data), _ptr(scale), _ptr(shift), text.cmid,
                                  _ptr(text.data), text.classes_padded, text.n_classes, _ptr(logits),
                                  text.classes_padded, _ptr(label), _stream())
    _lib.check(rc, "m3d_head_fused_terms")
    if ev is not None:
        ev[1].record()
        flops = 2.0 * n * text.cmid * (x.c + text.n_classes)
        PROFILE.append(("head", flops, ev[0], ev[1], (n, x.c, text.n_classes, 1), x.data.numel() + n * 4))
    return label, logits


def gather_labels(voxel_label, v2p_map):
    lib = _lib.load()
    _need_cuda(voxel_label, v2p_map)
    assert voxel_label.dtype == torch.int32 and v2p_map.dtype == torch.int64
    n = v2p_map.shape[0]
    out = torch.empty((n,), dtype=torch.int64, device=v2p_map.device)
    rc = lib.m3d_gather_labels(_ptr(voxel_label), _ptr(v2p_map.contiguous()), n, _ptr(out), _stream())
    _lib.check(rc, "m3d_gather_labels")
    return out


def avgpool(features, pair_fwd, num_out=None, calc_count=False):
    """spconv.pytorch.ops.indice_avgpool_implicit_gemm(features, pair_fwd, num_out, calc_count): mean of the input
    rows every output row pairs with (opensegment3d/model.py:272-276,299-307).  Returns (out [num_out, C], count or None)."""
    lib = _lib.load()
    _need_cuda(features, pair_fwd)
    if num_out is None:
        num_out = pair_fwd.shape[1]
    assert pair_fwd.dtype == torch.int32 and pair_fwd.dim() == 2 and pair_fwd.stride(1) == 1 and pair_fwd.shape[1] >= num_out
    in_dtype = features.dtype
    f = features.float().contiguous()
    squeeze = f.dim() == 1
    if squeeze:
        f = f.unsqueeze(1)
    c = f.shape[1]
    out = torch.empty((num_out, c), dtype=torch.float32, device=f.device)
    count = torch.empty((num_out,), dtype=torch.int32, device=f.device) if calc_count else None
    rc = lib.m3d_avgpool_fwd(_ptr(f), c, _ptr(pair_fwd), pair_fwd.shape[0], pair_fwd.stride(0), num_out, _ptr(out),
                             _ptr(count), _stream())
    _lib.check(rc, "m3d_avgpool_fwd")
    if squeeze:
        out = out.squeeze(1)
    return out.to(in_dtype), count
