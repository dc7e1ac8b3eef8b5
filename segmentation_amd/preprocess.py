"""Host-side mirror of the steps in front of the voxelizer, on GPU tensors and through the C ABI.

Reference: convert_stray.py:18-40 (Stray Scanner cloud -> coord / color arrays),
mosaic3d/src/data/utils/transform.py:49-64,108-113,149-163,23-46 (Copy, NormalizeColor, CenterShift, Collect as
wired in configs/data/local/stray_refined_only.yaml:12-24), mosaic3d/src/data/utils/collate.py:73-104 (offset
keys become cumulative sums with a leading 0) and transform.py:726-887 (GridSample).
"""
from typing import Dict, List, Sequence

import torch

from . import _lib
from .ops import HASH, _need_cuda, _ptr, _stream, _workspace


def stray_convert(points: torch.Tensor, colors01: torch.Tensor = None):
    """points [N,3] float64 (Y-up), colours [N,3] float64 in [0,1] -> (coord [N,3] f32 Z-up, color [N,3] u8)."""
    lib = _lib.load()
    _need_cuda(points)
    points = points.contiguous().double()
    n = points.shape[0]
    coord = torch.empty((n, 3), dtype=torch.float32, device=points.device)
    color = None
    if colors01 is not None:
        colors01 = colors01.contiguous().double()
        assert colors01.shape == (n, 3)
        color = torch.empty((n, 3), dtype=torch.uint8, device=points.device)
    rc = lib.m3d_stray_convert(_ptr(points), _ptr(colors01), n, _ptr(coord), _ptr(color), _stream())
    _lib.check(rc, "m3d_stray_convert")
    return coord, color


class Backprojector:
    """Accumulates the world points of Stray Scanner frames on the GPU (m3d_backproject): `add_frame` appends the
    valid pixels of one depth frame (already rotated to Mosaic3D's Z-up axes), `result()` returns coord [M,3] f32
    and color [M,3] u8, ready for `val_sample`.  The running count stays on the device; frames are enqueued
    without host read-backs."""

    def __init__(self, capacity: int, device="cuda", with_color: bool = True):
        self.capacity = int(capacity)
        self.coord = torch.empty((self.capacity, 3), dtype=torch.float32, device=device)
        self.color = torch.empty((self.capacity, 3), dtype=torch.uint8, device=device) if with_color else None
        self.count = torch.zeros((1,), dtype=torch.int64, device=device)

    def add_frame(self, depth_u16: torch.Tensor, intrinsics, pose_c2w, conf: torch.Tensor = None, rgb: torch.Tensor = None,
                  depth_scale: float = 1e-3, depth_max: float = 5.0, conf_min: int = 2, stride: int = 1):
        import ctypes
        lib = _lib.load()
        _need_cuda(depth_u16)
        h, w = depth_u16.shape
        depth_u16 = depth_u16.contiguous()
        assert depth_u16.dtype in (torch.uint16, torch.int16)
        if conf is not None:
            conf = conf.contiguous()
            assert conf.dtype == torch.uint8 and conf.shape == (h, w)
        hc = wc = 0
        if self.color is not None:
            assert rgb is not None and rgb.dtype == torch.uint8 and rgb.dim() == 3 and rgb.shape[2] == 3
            rgb = rgb.contiguous()
            hc, wc = rgb.shape[:2]
        else:
            rgb = None
        k = (ctypes.c_double * 4)(*[float(x) for x in intrinsics])
        p = (ctypes.c_double * 16)(*[float(x) for x in torch.as_tensor(pose_c2w, dtype=torch.float64).flatten().tolist()])
        ws = _workspace(lib.m3d_backproject_workspace_bytes(h, w, int(stride)), depth_u16.device)
        rc = lib.m3d_backproject(_ptr(depth_u16), _ptr(conf), _ptr(rgb), h, w, hc, wc, k, p, float(depth_scale),
                                 float(depth_max), int(conf_min), int(stride), _ptr(self.coord), _ptr(self.color),
                                 self.capacity, _ptr(self.count), _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "m3d_backproject")

    def result(self):
        m = int(self.count.item())
        return self.coord[:m], (None if self.color is None else self.color[:m])


def center_shift_normalize(coord: torch.Tensor, color_u8: torch.Tensor = None, apply_z: bool = True):
    """CenterShift(apply_z) + NormalizeColor + `.float()`: returns (coord - shift, feat or None, shift [3])."""
    lib = _lib.load()
    _need_cuda(coord)
    coord = coord.contiguous().float()
    n = coord.shape[0]
    if n == 0:
        raise ValueError("center_shift_normalize needs at least one point")
    dev = coord.device
    out = torch.empty_like(coord)
    shift = torch.empty((3,), dtype=torch.float32, device=dev)
    feat = None
    if color_u8 is not None:
        assert color_u8.dtype == torch.uint8 and color_u8.shape == (n, 3)
        color_u8 = color_u8.contiguous()
        feat = torch.empty((n, 3), dtype=torch.float32, device=dev)
    ws = _workspace(lib.m3d_center_shift_workspace_bytes(n), dev)
    rc = lib.m3d_center_shift_normalize(_ptr(coord), _ptr(color_u8), n, int(bool(apply_z)), _ptr(out), _ptr(feat),
                                        _ptr(shift), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "m3d_center_shift_normalize")
    return out, feat, shift


def val_sample(coord: torch.Tensor, color_u8: torch.Tensor, condition: str = "ARKitScenes", apply_z: bool = True) -> Dict:
    """One sample as the val pipeline of stray_refined_only.yaml:12-24 leaves it (Copy, CenterShift,
    NormalizeColor, Add, ToTensor, Collect)."""
    shifted, feat, _ = center_shift_normalize(coord, color_u8, apply_z)
    n = coord.shape[0]
    return dict(origin_coord=coord.float(), coord=shifted, color=feat, feat=feat, condition=condition,
                offset=torch.tensor([n]), pc_count=torch.tensor([n]))


def collate(samples: Sequence[Dict], grid_size: float) -> Dict:
    """point_collate_fn_with_masks for dict samples (collate.py:73-104,107-179): tensors are concatenated, strings
    listed, every key containing "offset" becomes the cumulative sum with a leading 0; `grid_size` is attached."""
    samples = [s for s in samples if s is not None]
    out = {}
    for key in samples[0]:
        vals = [s[key] for s in samples]
        if isinstance(vals[0], str):
            out[key] = list(vals)
        elif isinstance(vals[0], torch.Tensor):
            out[key] = torch.cat(vals)
            if "offset" in key:
                cs = torch.cumsum(out[key], dim=0)
                out[key] = torch.cat((torch.zeros(1, dtype=torch.int32, device=cs.device), cs))
        else:
            out[key] = vals
    out["grid_size"] = grid_size
    return out


def grid_sample(coord: torch.Tensor, grid_size: float, hash_type: str = "fnv", mode: str = "train",
                return_displacement: bool = False, generator: torch.Generator = None) -> Dict:
    """GridSample on the GPU (transform.py:726-887).  Deterministic outputs (bit-exact with the reference):
    `grid_coord` [N,3] per point, `inverse` [N], `count` [M], `voxel_grid_coord` [M,3], `min_coord` [1,3] float64,
    optional `displacement` [N,3] float64.  Selection: mode "train" -> `index` [M], one random point per voxel
    (torch RNG, not numpy's); mode "test" -> `parts`, a list of count.max() index tensors, part i holding point
    i % count of every voxel (points ordered by index inside a voxel)."""
    lib = _lib.load()
    _need_cuda(coord)
    assert mode in ("train", "test")
    coord = coord.contiguous().float()
    n = coord.shape[0]
    if n == 0:
        raise ValueError("grid_sample needs at least one point")
    dev = coord.device
    grid = torch.empty((n, 3), dtype=torch.int32, device=dev)
    inverse = torch.empty((n,), dtype=torch.int64, device=dev)
    vcoords = torch.empty((n, 4), dtype=torch.int32, device=dev)
    count = torch.empty((n,), dtype=torch.int32, device=dev)
    keys = torch.empty((n,), dtype=torch.int64, device=dev)
    meta = torch.empty((16,), dtype=torch.int64, device=dev)
    ws = _workspace(lib.m3d_grid_sample_workspace_bytes(n), dev)
    rc = lib.m3d_grid_sample(_ptr(coord), n, float(grid_size), HASH[hash_type], _ptr(grid), _ptr(inverse), _ptr(vcoords),
                             _ptr(count), _ptr(keys), _ptr(meta), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "m3d_grid_sample")
    m = meta.cpu().tolist()
    nv = m[0]
    count = count[:nv]
    gmin = torch.tensor(m[4:7], dtype=torch.float64)
    out = dict(grid_coord=grid, inverse=inverse, count=count, voxel_grid_coord=vcoords[:nv, 1:], keys=keys[:nv],
               min_coord=(gmin * float(grid_size)).reshape(1, 3), n_voxels=nv)
    if return_displacement:
        scaled = coord.double() / float(grid_size) - gmin.to(dev)
        out["displacement"] = scaled - grid.double() - 0.5
    # points ordered by (voxel, index); voxel v owns positions [start[v], start[v] + count[v])
    order = torch.sort(inverse, stable=True).indices
    start = torch.cumsum(count.long(), 0) - count.long()
    if mode == "train":
        r = torch.randint(0, int(count.max()), (nv,), device=dev, generator=generator) % count.long()
        out["index"] = order[start + r]
    else:
        out["parts"] = [order[start + (i % count.long())] for i in range(int(count.max()))]
    return out
