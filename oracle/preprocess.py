"""CPU restatement of the steps in front of the voxelizer (test infrastructure only).

Follows, line by line:
  * convert_stray.py:18-40                      Stray Scanner cloud -> coord.npy / color.npy
  * mosaic3d/src/data/utils/transform.py:149-163 CenterShift, :108-113 NormalizeColor, :67-92 ToTensor,
    :23-46 Collect (wiring: configs/data/local/stray_refined_only.yaml:12-24)
  * mosaic3d/src/data/utils/collate.py:73-104    collate_fn (offset keys -> cumulative sums with a leading 0)
  * mosaic3d/src/data/utils/transform.py:751-887 GridSample (hash, unique, inverse, grid_coord, min_coord,
    displacement; the random / per-partition point selection is not reproducible bit for bit and is restated with
    a stable order instead)
Pinned by tests/golden/reference_preprocess.npz (outputs of the reference's own classes).
"""
import numpy as np

FNV_OFFSET = np.uint64(14695981039346656037)
FNV_PRIME = np.uint64(1099511628211)


def stray_convert(points, colors01):
    """convert_stray.py:24-39: Y-up -> Z-up (x, y, z) -> (x, -z, y); colours in [0, 1] -> u8 by truncation."""
    r = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]])
    coord = (np.asarray(points) @ r.T).astype(np.float32)
    color = (np.asarray(colors01) * 255).astype(np.uint8)
    return coord, color


def center_shift(coord, apply_z=True):
    """transform.py:154-163, on float32 coordinates (all arithmetic stays float32)."""
    coord = np.array(coord, dtype=np.float32, copy=True)
    x_min, y_min, z_min = coord.min(axis=0)
    x_max, y_max, _ = coord.max(axis=0)
    shift = [(x_min + x_max) / 2, (y_min + y_max) / 2, z_min if apply_z else 0]
    coord -= shift
    return coord, np.array(shift, dtype=np.float32)


def normalize_color(color_u8):
    """transform.py:110-112 (float64: integer array / Python float) then ToTensor/Collect `.float()`."""
    return (np.asarray(color_u8) / 127.5 - 1).astype(np.float32)


def val_pipeline(coord, color_u8, apply_z=True):
    """Copy + CenterShift + NormalizeColor + ToTensor + Collect(feat_keys=[color]) of the val config."""
    shifted, _ = center_shift(coord, apply_z)
    feat = normalize_color(color_u8)
    n = shifted.shape[0]
    return dict(origin_coord=np.array(coord, dtype=np.float32), coord=shifted, color=feat, feat=feat,
                offset=np.array([n], dtype=np.int64), pc_count=np.array([n], dtype=np.int64))


def collate(samples, grid_size):
    """collate.py:73-104,107-163 for dict samples of arrays: concatenate, offset keys -> cumsum with leading 0."""
    out = {}
    for key in samples[0]:
        vals = [s[key] for s in samples]
        if isinstance(vals[0], str):
            out[key] = list(vals)
            continue
        out[key] = np.concatenate([np.asarray(v) for v in vals], 0)
        if "offset" in key:
            out[key] = np.concatenate([np.zeros(1, dtype=np.int64), np.cumsum(out[key])]).astype(np.int64)
    out["grid_size"] = grid_size
    return out


def fnv_hash_vec3(arr):
    """transform.py:876-887."""
    arr = np.asarray(arr).copy().astype(np.uint64, copy=False)
    h = FNV_OFFSET * np.ones(arr.shape[0], dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(arr.shape[1]):
            h = h * FNV_PRIME
            h = np.bitwise_xor(h, arr[:, j])
    return h


def ravel_hash_vec(arr):
    """transform.py:858-873."""
    arr = np.asarray(arr).copy()
    arr -= arr.min(0)
    arr = arr.astype(np.uint64, copy=False)
    arr_max = arr.max(0).astype(np.uint64) + np.uint64(1)
    keys = np.zeros(arr.shape[0], dtype=np.uint64)
    for j in range(arr.shape[1] - 1):
        keys += arr[:, j]
        keys *= arr_max[j + 1]
    keys += arr[:, -1]
    return keys


def grid_sample(coord, grid_size, hash_type="fnv"):
    """The deterministic part of GridSample.__call__ (transform.py:751-762, 813-826): per-point grid coordinate,
    hash key, voxel id of every point (`inverse`, voxels in ascending key order), per-voxel count and grid
    coordinate, min_coord and per-point displacement to the voxel centre."""
    scaled = np.asarray(coord) / np.array(grid_size)
    grid = np.floor(scaled).astype(int)
    gmin = grid.min(0)
    grid -= gmin
    scaled = scaled - gmin
    key = (fnv_hash_vec3 if hash_type == "fnv" else ravel_hash_vec)(grid)
    idx_sort = np.argsort(key, kind="stable")
    ukey, first, inv_sorted, count = np.unique(key[idx_sort], return_index=True, return_inverse=True, return_counts=True)
    inverse = np.zeros_like(inv_sorted)
    inverse[idx_sort] = inv_sorted
    return dict(grid_coord=grid.astype(np.int32), key=key, inverse=inverse.astype(np.int64), count=count.astype(np.int32),
                voxel_grid_coord=grid[idx_sort[first]].astype(np.int32), min_coord=(gmin * np.array(grid_size)).reshape(1, 3),
                displacement=scaled - grid - 0.5, idx_sort=idx_sort)


def grid_sample_test_parts(inverse, count):
    """Test mode (transform.py:827-855) with points ordered by index inside a voxel: part i takes point i % count
    of every voxel."""
    idx_sort = np.argsort(inverse, kind="stable")
    start = np.cumsum(np.insert(count, 0, 0)[0:-1])
    return [idx_sort[start + i % count] for i in range(int(count.max()))]


def backproject(depth_u16, intrinsics, pose_c2w, conf=None, rgb=None, depth_scale=1e-3, depth_max=5.0, conf_min=2,
                stride=1):
    """Pinhole back-projection of one depth frame followed by the converter's rotation (convert_stray.py:24-29).
    The reference has no back-projection of its own (it starts from a cloud made elsewhere): parity unpinned,
    this is the definition the GPU kernel is held to.  Camera axes x right, y down, z forward; float64 maths;
    kept pixels in row-major order."""
    depth = np.asarray(depth_u16)[::stride, ::stride]
    h, w = np.asarray(depth_u16).shape
    v, u = np.meshgrid(np.arange(0, h, stride), np.arange(0, w, stride), indexing="ij")
    keep = depth != 0
    if conf is not None:
        keep &= np.asarray(conf)[::stride, ::stride] >= conf_min
    d = depth.astype(np.float64) * depth_scale
    keep &= d <= depth_max
    fx, fy, cx, cy = [float(x) for x in intrinsics]
    pose = np.asarray(pose_c2w, dtype=np.float64).reshape(4, 4)
    u, v, d = u[keep].astype(np.float64), v[keep].astype(np.float64), d[keep]
    xc, yc, zc = (u - cx) / fx * d, (v - cy) / fy * d, d
    world = [pose[r, 0] * xc + pose[r, 1] * yc + pose[r, 2] * zc + pose[r, 3] for r in range(3)]
    coord = np.stack([world[0], -world[2], world[1]], 1).astype(np.float32)
    color = None
    if rgb is not None:
        rgb = np.asarray(rgb)
        hc, wc = rgb.shape[:2]
        uc = np.minimum((u.astype(np.int64) * wc) // w, wc - 1)
        vc = np.minimum((v.astype(np.int64) * hc) // h, hc - 1)
        color = rgb[vc, uc]
    return coord, color
