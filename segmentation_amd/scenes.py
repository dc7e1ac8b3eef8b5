"""Seeded synthetic ScanNet-shaped scenes (there is no dataset access): an axis-aligned room with
floor, ceiling, four walls and 10-20 furniture boxes sampled on their surfaces, N(0, 3 mm) jitter,
u8 colours.  Output follows the reference's validation input contract
(mosaic3d/configs/data/local/stray_refined_only.yaml:12-24): CenterShift(apply_z=True)
(mosaic3d/src/data/utils/transform.py:149-163), NormalizeColor (:108-113), feat = color."""
import numpy as np


def _sample_box_surface(rng, lo, hi, n):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    d = hi - lo
    areas = np.array([d[1] * d[2], d[1] * d[2], d[0] * d[2], d[0] * d[2], d[0] * d[1], d[0] * d[1]])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    p = lo + rng.random((n, 3)) * d
    ax = face // 2
    side = face % 2
    p[np.arange(n), ax] = np.where(side == 0, lo[ax], hi[ax])
    return p, face


def make_scene(seed=0, target_voxels=250_000, grid_size=0.02, points_per_voxel=2.5, room=None,
               stray_axes=False):
    """Returns dict(coord [N,3] f32, color [N,3] u8, feat [N,3] f32).  The surface area is scaled so that
    the number of occupied `grid_size` voxels lands near `target_voxels`."""
    rng = np.random.default_rng(seed)
    if room is None:
        L, W = rng.uniform(4, 8, size=2)
        H = 2.6
    else:
        L, W, H = room
    boxes = [((0, 0, 0), (L, W, H))]
    for _ in range(int(rng.integers(10, 21))):
        sz = rng.uniform([0.4, 0.4, 0.3], [2.0, 1.5, 1.6])
        org = rng.uniform([0, 0, 0], [max(L - sz[0], 0.1), max(W - sz[1], 0.1), 0.4])
        boxes.append((org, org + sz))
    area = sum(2 * ((h[0] - l[0]) * (h[1] - l[1]) + (h[0] - l[0]) * (h[2] - l[2]) + (h[1] - l[1]) * (h[2] - l[2]))
               for l, h in (map(np.asarray, b) for b in boxes))
    # a surface of area A occupies ~1.15 * A / g^2 voxels once jittered; scale the scene to hit the target
    s = np.sqrt(target_voxels * grid_size ** 2 / (1.15 * area))
    n_pts = int(target_voxels * points_per_voxel)
    pts, cols = [], []
    areas = np.array([2 * ((h[0] - l[0]) * (h[1] - l[1]) + (h[0] - l[0]) * (h[2] - l[2]) + (h[1] - l[1]) * (h[2] - l[2]))
                      for l, h in (map(np.asarray, b) for b in boxes)])
    counts = rng.multinomial(n_pts, areas / areas.sum())
    for (lo, hi), cnt in zip(boxes, counts):
        p, face = _sample_box_surface(rng, np.asarray(lo) * s, np.asarray(hi) * s, cnt)
        base = rng.integers(30, 226, size=(6, 3))
        c = base[face] + rng.integers(-20, 21, size=(cnt, 3))
        pts.append(p)
        cols.append(np.clip(c, 0, 255))
    coord = np.concatenate(pts) + rng.normal(0, 0.003, size=(n_pts, 3))
    color = np.concatenate(cols).astype(np.uint8)
    perm = rng.permutation(n_pts)
    coord, color = coord[perm], color[perm]
    if stray_axes:
        # a Stray capture is Y-up; convert_stray.py:24-27 maps (x, y, z) -> (x, -z, y)
        R = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64)
        y_up = coord @ np.linalg.inv(R).T
        coord = y_up @ R.T
    coord = coord.astype(np.float32)
    # CenterShift(apply_z=True)
    mn, mx = coord.min(0), coord.max(0)
    shift = np.array([(mn[0] + mx[0]) / 2, (mn[1] + mx[1]) / 2, mn[2]], dtype=np.float32)
    coord = coord - shift
    feat = (color.astype(np.float32) / 127.5 - 1).astype(np.float32)
    return dict(coord=coord.astype(np.float32), color=color, feat=feat)


def make_batch(seeds, **kw):
    scenes = [make_scene(seed=s, **kw) for s in seeds]
    coord = np.concatenate([s["coord"] for s in scenes])
    feat = np.concatenate([s["feat"] for s in scenes])
    offset = np.concatenate([[0], np.cumsum([s["coord"].shape[0] for s in scenes])]).astype(np.int64)
    return dict(coord=coord, feat=feat, offset=offset)


def text_embeddings(n_classes, dim=768, seed=1):
    """Stand-in for the CLIP text tower's output (external frozen model, not fetchable offline):
    unit-norm random rows, like caption_utils.forward_text_encoder(normalize=True)."""
    rng = np.random.default_rng(seed)
    e = rng.standard_normal((n_classes, dim)).astype(np.float32)
    return e / np.linalg.norm(e, axis=1, keepdims=True)


def make_caption_batch(seeds, target_voxels, device, captions_per_scene=300, text_dim=768, unique_texts=150,
                       condition="ScanNet", grid_size=0.02, rng_seed=100):
    """One training batch of the caption-supervised step (train_spunet_multidata_ppt): len(seeds) synthetic scenes of
    one dataset condition, `captions_per_scene` captions per scene, each a random subset of 100-5000 of the scene's
    points and a unit text row drawn from `unique_texts` rows (so that rows repeat, as captions do), passed as
    precomputed `embedding` rows -- the dict mosaic3d/src/data/utils/collate.py:107-179 hands to
    DenseLanguageLitModule.training_step (language_module.py:209-277)."""
    import torch
    from . import train
    rng = np.random.default_rng(rng_seed)
    coords, feats, offs, embs, idxs = [], [], [0], [], []
    for seed in seeds:
        sc = make_scene(seed=seed, target_voxels=target_voxels, grid_size=grid_size)
        n = sc["coord"].shape[0]
        coords.append(sc["coord"])
        feats.append(sc["feat"])
        offs.append(offs[-1] + n)
        rows = rng.normal(size=(unique_texts, text_dim)).astype(np.float32)
        rows /= np.linalg.norm(rows, axis=1, keepdims=True)
        uniq = torch.from_numpy(rows).to(device)
        embs.append([uniq[int(rng.integers(0, unique_texts))] for _ in range(captions_per_scene)])
        idxs.append([torch.from_numpy(rng.choice(n, size=int(rng.integers(100, min(5000, n) + 1)), replace=False))
                     for _ in range(captions_per_scene)])
    pi, co, npc = train.flatten_caption_indices(idxs, torch.tensor(offs))
    return dict(coord=torch.from_numpy(np.concatenate(coords)).to(device),
                feat=torch.from_numpy(np.concatenate(feats)).to(device), offset=torch.tensor(offs), grid_size=grid_size,
                condition=[condition],
                caption_data=dict(embedding=embs, point_indices=pi.to(device), caption_offsets=co,
                                  num_points_per_caption=npc))
