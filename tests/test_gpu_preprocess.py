"""GPU front end (m3d_stray_convert, m3d_center_shift_normalize, m3d_grid_sample, preprocess.collate) against the
reference's own outputs (tests/golden/reference_preprocess.npz) and the CPU oracle: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess as P
from segmentation_amd import preprocess as pp

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_preprocess.npz"))
CLOUDS = ("a", "b", "c")


@pytest.mark.parametrize("name", CLOUDS)
def test_val_pipeline_matches_reference(device, name):
    coord = torch.from_numpy(G[f"{name}_coord"]).to(device)
    color = torch.from_numpy(G[f"{name}_color"]).to(device)
    s = pp.val_sample(coord, color)
    assert np.array_equal(s["coord"].cpu().numpy(), G[f"{name}_val_coord"])
    assert np.array_equal(s["feat"].cpu().numpy(), G[f"{name}_val_feat"])
    assert np.array_equal(s["offset"].numpy(), G[f"{name}_val_offset"])
    noz, _, shift = pp.center_shift_normalize(coord, None, apply_z=False)
    assert np.array_equal(noz.cpu().numpy(), G[f"{name}_shift_noz"]) and float(shift[2]) == 0.0


def test_collate_matches_reference(device):
    samples = [pp.val_sample(torch.from_numpy(G[f"{n}_coord"]).to(device), torch.from_numpy(G[f"{n}_color"]).to(device))
               for n in CLOUDS]
    b = pp.collate(samples, 0.02)
    assert np.array_equal(b["offset"].numpy(), G["batch_offset"]) and b["offset"].dtype in (torch.int64, torch.int32)
    assert np.array_equal(b["pc_count"].numpy(), G["batch_pc_count"])
    assert np.array_equal(b["coord"].cpu().numpy(), G["batch_coord"])
    assert np.array_equal(b["feat"].cpu().numpy(), G["batch_feat"])
    assert b["condition"] == ["ARKitScenes"] * 3 and b["grid_size"] == 0.02


@pytest.mark.parametrize("name", CLOUDS)
@pytest.mark.parametrize("hash_type", ["fnv", "ravel"])
@pytest.mark.parametrize("g", [0.05, 0.02])
def test_grid_sample_matches_reference(device, name, hash_type, g):
    pre = f"{name}_gs_{hash_type}_{int(g * 100)}_"
    coord = torch.from_numpy(G[f"{name}_coord"]).to(device)
    r = pp.grid_sample(coord, g, hash_type, mode="test", return_displacement=True)
    o = P.grid_sample(G[f"{name}_coord"], g, hash_type)
    assert np.array_equal(r["inverse"].cpu().numpy(), G[pre + "inverse"])
    assert np.array_equal(r["voxel_grid_coord"].cpu().numpy(), G[pre + "voxel_grid_coord"])
    assert np.array_equal(r["min_coord"].numpy(), G[pre + "min_coord"])
    assert np.array_equal(r["grid_coord"].cpu().numpy(), o["grid_coord"])
    assert np.array_equal(r["count"].cpu().numpy(), o["count"])
    assert np.array_equal(r["keys"].cpu().numpy().view(np.uint64), np.unique(o["key"]))
    assert np.array_equal(r["displacement"].cpu().numpy(), o["displacement"])
    assert len(r["parts"]) == G[pre + "n_parts"][0]
    ref_parts = P.grid_sample_test_parts(o["inverse"], o["count"])
    for a, b in zip(r["parts"], ref_parts):
        assert np.array_equal(a.cpu().numpy(), b)
    t = pp.grid_sample(coord, g, hash_type, mode="train", generator=torch.Generator(device=device).manual_seed(3))
    idx = t["index"].cpu().numpy()
    assert np.array_equal(o["inverse"][idx], np.arange(len(o["count"])))  # one point of every voxel, in voxel order


def test_grid_sample_large_cloud_and_negative_coordinates(device):
    rng = np.random.default_rng(1)
    coord = (rng.uniform(-4, 4, size=(200000, 3))).astype(np.float32)
    r = pp.grid_sample(torch.from_numpy(coord).to(device), 0.02, "fnv", mode="test")
    o = P.grid_sample(coord, 0.02, "fnv")
    assert np.array_equal(r["inverse"].cpu().numpy(), o["inverse"])
    assert np.array_equal(r["voxel_grid_coord"].cpu().numpy(), o["voxel_grid_coord"])


def test_stray_convert(device):
    rng = np.random.default_rng(2)
    pts = rng.normal(size=(5000, 3))
    col = rng.uniform(0, 1, size=(5000, 3))
    col[:4] = [[0, 1, 0.5], [0.999, 0.2, 254.9 / 255], [1 / 255, 2 / 255, 3 / 255], [0.1, 0.7, 0.3]]
    coord, color = pp.stray_convert(torch.from_numpy(pts).to(device), torch.from_numpy(col).to(device))
    rc, rcol = P.stray_convert(pts, col)
    assert np.array_equal(coord.cpu().numpy(), rc) and np.array_equal(color.cpu().numpy(), rcol)
    only, none = pp.stray_convert(torch.from_numpy(pts).to(device))
    assert none is None and np.array_equal(only.cpu().numpy(), rc)


def _frame(rng, h=192, w=256, hc=1440, wc=1920):
    depth = rng.integers(300, 6000, size=(h, w)).astype(np.uint16)
    depth[rng.uniform(size=(h, w)) < 0.1] = 0
    conf = rng.integers(0, 3, size=(h, w)).astype(np.uint8)
    rgb = rng.integers(0, 256, size=(hc, wc, 3)).astype(np.uint8)
    ang = rng.uniform(-1, 1, size=3)
    cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    rot = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ \
        np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    pose = np.eye(4)
    pose[:3, :3] = rot
    pose[:3, 3] = rng.uniform(-2, 2, size=3)
    intr = (212.3, 211.9, w / 2 - 0.7, h / 2 + 0.4)
    return depth, conf, rgb, intr, pose


@pytest.mark.parametrize("stride", [1, 2])
def test_backproject_frames_accumulate_in_pixel_order(device, stride):
    # no back-projection exists in the reference: the oracle's pinhole definition is the bar (parity unpinned)
    rng = np.random.default_rng(5)
    frames = [_frame(rng) for _ in range(3)]
    bp = pp.Backprojector(capacity=3 * 192 * 256, device=device)
    ref_c, ref_k = [], []
    for depth, conf, rgb, intr, pose in frames:
        bp.add_frame(torch.from_numpy(depth.view(np.int16)).to(device), intr, pose, conf=torch.from_numpy(conf).to(device), rgb=torch.from_numpy(rgb).to(device), stride=stride)
        c, k = P.backproject(depth, intr, pose, conf=conf, rgb=rgb, stride=stride)
        ref_c.append(c)
        ref_k.append(k)
    coord, color = bp.result()
    ref_c, ref_k = np.concatenate(ref_c), np.concatenate(ref_k)
    assert coord.shape[0] == ref_c.shape[0] > 5000
    assert np.array_equal(color.cpu().numpy(), ref_k)
    # float64 maths on both sides; fma contraction on the GPU may differ in the last float32 bit
    assert np.abs(coord.cpu().numpy() - ref_c).max() <= 1e-6 * np.abs(ref_c).max()


def test_backproject_drops_a_frame_that_does_not_fit(device):
    rng = np.random.default_rng(6)
    depth, conf, rgb, intr, pose = _frame(rng)
    n_valid = P.backproject(depth, intr, pose, conf=conf)[0].shape[0]
    bp = pp.Backprojector(capacity=n_valid + 10, device=device, with_color=False)
    d = torch.from_numpy(depth.view(np.int16)).to(device)
    bp.add_frame(d, intr, pose, conf=torch.from_numpy(conf).to(device))
    bp.add_frame(d, intr, pose, conf=torch.from_numpy(conf).to(device))
    coord, color = bp.result()
    assert coord.shape[0] == n_valid and color is None
