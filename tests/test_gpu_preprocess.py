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
