"""oracle/preprocess.py against the outputs of the reference's own transform / collate classes
(tests/golden/reference_preprocess.npz, made by tests/golden/make_golden_preprocess.py)."""
import os

import numpy as np
import pytest

from oracle import preprocess as P

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_preprocess.npz"))
CLOUDS = ("a", "b", "c")


@pytest.mark.parametrize("name", CLOUDS)
def test_val_pipeline_bit_exact(name):
    r = P.val_pipeline(G[f"{name}_coord"], G[f"{name}_color"], apply_z=True)
    assert np.array_equal(r["coord"], G[f"{name}_val_coord"]) and r["coord"].dtype == np.float32
    assert np.array_equal(r["feat"], G[f"{name}_val_feat"]) and r["feat"].dtype == np.float32
    assert np.array_equal(r["offset"], G[f"{name}_val_offset"])
    assert np.array_equal(P.center_shift(G[f"{name}_coord"], apply_z=False)[0], G[f"{name}_shift_noz"])


def test_collate_offsets():
    samples = [P.val_pipeline(G[f"{n}_coord"], G[f"{n}_color"]) for n in CLOUDS]
    b = P.collate(samples, 0.02)
    assert np.array_equal(b["offset"], G["batch_offset"]) and np.array_equal(b["pc_count"], G["batch_pc_count"])
    assert np.array_equal(b["coord"], G["batch_coord"]) and np.array_equal(b["feat"], G["batch_feat"])
    assert b["grid_size"] == G["batch_grid_size"][0]


@pytest.mark.parametrize("name", CLOUDS)
def test_hashes_bit_exact(name):
    assert np.array_equal(P.fnv_hash_vec3(G[f"{name}_hash_in"]), G[f"{name}_fnv3"])
    assert np.array_equal(P.ravel_hash_vec(G[f"{name}_hash_in"]), G[f"{name}_ravel3"])


@pytest.mark.parametrize("name", CLOUDS)
@pytest.mark.parametrize("hash_type", ["fnv", "ravel"])
@pytest.mark.parametrize("g", [0.05, 0.02])
def test_grid_sample_bit_exact(name, hash_type, g):
    pre = f"{name}_gs_{hash_type}_{int(g * 100)}_"
    r = P.grid_sample(G[f"{name}_coord"], g, hash_type)
    assert np.array_equal(r["inverse"], G[pre + "inverse"])
    assert np.array_equal(r["voxel_grid_coord"], G[pre + "voxel_grid_coord"])
    assert np.array_equal(r["min_coord"], G[pre + "min_coord"])
    parts = P.grid_sample_test_parts(r["inverse"], r["count"])
    assert len(parts) == G[pre + "n_parts"][0]
    seen = np.concatenate(parts)
    assert set(seen.tolist()) == set(range(len(G[f"{name}_coord"])))  # every point appears in some part
    for part in parts:  # every part holds exactly one point of every voxel
        assert np.array_equal(np.sort(r["inverse"][part]), np.arange(len(r["count"])))


def test_stray_convert_known_answers():
    pts = np.array([[1.0, 2.0, 3.0], [-0.5, 0.25, 4.0]])
    col = np.array([[0.0, 1.0, 0.5], [0.999, 0.2, 254.9 / 255]])
    coord, color = P.stray_convert(pts, col)
    assert np.array_equal(coord, np.array([[1, -3, 2], [-0.5, -4, 0.25]], dtype=np.float32))
    assert np.array_equal(color, np.array([[0, 255, 127], [254, 51, 254]], dtype=np.uint8))
