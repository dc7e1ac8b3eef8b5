"""The benchmark configurations that no other GPU test runs end to end.

configs[0]: a Stray-Scanner-like capture (depth frames + poses) -> back-projection -> val transforms -> collate ->
open-vocabulary labels, as ONE chain, against the same chain on the CPU oracle (the reference has no
back-projection: its converter starts from a fused cloud, convert_stray.py:18-40; from the converted cloud on, the
oracle restates the reference).
configs[4]: one ~2 M-voxel scene at 1 cm through size-independent properties (the oracle would take minutes) and
label agreement between precisions, with 768- and 512-channel text embeddings."""
import numpy as np
import pytest
import torch

import bench
import oracle
from oracle import preprocess as P
from oracle import spunet as ospunet
from segmentation_amd import ops, preprocess as pp, scenes, spunet
from segmentation_amd.engine import SpUNetEngine

pytestmark = pytest.mark.gpu


def _capture(n_frames=3, h=384, w=512):
    """Synthetic depth frames of an undulating surface about 2 m away, seen from nearby poses."""
    rng = np.random.default_rng(11)
    v, u = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    frames = []
    for f in range(n_frames):
        d = 2000 + 350 * np.sin(u / 37.0 + f) * np.cos(v / 29.0) + 120 * np.sin((u + v) / 11.0) + rng.normal(0, 2, (h, w))
        depth = np.clip(d, 300, 6000).astype(np.uint16)
        depth[rng.uniform(size=(h, w)) < 0.03] = 0
        conf = np.full((h, w), 2, dtype=np.uint8)
        conf[rng.uniform(size=(h, w)) < 0.05] = 1
        rgb = rng.integers(0, 256, size=(h // 2, w // 2, 3)).astype(np.uint8)
        ang = 0.25 * (f - 1)
        rot = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        pose = np.eye(4)
        pose[:3, :3] = rot
        pose[:3, 3] = [0.6 * (f - 1), 0.05 * f, 0.0]
        frames.append((depth, conf, rgb, (420.0, 419.0, w / 2 - 0.5, h / 2 + 0.3), pose))
    return frames


def test_config0_capture_to_labels_chain(device):
    frames = _capture()
    bp = pp.Backprojector(capacity=3 * 384 * 512, device=device)
    ref_c, ref_k = [], []
    for depth, conf, rgb, intr, pose in frames:
        bp.add_frame(torch.from_numpy(depth.view(np.int16)).to(device), intr, pose, conf=torch.from_numpy(conf).to(device),
                     rgb=torch.from_numpy(rgb).to(device))
        c, k = P.backproject(depth, intr, pose, conf=conf, rgb=rgb)
        ref_c.append(c)
        ref_k.append(k)
    coord, color = bp.result()
    ref_c, ref_k = np.concatenate(ref_c), np.concatenate(ref_k)
    assert coord.shape[0] == ref_c.shape[0] > 400_000 and np.array_equal(color.cpu().numpy(), ref_k)
    assert np.abs(coord.cpu().numpy() - ref_c).max() <= 1e-6 * np.abs(ref_c).max()
    # val pipeline + collate (stray_refined_only.yaml:12-24, collate.py:73-104) on the GPU ...
    sample = pp.val_sample(coord, color, condition="ARKitScenes")
    batch = pp.collate([sample], grid_size=0.02)
    assert batch["offset"].tolist() == [0, coord.shape[0]] and batch["condition"] == ["ARKitScenes"]
    # ... and the same steps restated on the CPU, from the same converted cloud
    ob = P.collate([P.val_pipeline(coord.cpu().numpy(), color.cpu().numpy(), apply_z=True)], 0.02)
    oc, ofeat = ob["coord"], ob["feat"]
    assert list(ob["offset"]) == [0, oc.shape[0]]
    assert np.array_equal(batch["coord"].cpu().numpy(), oc) and np.array_equal(batch["feat"].cpu().numpy(), ofeat)
    cfg = ospunet.SPUNET34C   # the network of configs/model/spunet34c+ppt.yaml: layers (2, 3, 4, 6, 2, 2, 2, 2)
    params = ospunet.make_spunet_params(cfg, seed=2, trained_like=True)
    emb = torch.from_numpy(scenes.text_embeddings(20, cfg["out_channels"], seed=3))
    off = np.array([0, oc.shape[0]], dtype=np.int64)
    sp, ref = ospunet.spunet_from_points(params, cfg, oc, ofeat, off, 0.02, condition="ARKitScenes")
    ref_pred = oracle.head_predict(ref, sp["v2p_map"], emb)
    assert 60_000 < sp["voxel_coords"].shape[0] < 200_000
    sd = {("backbone." + k if not k.startswith("embedding_table") else k): v for k, v in params.items()}
    for prec in ("bf16x6", "bf16x3"):
        eng = SpUNetEngine(sd, cfg, batch["condition"][0], device, precision=prec)
        eng.set_text_embeddings(emb)
        out = eng.predict(batch["coord"], batch["feat"], batch["offset"], batch["grid_size"])
        assert out["n_voxels"] == sp["voxel_coords"].shape[0]
        assert np.array_equal(out["v2p_map"].cpu().numpy(), sp["v2p_map"])
        agree = (out["pred"].cpu() == ref_pred).float().mean().item()
        assert agree >= 0.999, (prec, agree)


@pytest.fixture(scope="module")
def huge(device):
    sc = scenes.make_scene(seed=0, target_voxels=2_000_000, grid_size=0.01, room=(15.0, 15.0, 3.0))
    n = sc["coord"].shape[0]
    return dict(coord=torch.from_numpy(sc["coord"]).to(device), feat=torch.from_numpy(sc["feat"]).to(device),
                offset=np.array([0, n], dtype=np.int64), n=n)


def test_config4_two_million_voxel_maps(device, huge):
    vox = ops.voxelize(huge["coord"], huge["feat"], huge["offset"], 0.01, return_keys=True, return_grid_coord=True)
    n0 = vox["n_voxels"]
    assert 1_800_000 < n0 < 3_000_000
    assert int(np.prod([float(s) for s in vox["spatial_shape"]])) > 2 ** 27   # beyond the occupancy-bitmap sizes of 2 cm scenes
    keys = vox["keys"].cpu().numpy().view(np.uint64)
    assert (keys[1:] > keys[:-1]).all()
    assert torch.equal(vox["voxel_coords"][:, 1:][vox["v2p_map"]], vox["grid_coord"])
    assert np.array_equal(oracle.fnv_hash_vec(vox["voxel_coords"][:200_000].cpu().numpy()), keys[:200_000])
    idx, shape = vox["voxel_coords"], vox["spatial_shape"]
    pair = ops.subm_rulebook(idx, shape)
    ar = torch.arange(n0, device=device, dtype=torch.int32)
    assert torch.equal(pair[13], ar)
    for k in (1, 9, 22):
        p = pair[k].long()
        ok = p >= 0
        assert torch.equal(pair[26 - k][p[ok]], ar[ok])
        off = torch.tensor([k // 9 - 1, (k // 3) % 3 - 1, k % 3 - 1], device=device, dtype=torch.int32)
        assert (idx[p[ok]][:, 1:] - idx[ok][:, 1:] == off).all() and (idx[p[ok]][:, 0] == idx[ok][:, 0]).all()
    # every voxel finds exactly the neighbours that exist: pair counts equal a brute-force count through sorted keys
    lin = ((idx[:, 1].long() * shape[1] + idx[:, 2].long()) * shape[2] + idx[:, 3].long())
    srt = torch.sort(lin).values
    for k in (0, 14, 25):
        d = ((k // 9 - 1) * shape[1] + ((k // 3) % 3 - 1)) * shape[2] + (k % 3 - 1)
        q = lin + d
        pos = torch.searchsorted(srt, q).clamp_max(n0 - 1)
        # (a wrapped coordinate cannot alias: every axis keeps 128 empty cells of padding)
        assert int((srt[pos] == q).sum()) == int((pair[k] >= 0).sum())
    chain = ops.strided_chain(idx, shape, 1, 4)
    cur, n_in = idx, n0
    for oi, oshape, pf, pb in chain:
        m = oi.shape[0]
        assert pb.shape == (8, n_in) and pf.shape == (8, m)
        hit = pb >= 0
        assert bool((hit.sum(0) == 1).all())                       # every input voxel has exactly one parent
        parent = pb.max(0).values.long()
        assert torch.equal(oi[parent][:, 1:], cur[:, 1:] >> 1)     # which is the voxel at coord >> 1
        lin_o = ((oi[:, 1].long() * oshape[1] + oi[:, 2].long()) * oshape[2] + oi[:, 3].long())
        assert bool((lin_o[1:] > lin_o[:-1]).all())                # outputs in ascending linear order
        assert int((pf >= 0).sum()) == n_in
        cur, n_in = oi, m


@pytest.mark.parametrize("dim", [768, 512])
def test_config4_label_agreement_between_precisions(device, huge, dim):
    cfg = dict(spunet.SPUNET34C, out_channels=dim)
    torch.manual_seed(0)
    model = spunet.PPT(spunet.SpUNet(**{k: cfg[k] for k in ("in_channels", "out_channels", "base_channels",
                                                             "context_channels", "channels", "layers", "conditions")}),
                       cfg["conditions"], 256).eval()
    emb = torch.from_numpy(scenes.text_embeddings(200, dim, seed=1))
    preds = {}
    for prec in ("bf16x6", "bf16x3", "bf16"):
        eng = SpUNetEngine(model, cfg, "ScanNetPP", device, precision=prec)
        eng.set_text_embeddings(emb)
        with torch.no_grad():
            preds[prec] = eng.predict(huge["coord"], huge["feat"], huge["offset"], 0.01)["pred"]
        del eng
        torch.cuda.empty_cache()
    assert preds["bf16x6"].shape[0] == huge["n"]
    assert (preds["bf16x3"] == preds["bf16x6"]).float().mean().item() >= 0.999
    assert (preds["bf16"] == preds["bf16x6"]).float().mean().item() >= 0.99   # stated tolerance of plain bf16
