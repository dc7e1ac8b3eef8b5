"""Host-side enqueue time of the map and backbone phases against their GPU time, and a cProfile of five forward passes."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cProfile, pstats
import bench
from segmentation_amd import ops, scenes
dev = torch.device("cuda:0")
eng, model, cfg = bench.build_engine(dev, "bf16x3")
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
coord, feat = torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev)
off = np.array([0, n])
with torch.no_grad():
    for _ in range(3): eng.predict(coord, feat, off, 0.02)
    vox = ops.voxelize(coord, feat, off, 0.02, pad=128)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); levels = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], 1); t1 = time.perf_counter()
    torch.cuda.synchronize(); t1b = time.perf_counter()
    x = eng.backbone(vox["voxel_feats"], levels); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"build_levels host {1e3*(t1-t0):.2f} ms (done {1e3*(t1b-t0):.2f}); backbone host {1e3*(t2-t1b):.2f} ms (done {1e3*(t3-t1b):.2f})")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        levels = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], 1)
        x = eng.backbone(vox["voxel_feats"], levels)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
