"""Fill of the tile schedules of the benchmark scene: real pairs against the (row, offset) slots the convolution
kernel issues matrix work for, at workgroup (128-row) and wave (32-row) granularity, per level.
usage: python3 tools/fill_stats.py   (M3D_SCHED_MODES as the engine reads it)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import ops, scenes

dev = torch.device("cuda:0")
eng, model, cfg = bench.build_engine(dev, "bf16x6")
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
vox = ops.voxelize(torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n]), 0.02)
levels = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], 1)
for l, lv in enumerate(levels):
    for name, s in (("subm", lv.subm), ("down", lv.down), ("up", lv.up)):
        if s is None:
            continue
        valid = (s.nbr >= 0) & (s.out_row >= 0)[None, :]          # [k, npad]
        pairs = int(valid.sum())
        w = valid.view(s.k, -1, 32).any(2)                         # [k, npad/32] wave tiles
        b = valid.view(s.k, -1, 128).any(2)                        # [k, npad/128] workgroup tiles
        print(f"level {l} {name}: rows {s.n} pairs {pairs} ({pairs / s.n:.2f}/row)  active offsets per 128-tile "
              f"{b.sum().item() / b.shape[1]:.2f}  fill at 128 rows {pairs / (b.sum().item() * 128):.3f}  "
              f"at 32 rows {pairs / (w.sum().item() * 32):.3f}", flush=True)
