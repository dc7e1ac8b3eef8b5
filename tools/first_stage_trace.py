"""Kernel sequence of the first stage (voxelizer + kernel maps + schedules) of one benchmark scene: run under
rocprofv3 --kernel-trace, then list with tools/first_stage_list.py."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from segmentation_amd import ops, scenes
dev = torch.device("cuda:0")
eng, model, cfg = bench.build_engine(dev, "bf16x6")
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
coord, feat, off = torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n])
for it in range(4):
    torch.cuda.synchronize()
    vox = ops.voxelize(coord, feat, off, 0.02, pad=128)
    levels = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], vox["batch_size"])
    torch.cuda.synchronize()
