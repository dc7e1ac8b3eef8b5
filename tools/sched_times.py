"""Time of the pattern schedule (m3d_build_schedule, mode 3) and the submanifold map per resolution level."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import ops, scenes

dev = torch.device("cuda:0")
eng, model, cfg = bench.build_engine(dev, "bf16x3")
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
coord, feat = torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev)
vox = ops.voxelize(coord, feat, np.array([0, n]), 0.02, pad=128)
levels = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], 1)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for l, lv in enumerate(levels):
    t_map = timed(lambda: ops.subm_rulebook(lv.indices, lv.shape))
    t_s3 = timed(lambda: ops.build_schedule(lv.pair_subm, lv.n, lv.indices, 3, lv.shape, 1))
    line = f"level {l} n={lv.n:7d}  subm map {t_map:7.1f} us  pattern schedule {t_s3:7.1f} us"
    if lv.pair_fwd is not None:
        t_d = timed(lambda: ops.build_schedule(lv.pair_fwd, lv.n_next, None, 0))
        t_u = timed(lambda: ops.build_schedule(lv.pair_bwd, lv.n, None, 2, base_order=lv.subm.out_row))
        line += f"  down {t_d:6.1f} us  up {t_u:6.1f} us"
    print(line)
