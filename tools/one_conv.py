"""One level-0 submanifold 96->96 convolution, for counter collection: python3 tools/one_conv.py [reps]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from segmentation_amd import ops, scenes
kind = "dma"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
sc = scenes.make_scene(seed=0, target_voxels=290000)
n = sc["coord"].shape[0]
vox = ops.voxelize(torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n]), 0.02)
idx = vox["voxel_coords"]; N = idx.shape[0]
pair = ops.subm_rulebook(idx, vox["spatial_shape"])
mode = int(os.environ.get("SMODE", "1539"))
sched = ops.build_schedule(pair, N, idx, mode, vox["spatial_shape"], 1)
x = ops.split_f32(torch.randn(N, 96, device=dev), 2)
w = ops.pack_weight(torch.randn(96, 27, 96) * 0.03, "bf16x3", dev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops.conv_fwd(x, w, sched, N)
e0.record()
for _ in range(reps): ops.conv_fwd(x, w, sched, N)
e1.record(); torch.cuda.synchronize()
print(kind, "mode", mode, f"{e0.elapsed_time(e1) / reps * 1e3:.1f} us")
