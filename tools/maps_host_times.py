"""Host time of each op while the kernel maps are built, against the GPU time of the whole phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import ops, scenes
dev = torch.device("cuda:0")
eng, model, cfg = bench.build_engine(dev, "bf16x3")
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
coord, feat = torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev)
off = np.array([0, n])
acc = {}
def wrap(name):
    f = getattr(ops, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(ops, name, g)
for nm in ("voxelize", "subm_rulebook", "strided_rulebook", "build_schedule"): wrap(nm)
with torch.no_grad():
    for _ in range(3): eng.predict(coord, feat, off, 0.02)
    acc.clear()
    R = 10
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tv = tm = 0.0
    for _ in range(R):
        torch.cuda.synchronize()
        ev[0].record(); vox = ops.voxelize(coord, feat, off, 0.02, pad=128); ev[1].record()
        levels = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], vox["batch_size"]); ev[2].record()
        torch.cuda.synchronize(); tv += ev[0].elapsed_time(ev[1]); tm += ev[1].elapsed_time(ev[2])
    print(f"event time: voxelize {tv/R:.3f} ms, maps {tm/R:.3f} ms")
    for k, v in acc.items():
        print(f"{k:18s} calls/step {len(v)/R:5.1f}  host ms/step {1e3*sum(v)/R:7.3f}  per call us {1e6*sum(v)/len(v):7.1f}")
