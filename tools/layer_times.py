"""Per-launch table of the convolution kernel over one forward (HIP events on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import ops, scenes

dev = torch.device("cuda:0")
eng, model, cfg = bench.build_engine(dev, sys.argv[2] if len(sys.argv) > 2 else "f32")
tv = int(sys.argv[1]) if len(sys.argv) > 1 else bench.TARGET_VOXELS
sc = scenes.make_scene(seed=0, target_voxels=tv)
n = sc["coord"].shape[0]
coord, feat = torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev)
off = np.array([0, n])
with torch.no_grad():
    for _ in range(2):
        eng.predict(coord, feat, off, 0.02)
    ops.PROFILE = []
    eng.predict(coord, feat, off, 0.02)
    ops.PROFILE = []
    reps = 3
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        eng.predict(coord, feat, off, 0.02)
    t1.record()
    torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
per = len(prof) // reps
print(f"N0={eng.last_stats['n_voxels']} step(with events)={t0.elapsed_time(t1)/reps:.2f} ms")
tot_t = tot_f = 0
for i in range(per):
    ts = [prof[i + r * per][2].elapsed_time(prof[i + r * per][3]) for r in range(reps)]
    kind, fl, _, _, shp, _ = prof[i]
    t = float(np.median(ts))
    tot_t += t; tot_f += fl
    print(f"{i:3d} {kind:6s} n={shp[0]:7d} cin={shp[1]:4d} cout={shp[2]:4d} k={shp[3]:2d}  {t*1e3:8.1f} us  {fl/1e9:7.2f} GF  {fl/t/1e9:7.1f} TF/s")
print(f"conv total {tot_t:.2f} ms, {tot_f/1e9:.1f} GF, {tot_f/tot_t/1e9:.1f} TF/s")
