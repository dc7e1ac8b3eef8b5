"""The large-scene benchmark case: one scene (~2 M voxels at 1 cm, 15 x 15 x 3 m room), C = 200, D = 768.
Prints throughput, level sizes and peak memory; checks labels against the exact-f32 engine."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import scenes
dev = torch.device("cuda:0")
tv = int(sys.argv[1]) if len(sys.argv) > 1 else 2_300_000
sc = scenes.make_scene(seed=0, target_voxels=tv, grid_size=0.01, room=(15.0, 15.0, 3.0))
n = sc["coord"].shape[0]
coord, feat, off = torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n])
res = {}
for prec in ("bf16x3", "f32"):
    eng, _, _ = bench.build_engine(dev, prec)
    with torch.no_grad():
        out = eng.predict(coord, feat, off, 0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = eng.predict(coord, feat, off, 0.01)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    res[prec] = out["pred"].cpu()
    print(f"{prec}: {n} points, levels {eng.last_stats['n_voxels']}, {dt * 1e3:.1f} ms, {out['n_voxels'] / dt / 1e6:.1f} M voxels/s, "
          f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    del eng
print("label agreement bf16x3 vs f32:", float((res["bf16x3"] == res["f32"]).float().mean()))
