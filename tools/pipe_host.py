import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from segmentation_amd import scenes
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
eng, model, cfg = bench.build_engine(dev, prec)
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
one = (torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n]), 0.02)
for ns in [int(x) for x in os.environ.get("NS", "2,3").split(",")]:
    eng.predict_many([one] * 4, net_streams=ns)
    torch.cuda.synchronize()
    eng.pipe_log = []
    t0 = time.perf_counter()
    eng.predict_many([one] * 30, net_streams=ns)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30 * 1e3
    log = eng.pipe_log
    a = np.array([(x[0], x[1]) for x in log[5:-2]]) * 1e3
    base = log[0][2]
    spans = [(base.elapsed_time(x[2]), base.elapsed_time(x[3])) for x in log]
    print("network of scene i on the GPU (ms from the first network's start): " +
          "  ".join(f"{i}:{b:.1f}-{e:.1f}" for i, (b, e) in enumerate(spans[8:16], 8)))
    dur = np.array([e - b for b, e in spans[5:-2]])
    gap = np.array([spans[i + 1][0] - spans[i][0] for i in range(5, len(spans) - 3)])
    print(f"network duration {dur.mean():.2f} ms, start-to-start {gap.mean():.2f} ms")
    print(f"{prec} net_streams={ns}: {dt:.2f} ms/scene; host per scene: network enqueue {a[:,0].mean():.2f}, first stage of next {a[:,1].mean():.2f} ms")
