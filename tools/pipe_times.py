"""Throughput of consecutive scenes: one at a time (predict) against the two-stage pipeline (predict_many).
usage: python3 tools/pipe_times.py [precision] [scenes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import scenes
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng, model, cfg = bench.build_engine(dev, prec)
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
one = (torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n]), 0.02)
with torch.no_grad():
    for _ in range(3):
        eng.predict(*one)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        eng.predict(*one)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.predict_many([one] * 3)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    eng.predict_many([one] * K)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    eng.predict_many([one] * 4, net_streams=2)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    eng.predict_many([one] * K, net_streams=2)
    torch.cuda.synchronize(); t5 = time.perf_counter()
    eng.predict_many([one] * 6, net_streams=3)
    torch.cuda.synchronize(); t6 = time.perf_counter()
    eng.predict_many([one] * K, net_streams=3)
    torch.cuda.synchronize(); t7 = time.perf_counter()
print(f"{prec}: three network streams {(t7 - t6) / K * 1e3:.3f} ms/scene")
print(f"{prec}: one at a time {(t1 - t0) / K * 1e3:.3f} ms/scene, pipelined {(t3 - t2) / K * 1e3:.3f} ms/scene, "
      f"two network streams {(t5 - t4) / K * 1e3:.3f} ms/scene")
