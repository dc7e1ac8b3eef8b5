"""Only the pipelined sequence of scenes (for a kernel trace): python3 tools/pipe_trace.py [precision] [scenes] [net_streams].
Analyse the trace with tools/pipe_busy.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import scenes
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 2
eng, model, cfg = bench.build_engine(dev, prec)
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
one = (torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n]), 0.02)
with torch.no_grad():
    eng.predict_many([one] * 4, net_streams=ns)
    torch.cuda.synchronize()
    time.sleep(0.2)  # a gap in the trace: what follows is the steady sequence
    t0 = time.perf_counter()
    eng.predict_many([one] * K, net_streams=ns)
    torch.cuda.synchronize()
    print(f"{prec}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms/scene over {K} scenes, {ns} network streams")
