"""Where the host time of one scene goes (cProfile over the enqueue of 20 scenes, GPU work left to run behind).
usage: python3 tools/host_profile.py [precision]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import scenes
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
eng, model, cfg = bench.build_engine(dev, prec)
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
one = (torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n]), 0.02)
with torch.no_grad():
    eng.predict_many([one] * 4, net_streams=2)
    pr = cProfile.Profile()
    pr.enable()
    eng.predict_many([one] * 20, net_streams=2)
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
