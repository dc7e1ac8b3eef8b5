"""Soak of the scene pipeline: a long sequence of scenes of very different sizes through predict_many (lists and a
generator of fresh tensors), every result compared bit for bit with the same scene through predict():
python3 tools/soak_pipeline.py [n_scenes] [precision]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import scenes

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 120
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x6"
dev = torch.device("cuda:0")
eng, model, cfg = bench.build_engine(dev, prec)
rng = np.random.default_rng(7)
sizes = [2000, 5000, 12000, 30000, 60000, 90000, 130000, 180000, 220000, 273000, 800, 45000]
pool = []
for i, t in enumerate(sizes):
    sc = scenes.make_scene(seed=50 + i, target_voxels=t)
    pool.append((sc["coord"], sc["feat"], np.array([0, sc["coord"].shape[0]], dtype=np.int64)))
dpool = [(torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev), o) for c, f, o in pool]
with torch.no_grad():
    ref = [eng.predict(c, f, o, bench.GRID)["pred"].clone() for c, f, o in dpool]
    torch.cuda.synchronize()
    order = rng.integers(0, len(pool), size=n_seq)
    t0 = time.perf_counter()
    outs = eng.predict_many([(dpool[i][0], dpool[i][1], dpool[i][2], bench.GRID) for i in order], net_streams=3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bad = [k for k, i in enumerate(order) if not torch.equal(outs[k]["pred"], ref[i])]
    print(f"list: {n_seq} scenes in {dt * 1e3:.1f} ms, mismatching scenes: {bad}", flush=True)

    def fresh():
        for i in order:   # new device tensors per scene, dropped by the caller as soon as they are handed over
            c, f, o = pool[i]
            yield torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev), o, bench.GRID

    t0 = time.perf_counter()
    outs = eng.predict_many(fresh(), net_streams=3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bad2 = [k for k, i in enumerate(order) if not torch.equal(outs[k]["pred"], ref[i])]
    print(f"generator: {n_seq} scenes in {dt * 1e3:.1f} ms, mismatching scenes: {bad2}", flush=True)
    again = [eng.predict(c, f, o, bench.GRID)["pred"] for c, f, o in dpool]
    bad3 = [i for i in range(len(pool)) if not torch.equal(again[i], ref[i])]
    print("predict() afterwards, mismatching scenes:", bad3, flush=True)
sys.exit(1 if bad or bad2 or bad3 else 0)
