"""Timing-only ablations of one big SubM layer (results are wrong when M3D_DBG != 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from segmentation_amd import ops, scenes
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
cin, cout = 96, 96
sc = scenes.make_scene(seed=0, target_voxels=290000)
n = sc["coord"].shape[0]
vox = ops.voxelize(torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n]), 0.02)
idx = vox["voxel_coords"]; N = idx.shape[0]
pair = ops.subm_rulebook(idx, vox["spatial_shape"])
SMODE = int(os.environ.get("SMODE", "1539"))
sched = ops.build_schedule(pair, N, idx, SMODE, vox["spatial_shape"], 1)
x = torch.randn(N, cin, device=dev); w = ops.pack_weight(torch.randn(cout, 27, cin) * 0.03, prec, dev)
xd = torch.randn(N, 768, device=dev); wd = ops.pack_weight(torch.randn(256, 1, 768) * 0.03, prec, dev)
if len(sys.argv) > 2 and sys.argv[2] == "dma":
    T = {"bf16x3": 2, "bf16x6": 3}[prec]
    x, xd = ops.split_f32(x, T), ops.split_f32(xd, T)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
CASES = [(0, "full"), (1, "no MFMA"), (2, "no A loads"), (4, "no W loads"), (6, "no A,W loads"),
         (1 | 2, "no MFMA, no A"), (1 | 4, "no MFMA, no W"), (1 | 2 | 4, "skeleton")]
if len(sys.argv) > 2 and sys.argv[2] == "dma":
    CASES = [(0, "full"), (1, "no compute"), (2, "A from zero row"), (4, "W from one tile"), (6, "A zero, W one"),
             (8, "A lanes masked"), (8 | 4, "A masked, W one"), (1 | 2 | 4, "no compute, A zero, W one"), (1 | 8, "no compute, A masked")]
for dbg, name in CASES:
    os.environ["M3D_DBG"] = str(dbg)
    a = t(lambda: ops.conv_fwd(x, w, sched, N))
    b = t(lambda: ops.conv_fwd(xd, wd, None, N))
    print(f"{prec} dbg={dbg:2d} {name:22s} subm96x96: {a:8.1f} us   dense768x256: {b:8.1f} us")
