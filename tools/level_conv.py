"""One convolution on level L of the benchmark scene under forced tile plans:
python3 tools/level_conv.py LEVEL[d|u] CIN COUT "wm,wn,nt,nz" ["wm,wn,nt,nz" ...]   ("-" = the library's own plan).
LEVEL alone is the submanifold conv on that level; "Lh" a dense (k = 1) layer on its rows; "Ld" the strided conv L -> L+1, "Lu" the inverse conv L+1 -> L."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import ops, scenes

kind = sys.argv[1][-1] if sys.argv[1][-1] in "duh" else "s"
level, cin, cout = int(sys.argv[1].rstrip("duh")), int(sys.argv[2]), int(sys.argv[3])
plans = sys.argv[4:] or ["-"]
os.environ["M3D_PLAN"] = "0,0,0,0"  # sizes the workspace for any split count; an invalid plan is ignored
dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "bf16x3")
eng, model, cfg = bench.build_engine(dev, prec)
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
vox = ops.voxelize(torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n]), 0.02)
levels = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], 1)
lv = levels[level]
sched, N, n_in, k = {"s": (lv.subm, lv.n, lv.n, 27), "d": (lv.down, lv.n_next, lv.n, 8), "u": (lv.up, lv.n, lv.n_next, 8),
                     "h": (None, lv.n, lv.n, 1)}[kind]
x = torch.randn(n_in, cin, device=dev)
if os.environ.get("F32IN", "0") == "0":  # F32IN=1: plain f32 rows (the split, if any, happens inside the kernel)
    x = ops.split_f32(x, ops.TERMS[prec])
w = ops.pack_weight(torch.randn(cout, k, cin) * 0.03, prec, dev)
ref = None
for plan in plans:
    os.environ["M3D_PLAN"] = "0,0,0,0" if plan == "-" else plan
    out = ops.conv_fwd(x, w, sched, N)
    out = out[0] if isinstance(out, tuple) else out
    if ref is None:
        ref = out.clone()
    err = ((out - ref).norm() / ref.norm()).item()
    reps = 20
    for _ in range(5):
        ops.conv_fwd(x, w, sched, N)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.conv_fwd(x, w, sched, N)
    e1.record(); torch.cuda.synchronize()
    print(f"{prec} level {level}{kind} n={N} {cin}->{cout} plan {plan:10s} {e0.elapsed_time(e1) / reps * 1e3:7.1f} us  rel diff {err:.2e}", flush=True)
