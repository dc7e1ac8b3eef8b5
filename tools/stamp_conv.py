"""Cycle shares of the LDS-DMA convolution's stage loop, from the diagnostic build (make -C segmentation_amd/csrc stamp):
M3D_LIB_PATH=segmentation_amd/libm3d_hip_stamp.so PREC=bf16x3 python tools/stamp_conv.py LEVEL CIN COUT"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import ops, scenes, _lib

level, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prec = os.environ.get("PREC", "bf16x3")
dev = torch.device("cuda:0")
eng, model, cfg = bench.build_engine(dev, prec)
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
vox = ops.voxelize(torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n]), 0.02)
levels = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], 1)
lv = levels[level]
x = ops.split_f32(torch.randn(lv.n, cin, device=dev), ops.TERMS[prec])
w = ops.pack_weight(torch.randn(cout, 27, cin) * 0.03, prec, dev)
lib = _lib.load()
fn = ctypes.CDLL(_lib.LIB_PATH).m3d_debug_stamps
for _ in range(3):
    ops.conv_fwd(x, w, lv.subm, lv.n)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
fn(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.conv_fwd(x, w, lv.subm, lv.n); e1.record(); torch.cuda.synchronize()
fn(out, 0)
wait, issue, comp, stages = out[0], out[1], out[2], out[3]
tot = wait + issue + comp
print(f"{prec} level {level} {cin}->{cout}: {e0.elapsed_time(e1)*1e3:.0f} us (stamped build); per wave-stage cycles: "
      f"wait+barrier {wait/stages:.0f} ({100*wait/tot:.0f} %), issue {issue/stages:.0f} ({100*issue/tot:.0f} %), "
      f"matrix {comp/stages:.0f} ({100*comp/tot:.0f} %), total {tot/stages:.0f}; wave-stages {stages}")
