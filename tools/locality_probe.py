"""How much of a level-0 convolution's time is the cost of missing in L2: the same kernel map with every neighbour
index replaced by one inside a window of W rows (same patterns, tile masks and work; gathers then hit in cache).
usage: PREC=bf16x6 python3 tools/locality_probe.py CIN COUT [W ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import ops, scenes

cin, cout = int(sys.argv[1]), int(sys.argv[2])
windows = [int(v) for v in sys.argv[3:]] or [0, 1024, 8192, 65536]
dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "bf16x6")
eng, model, cfg = bench.build_engine(dev, prec)
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
vox = ops.voxelize(torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n]), 0.02)
idx, shape = vox["voxel_coords"], list(vox["spatial_shape"])
N = idx.shape[0]
pair = ops.subm_rulebook(idx, shape, 1)
x = ops.split_f32(torch.randn(N, cin, device=dev), ops.TERMS[prec])
w = ops.pack_weight(torch.randn(cout, 27, cin) * 0.03, prec, dev)
for W in windows:
    p = pair
    if W:
        fake = (torch.arange(N, device=dev, dtype=pair.dtype) % W)[None, :].expand_as(pair)
        p = torch.where(pair >= 0, fake, pair).contiguous()
    sched = eng._subm_schedule(0, p, N, idx, shape, 1)
    for _ in range(5):
        ops.conv_fwd(x, w, sched, N)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        ops.conv_fwd(x, w, sched, N)
    e1.record(); torch.cuda.synchronize()
    print(f"{prec} level 0 {cin}->{cout} window {W or 'real map':>8}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
