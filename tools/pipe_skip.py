"""What the convolutions of the small levels cost the pipelined sequence: the same 30-scene sequence with every
convolution of at most SKIP output rows left out (its outputs stay uninitialised; timing only, results are garbage).
Usage: python3 tools/pipe_skip.py [precision]   (thresholds from the environment: SKIPS=0,1000,5000,20000,100000)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from segmentation_amd import ops, scenes, _lib

dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
eng, model, cfg = bench.build_engine(dev, prec)
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
one = (torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev), np.array([0, n]), 0.02)
real = ops.conv_fwd
skip = [0]
left_out = [0]


def conv_fwd(in0, weight_packed, sched, n_out, in1=None, scale=None, shift=None, residual=None, relu=False, out=None,
             flop_cout=None, want_f32=True, split_terms=0):
    if n_out > skip[0]:
        return real(in0, weight_packed, sched, n_out, in1, scale, shift, residual, relu, out, flop_cout, want_f32, split_terms)
    left_out[0] += 1
    cout = weight_packed.cout
    if out is None and want_f32:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    if not split_terms:
        return out
    sp = ops.SplitTensor(torch.empty((_lib.load().m3d_sf32_bytes(n_out, cout, split_terms),), dtype=torch.uint8, device=dev),
                         n_out, cout, split_terms)
    return out, sp


ops.conv_fwd = conv_fwd
for s in [int(x) for x in os.environ.get("SKIPS", "0,1000,5000,20000,100000").split(",")]:
    skip[0] = s
    eng.predict_many([one] * 4, net_streams=3)
    torch.cuda.synchronize()
    left_out[0] = 0
    t0 = time.perf_counter()
    eng.predict_many([one] * 30, net_streams=3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30 * 1e3
    print(f"{prec}: convolutions of <= {s} rows left out ({left_out[0] // 30} per scene): {dt:.2f} ms per scene", flush=True)
