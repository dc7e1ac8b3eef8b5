"""Child process of test_gpu_conv_sf32.py::test_three_workgroups_per_cu_variant: the same large three-term
convolutions, run with whatever M3D_GMAP_TILES the parent put in the environment; the outputs go to an .npz file."""
import sys
import numpy as np
import torch

import oracle
from segmentation_amd import ops, scenes


def run(device, target=150000):
    sc = scenes.make_scene(seed=11, target_voxels=target)
    off = np.array([0, sc["coord"].shape[0]], dtype=np.int64)
    sp = oracle.sparsify(sc["coord"], sc["feat"], off, 0.02)
    idx = torch.from_numpy(sp["voxel_coords"]).to(device)
    n = idx.shape[0]
    pair = ops.subm_rulebook(idx, sp["spatial_shape"])
    sched = ops.build_schedule(pair, n, idx, 1, sp["spatial_shape"], 1)
    g = torch.Generator().manual_seed(5)
    xa, xb = torch.randn(n, 96, generator=g), torch.randn(n, 32, generator=g)
    w1 = torch.randn(96, 3, 3, 3, 96, generator=g) / np.sqrt(9 * 96)
    w2 = torch.randn(96, 3, 3, 3, 128, generator=g) / np.sqrt(9 * 128)
    scale, shift = torch.rand(96, generator=g) + 0.5, torch.randn(96, generator=g)
    res = torch.randn(n, 96, generator=g)
    sa, sb = ops.split_f32(xa.to(device), 3), ops.split_f32(xb.to(device), 3)
    out1 = ops.conv_fwd(sa, ops.pack_weight(w1, "bf16x6", device), sched, n)
    out2, osp = ops.conv_fwd(sa, ops.pack_weight(w2, "bf16x6", device), sched, n, in1=sb, scale=scale.to(device),
                             shift=shift.to(device), residual=res.to(device), relu=True, split_terms=3)
    torch.cuda.synchronize()
    return dict(n=n, pair=pair.cpu().numpy(), xa=xa, xb=xb, w1=w1, w2=w2, scale=scale, shift=shift, res=res,
                out1=out1.cpu(), out2=out2.cpu(), osp=osp.data.cpu())


if __name__ == "__main__":
    r = run(torch.device("cuda:0"))
    np.savez(sys.argv[1], out1=r["out1"].numpy(), out2=r["out2"].numpy(), osp=r["osp"].numpy())
