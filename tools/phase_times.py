"""Wall-clock split of one step: voxelize / kernel maps / backbone / last layer + head (each synchronised)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from segmentation_amd import ops, scenes
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
eng, model, cfg = bench.build_engine(dev, prec)
sc = scenes.make_scene(seed=0, target_voxels=bench.TARGET_VOXELS)
n = sc["coord"].shape[0]
coord, feat = torch.from_numpy(sc["coord"]).to(dev), torch.from_numpy(sc["feat"]).to(dev)
off = np.array([0, n])
def sync(): torch.cuda.synchronize()
with torch.no_grad():
    for _ in range(3): eng.predict(coord, feat, off, 0.02)
    T = {"vox": 0, "maps": 0, "net": 0, "head": 0, "total": 0}
    R = 10
    for _ in range(R):
        sync(); t0 = time.perf_counter()
        vox = ops.voxelize(coord, feat, off, 0.02, pad=128); sync(); t1 = time.perf_counter()
        levels = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], vox["batch_size"]); sync(); t2 = time.perf_counter()
        other = eng.text_fused is not None and eng._head_terms != eng._terms   # the head's format is not the backbone's
        x = eng.backbone(vox["voxel_feats"], levels, last_f32=other); sync(); t3 = time.perf_counter()
        if eng.text_fused is not None:   # last layer + logits + argmax in one kernel
            fin = eng.final_head if eng.final_head is not None else eng.final
            xs = ops.split_f32(x[0], eng._head_terms) if other else x[1]
            vl, _ = ops.head_fused(xs, fin.w, eng.text_fused, scale=fin.scale, shift=fin.shift)
        else:
            pair = eng._conv(eng.final, x, None, levels[0].n, need_f32=True, need_sf=eng._terms != 0)
            src = pair[1] if eng._terms else pair[0]
            logits = ops.conv_fwd(src, eng.text.w, None, levels[0].n)
            vl = ops.argmax_rows(logits, eng.n_classes)
        pl = ops.gather_labels(vl, vox["v2p_map"]); sync(); t4 = time.perf_counter()
        T["vox"] += t1 - t0; T["maps"] += t2 - t1; T["net"] += t3 - t2; T["head"] += t4 - t3
        sync(); t5 = time.perf_counter(); eng.predict(coord, feat, off, 0.02); sync(); T["total"] += time.perf_counter() - t5
    print({k: round(v / R * 1e3, 3) for k, v in T.items()}, "ms")
    # host-only cost of enqueueing the network (no sync inside): time to return from backbone()
    sync(); t0 = time.perf_counter(); x = eng.backbone(vox["voxel_feats"], levels); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    print(f"backbone enqueue {1e3*(t1-t0):.2f} ms, until done {1e3*(t2-t0):.2f} ms")
