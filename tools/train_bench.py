"""Training step timing (the training benchmark case): B scenes per GPU, spunet34c+ppt, 300 captions per scene with
random point subsets and random unit text rows, fwd + bwd + gradient all-reduce + SGD.
usage: python tools/train_bench.py [scenes_per_gpu] [target_voxels] [steps] [precision]   (torchrun for several ranks)
precision: how forward and input-gradient convolutions form their products (f32 | bf16x3 | bf16x6; default bf16x3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from segmentation_amd import ops, scenes, sparse, spunet, train

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
TV = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
PREC = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
sparse.set_precision(PREC)
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
dist = None
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl")
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
torch.manual_seed(0)
cfg = spunet.SPUNET34C
net = spunet.PPT(spunet.SpUNet(in_channels=3, out_channels=768, channels=cfg["channels"], layers=cfg["layers"])).to(dev)
rng = np.random.default_rng(100 + rank)
coords, feats, offs, embs, idxs = [], [], [0], [], []
for b in range(B):
    sc = scenes.make_scene(seed=1000 * rank + b, target_voxels=TV)
    coords.append(sc["coord"]); feats.append(sc["feat"]); offs.append(offs[-1] + sc["coord"].shape[0])
    n = sc["coord"].shape[0]
    uniq = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(150, 768)).astype(np.float32)), dim=1).to(dev)
    embs.append([uniq[int(rng.integers(0, 150))] for _ in range(300)])
    idxs.append([torch.from_numpy(rng.choice(n, size=int(rng.integers(100, 5000)), replace=False)) for _ in range(300)])
pi, co, npc = train.flatten_caption_indices(idxs, torch.tensor(offs))
batch = dict(coord=torch.from_numpy(np.concatenate(coords)).to(dev), feat=torch.from_numpy(np.concatenate(feats)).to(dev),
             offset=torch.tensor(offs), grid_size=0.02, condition=["ScanNet"],
             caption_data=dict(embedding=embs, point_indices=pi.to(dev), caption_offsets=co, num_points_per_caption=npc))
step = train.TrainStep(net, total_steps=1000, dist=dist)
for _ in range(2):
    loss = step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    loss = step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / STEPS
with torch.no_grad():
    nvox = ops.voxelize(batch["coord"], batch["feat"], batch["offset"], 0.02)["n_voxels"]
if rank == 0:
    print(f"train step [{PREC}]: {B} scenes/GPU x {world} GPU, {nvox} voxels/GPU, {batch['coord'].shape[0]} points/GPU: {dt * 1e3:.1f} ms/step, "
          f"{nvox * world / dt / 1e6:.2f} M voxels/s, loss {loss:.4f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
if dist is not None:
    dist.destroy_process_group()
