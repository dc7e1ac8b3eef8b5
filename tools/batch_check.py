"""Four scenes in one batch (340 k voxels): the default split-bf16 engine (fused head) against the exact-f32 engine."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from segmentation_amd import scenes
dev = torch.device("cuda:0")
b = scenes.make_batch([21, 22, 23, 24], target_voxels=90000)
coord, feat = torch.from_numpy(b["coord"]).to(dev), torch.from_numpy(b["feat"]).to(dev)
outs = {}
for prec in ("bf16x3", "f32"):
    eng, model, cfg = bench.build_engine(dev, prec)
    with torch.no_grad():
        o = eng.predict(coord, feat, b["offset"], 0.02)
    outs[prec] = o
    print(prec, "voxels", o["n_voxels"], "fused head", eng.text_fused is not None)
a, c = outs["bf16x3"]["pred"], outs["f32"]["pred"]
print("points", a.numel(), "label agreement", (a == c).float().mean().item())
assert torch.equal(outs["bf16x3"]["v2p_map"], outs["f32"]["v2p_map"])
assert (a == c).float().mean().item() >= 0.999
