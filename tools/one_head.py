"""The fused head alone on benchmark-sized input, for counter collection: [PREC=bf16x3|bf16x6|f16x3] python3 tools/one_head.py [reps] [n]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segmentation_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 273080
ncls = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "bf16x3")
x = ops.split_f32(torch.randn(n, 96, device=dev), ops.TERMS[prec])
w1 = ops.pack_weight(torch.randn(768, 1, 96) * 0.1, prec, dev)
tp = ops.head_pack_text(torch.nn.functional.normalize(torch.randn(ncls, 768, device=dev), dim=1), cin=96,
                        terms=ops.TERMS[prec])
shift = torch.randn(768, device=dev) * 0.1
ops.head_fused(x, w1, tp, shift=shift)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.head_fused(x, w1, tp, shift=shift)
e1.record(); torch.cuda.synchronize()
print(f"{prec} head n={n} classes={ncls} (padded {tp.classes_padded}): {e0.elapsed_time(e1) / reps * 1e3:.1f} us")
