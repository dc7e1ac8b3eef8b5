import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segmentation_amd import ops
dev = torch.device("cuda:0")
N = 273080
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cin, cout in [(96, 768), (128, 96), (768, 256)]:
    x = ops.split_f32(torch.randn(N, cin, device=dev), 2)
    w = ops.pack_weight(torch.randn(cout, 1, cin) * 0.05, "bf16x3", dev)
    res = torch.randn(N, cout, device=dev)
    for dbg in (0, 32):
        os.environ["M3D_DBG"] = str(dbg)
        a = t(lambda: ops.conv_fwd(x, w, None, N))
        b = t(lambda: ops.conv_fwd(x, w, None, N, split_terms=2))
        c = t(lambda: ops.conv_fwd(x, w, None, N, want_f32=False, split_terms=2))
        d = t(lambda: ops.conv_fwd(x, w, None, N, residual=res, relu=True, split_terms=2))
        print(f"{cin}->{cout} dbg={dbg}: f32 only {a:7.1f}  f32+sf {b:7.1f}  sf only {c:7.1f}  res+f32+sf {d:7.1f} us")
