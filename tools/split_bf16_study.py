"""CPU study: error of split-bf16 (2- and 3-term) emulated f32 matmuls through the whole SpUNet."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from oracle import spunet as S, sparse_conv as SC
from segmentation_amd import scenes

def split(x, terms):
    parts = []
    r = x.clone()
    for _ in range(terms):
        p = r.to(torch.bfloat16).to(torch.float32)
        parts.append(p)
        r = r - p
    return parts

def make_mm(terms, prods):
    def mm(a, b):  # a [n,k], b [k,m]
        A, B = split(a, terms), split(b, terms)
        out = torch.zeros(a.shape[0], b.shape[1])
        for i, j in prods:
            out += A[i] @ B[j]
        return out
    return mm

def patched_native(mm):
    def _native(feats, weight, pair, n_out):
        cout, cin = weight.shape[0], weight.shape[-1]
        w = weight.reshape(cout, -1, cin)
        out = torch.zeros(n_out, cout, dtype=feats.dtype)
        pair_t = torch.as_tensor(np.asarray(pair), dtype=torch.int64)
        for k in range(w.shape[1]):
            p = pair_t[k]; o = torch.nonzero(p >= 0).squeeze(1)
            if o.numel() == 0: continue
            out.index_add_(0, o, mm(feats.index_select(0, p[o]), w[:, k, :].T.contiguous()))
        return out
    return _native

cfg = S.SPUNET34C
sc = scenes.make_scene(1, 12000); n = len(sc['coord']); off = np.array([0, n])
sp = oracle.sparsify(sc['coord'], sc['feat'], off, 0.02)
lv = S.build_rulebooks(sp['voxel_coords'], sp['spatial_shape'])
for trained in (True, False):
    P = S.make_spunet_params(cfg, 0, trained_like=trained)
    vf = torch.from_numpy(sp['voxel_feats'])
    ref64 = None
    orig_native, orig_1x1 = SC._native, SC.conv1x1
    ref = S.spunet_forward(P, cfg, vf, sp['voxel_coords'], sp['spatial_shape'], levels=lv)
    # float64 truth
    P64 = {k: v.double() for k, v in P.items()}
    SC._native = lambda f, w, p, n_out: orig_native(f, w, p, n_out)
    truth = S.spunet_forward(P64, cfg, vf.double(), sp['voxel_coords'], sp['spatial_shape'], levels=lv) if False else None
    for name, terms, prods in (("bf16x3", 2, [(0,0),(0,1),(1,0)]),
                               ("bf16x6", 3, [(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)])):
        mm = make_mm(terms, prods)
        SC._native = patched_native(mm)
        S.subm_conv.__globals__['_native'] = SC._native
        def c1(feats, weight, bias=None, mm=mm):
            out = mm(feats, weight.reshape(weight.shape[0], weight.shape[-1]).T.contiguous())
            return out if bias is None else out + bias
        S.conv1x1 = c1
        S._block.__globals__['conv1x1'] = c1
        out = S.spunet_forward(P, cfg, vf, sp['voxel_coords'], sp['spatial_shape'], levels=lv)
        rel = ((out - ref).norm(dim=1) / ref.norm(dim=1)).numpy()
        emb = torch.from_numpy(scenes.text_embeddings(200, 768))
        agree = ((out @ emb.T).argmax(1) == (ref @ emb.T).argmax(1)).float().mean().item()
        print(f"trained_like={trained} {name}: per-voxel rel err max={rel.max():.3e} median={np.median(rel):.3e} label agree={agree:.5f}")
        SC._native = orig_native
        S.subm_conv.__globals__['_native'] = orig_native
        S.conv1x1 = orig_1x1; S._block.__globals__['conv1x1'] = orig_1x1
