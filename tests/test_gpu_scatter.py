"""torch_scatter.scatter / segment_csr as the reference's code calls them (structure.py:58, v1m3:410,
caption_loss.py:223), served on the GPU by m3d_scatter_rows / m3d_segment_csr: values and gradients against a direct
float64 restatement on the CPU; segment_csr additionally run-to-run identical (fixed summation order)."""
import numpy as np
import pytest
import torch

from segmentation_amd import sparse

pytestmark = pytest.mark.gpu


def _segment_ref(src, indptr, mean):
    out = []
    for s in range(len(indptr) - 1):
        seg = src[int(indptr[s]):int(indptr[s + 1])]
        out.append(seg.mean(0) if (mean and len(seg)) else seg.sum(0))
    return torch.stack(out)


@pytest.mark.parametrize("reduce", ["mean", "sum"])
@pytest.mark.parametrize("c", [3, 64, 200, 600])
def test_segment_csr(device, reduce, c):
    rng = np.random.default_rng(c)
    counts = np.array([5, 0, 1, 700, 33, 0, 2049, 64, 1])
    indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]))
    src = torch.from_numpy(rng.normal(size=(int(counts.sum()) + 7, c)).astype(np.float32))   # rows past indptr[-1] unused
    x = src.to(device).requires_grad_(True)
    out = sparse.segment_csr(x, indptr, reduce=reduce)
    ref_in = src.double().requires_grad_(True)
    ref = _segment_ref(ref_in, indptr, reduce == "mean")
    # (float32 sums of up to 2049 rows)
    assert out.shape == ref.shape and torch.allclose(out.double().cpu(), ref, rtol=2e-4, atol=2e-4)
    g = torch.from_numpy(rng.normal(size=tuple(ref.shape))).float()
    out.backward(g.to(device))
    ref.backward(g.double())
    assert torch.allclose(x.grad.double().cpu(), ref_in.grad, rtol=1e-6, atol=1e-7)
    again = sparse.segment_csr(x.detach(), indptr.to(device), reduce=reduce)
    assert torch.equal(again, out.detach())


@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_scatter_rows(device, reduce):
    rng = np.random.default_rng(3)
    n, c, n_out = 20000, 3, 3500
    index = torch.from_numpy(np.sort(rng.integers(0, n_out - 40, size=n)))     # sorted, as mean_pooling passes it; a tail unused
    src = torch.from_numpy(rng.normal(size=(n, c)).astype(np.float32))
    x = src.to(device).requires_grad_(True)
    out = sparse.scatter(x, index.to(device), dim=0, reduce=reduce, dim_size=n_out)
    ref_in = src.double().requires_grad_(True)
    ref = torch.zeros(n_out, c, dtype=torch.float64).index_add_(0, index, ref_in)
    cnt = torch.bincount(index, minlength=n_out).clamp(min=1).double().unsqueeze(1)
    if reduce == "mean":
        ref = ref / cnt
    assert torch.allclose(out.double().cpu(), ref, rtol=1e-4, atol=1e-4)
    g = torch.from_numpy(rng.normal(size=(n_out, c))).float()
    out.backward(g.to(device))
    ref.backward(g.double())
    assert torch.allclose(x.grad.double().cpu(), ref_in.grad, rtol=1e-6, atol=1e-7)
    # dim_size inferred from the index, and an unsorted index
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    out2 = sparse.scatter(src[perm].to(device), index[perm].to(device), reduce=reduce)
    assert out2.shape[0] == int(index.max()) + 1
    assert torch.allclose(out2.double().cpu(), ref[:out2.shape[0]].detach(), rtol=1e-4, atol=1e-4)
