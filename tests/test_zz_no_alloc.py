"""m3d_bn_stats_static (no allocation, partial sums in a fixed device array of the library) against
m3d_bn_stats (caller's workspace), through the C ABI: two calls in a row, the second with a smaller input,
bit-identical outputs; sizes beyond the fixed array are refused without a launch."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _vp(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _both(lib, x):
    n, c = x.shape
    dev = x.device
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = lib.m3d_bn_workspace_bytes(n, c)
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    m0, v0 = torch.empty(c, device=dev), torch.empty(c, device=dev)
    assert lib.m3d_bn_stats(_vp(x), n, c, _vp(m0), _vp(v0), _vp(ws), nb, stream) == 0, lib.m3d_last_error()
    m1, v1 = torch.full((c,), float("nan"), device=dev), torch.full((c,), float("nan"), device=dev)
    rc = lib.m3d_bn_stats_static(_vp(x), n, c, _vp(m1), _vp(v1), stream)
    assert rc == 0, lib.m3d_last_error()
    torch.cuda.synchronize()
    return (m0, v0), (m1, v1)


def test_static_scratch_matches_workspace_entry_point(device):
    from segmentation_amd import _lib
    lib = _lib.load()
    torch.cuda.set_device(device)
    g = torch.Generator(device="cpu").manual_seed(3)
    assert lib.m3d_bn_stats_static_max_bytes() == 4 << 20
    # first call: many workgroups (1024 partial rows of 256 channels = the whole array); second call: fewer rows
    # and fewer channels, so stale partials of the first call sit next to the live ones
    for n, c in ((300_000, 256), (7_001, 96), (5, 32)):
        x = (torch.randn((n, c), generator=g) * 3 + 1.5).to(device)
        (m0, v0), (m1, v1) = _both(lib, x)
        assert torch.equal(m0, m1) and torch.equal(v0, v1), (n, c)
        ref_m = x.double().mean(0)
        ref_v = x.double().var(0, unbiased=False)
        assert torch.allclose(m1.double(), ref_m, rtol=1e-6, atol=1e-6)
        assert torch.allclose(v1.double(), ref_v, rtol=1e-5, atol=1e-6)


def test_static_scratch_refuses_sizes_beyond_the_array(device):
    from segmentation_amd import _lib
    lib = _lib.load()
    torch.cuda.set_device(device)
    n, c = 400_000, 512  # 1024 partial rows x 512 channels x 16 B = 8 MiB > 4 MiB
    assert lib.m3d_bn_workspace_bytes(n, c) > lib.m3d_bn_stats_static_max_bytes()
    x = torch.zeros((8, c), device=device)  # never read: the size check comes first
    m = torch.full((c,), 7.0, device=device)
    v = torch.full((c,), 7.0, device=device)
    rc = lib.m3d_bn_stats_static(_vp(x), n, c, _vp(m), _vp(v), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -2 and b"static scratch" in lib.m3d_last_error()
    torch.cuda.synchronize()
    assert float(m.min()) == 7.0 and float(v.max()) == 7.0


def test_static_scratch_rejects_invalid_arguments_without_touching_the_outputs(device):
    from segmentation_amd import _lib
    lib = _lib.load()
    torch.cuda.set_device(device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.ones((16, 32), device=device)
    m = torch.full((32,), 7.0, device=device)
    v = torch.full((32,), 7.0, device=device)
    for n, c in ((0, 32), (-3, 32), (16, 30), (16, 0)):   # no rows; a channel count that is not a multiple of 4 / zero
        rc = lib.m3d_bn_stats_static(_vp(x), n, c, _vp(m), _vp(v), stream)
        assert rc < 0 and lib.m3d_last_error(), (n, c)
    rc = lib.m3d_bn_stats_static(ctypes.c_void_p(0), 16, 32, _vp(m), _vp(v), stream)
    assert rc < 0
    torch.cuda.synchronize()
    assert float(m.min()) == 7.0 and float(v.max()) == 7.0
    # and the entry point still works afterwards
    (m0, v0), (m1, v1) = _both(lib, x)
    assert torch.equal(m0, m1) and torch.equal(v0, v1)
