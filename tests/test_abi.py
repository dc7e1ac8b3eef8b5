"""The C-ABI library loads and exports every symbol include/m3d.h declares; argument validation
returns error codes without touching a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "m3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(m3d_\w+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = _header_symbols()
    for name in ("m3d_voxelize", "m3d_subm_rulebook", "m3d_strided_rulebook", "m3d_build_schedule",
                 "m3d_conv_fwd_f32", "m3d_conv_fwd_sf32", "m3d_split_f32", "m3d_conv_pack_weight", "m3d_argmax_rows", "m3d_gather_labels",
                 "m3d_last_error", "m3d_version"):
        assert name in syms


def test_library_exports_every_declared_symbol():
    from segmentation_amd import _lib
    lib = _lib.load()
    for name in _header_symbols():
        assert hasattr(lib, name), f"{name} declared in include/m3d.h but not exported"
    # and the Python binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    assert lib.m3d_version() >= 1


def test_argument_validation_without_gpu():
    from segmentation_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    rc = lib.m3d_voxelize(null, null, null, 10, 3, 1, 0.02, 128, 0, 0, null, null, null, null, null, null, null, 0, null)
    assert rc == -1 and b"null pointer" in lib.m3d_last_error()
    rc = lib.m3d_conv_fwd_f32(null, 32, 5, null, 0, 0, null, 27, 32, null, null, null, 5, 128, null, null, null, null, 0, null, 32,
                              null, 0, 0, null, 0, null)
    assert rc == -1
    rc = lib.m3d_conv_fwd_sf32(one := ctypes.c_void_p(8), 33, 5, null, 0, 0, one, 27, 32, null, null, null, 5, 128, null,
                               null, null, null, 0, one, 32, null, 2, null, 0, null)
    assert rc == -1 and b"multiples of 32" in lib.m3d_last_error()
    assert lib.m3d_schedule_tile_order(one, 100, one, null) == -1 and b"multiple of 128" in lib.m3d_last_error()
    assert lib.m3d_sf32_bytes(10, 64, 2) == 11 * 64 * 4
    shape = (ctypes.c_int64 * 3)(0, 4, 4)
    one = ctypes.c_void_p(8)  # never dereferenced: the shape check fails first
    rc = lib.m3d_subm_rulebook(one, 5, shape, 1, one, null, 0, null)
    assert rc == -1 and b"spatial_shape" in lib.m3d_last_error()
    assert lib.m3d_schedule_npad(1) == 256 and lib.m3d_schedule_npad(257) == 512
    assert lib.m3d_voxelize_workspace_bytes(1000, 3) > 1000 * 8
    assert lib.m3d_conv_packed_weight_bytes(96, 27, 96, 0) == 96 * 27 * 96 * 4
    assert lib.m3d_conv_packed_weight_bytes(96, 27, 96, 1) == 96 * 27 * 96 * 4
    assert lib.m3d_conv_packed_weight_bytes(96, 27, 96, 2) == 96 * 27 * 96 * 6
    assert lib.m3d_conv_packed_weight_bytes(32, 27, 3, 2) == 32 * 27 * 3 * 4  # direct kernel keeps f32


def test_ops_refuse_cpu_tensors():
    import torch
    from segmentation_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.subm_rulebook(torch.zeros((4, 4), dtype=torch.int32), [8, 8, 8])
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.voxelize(torch.zeros(5, 3), torch.zeros(5, 3), [0, 5], 0.02)
