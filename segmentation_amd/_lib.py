"""ctypes binding of libm3d_hip.so (the C ABI declared in include/m3d.h).

The library is the product: if it is missing the import fails loudly; there is no CPU or
PyTorch fallback for any operator."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M3D_LIB_PATH") or os.path.join(_HERE, "libm3d_hip.so")  # override: experiment builds

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_sz = ctypes.c_size_t
_f32 = ctypes.c_float

# name -> (restype, argtypes); mirrors include/m3d.h one to one
SIGNATURES = {
    "m3d_version": (_int, []),
    "m3d_last_error": (ctypes.c_char_p, []),
    "m3d_voxelize_workspace_bytes": (_sz, [_i64, _int]),
    "m3d_voxelize": (_int, [_vp, _vp, _vp, _i64, _int, _int, _f32, _int, _int, _int,
                            _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "m3d_voxel_pick_random_workspace_bytes": (_sz, [_i64]),
    "m3d_voxel_pick_random": (_int, [_vp, _vp, _i64, _int, _i64, ctypes.c_uint64, _vp, _vp, _vp, _sz, _vp]),
    "m3d_grid_sample_workspace_bytes": (_sz, [_i64]),
    "m3d_grid_sample": (_int, [_vp, _i64, ctypes.c_double, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "m3d_stray_convert": (_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "m3d_center_shift_workspace_bytes": (_sz, [_i64]),
    "m3d_center_shift_normalize": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "m3d_backproject_workspace_bytes": (_sz, [_int, _int, _int]),
    "m3d_backproject": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, ctypes.POINTER(ctypes.c_double),
                               ctypes.POINTER(ctypes.c_double), ctypes.c_double, ctypes.c_double, _int, _int, _vp, _vp,
                               _i64, _vp, _vp, _sz, _vp]),
    "m3d_subm_rulebook_workspace_bytes": (_sz, [_i64, ctypes.POINTER(_i64), _int]),
    "m3d_subm_rulebook": (_int, [_vp, _i64, ctypes.POINTER(_i64), _int, _vp, _vp, _sz, _vp]),
    "m3d_strided_rulebook_workspace_bytes": (_sz, [_i64, ctypes.POINTER(_i64), _int]),
    "m3d_strided_rulebook": (_int, [_vp, _i64, _vp, ctypes.POINTER(_i64), _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "m3d_schedule_npad": (_i64, [_i64]),
    "m3d_schedule_workspace_bytes": (_sz, [_i64]),
    "m3d_build_schedule": (_int, [_vp, _int, _i64, _i64, _vp, ctypes.POINTER(_i64), _int, _int, _vp,
                                  _vp, _vp, _vp, _vp, _sz, _vp]),
    "m3d_conv_workspace_bytes": (_sz, [_i64, _int, _int]),
    "m3d_conv_fwd_f32": (_int, [_vp, _int, _i64, _vp, _int, _i64, _vp, _int, _int, _vp, _vp, _vp, _i64, _i64, _vp,
                                _vp, _vp, _vp, _int, _vp, _i64, _vp, _int, _int, _vp, _sz, _vp]),
    "m3d_schedule_tile_order": (_int, [_vp, _i64, _vp, _vp]),
    "m3d_conv_fwd_sf32": (_int, [_vp, _int, _i64, _vp, _int, _i64, _vp, _int, _int, _vp, _vp, _vp, _i64, _i64, _vp,
                                 _vp, _vp, _vp, _int, _vp, _i64, _vp, _int, _vp, _sz, _vp]),
    "m3d_conv_bwd_weight_workspace_bytes": (_sz, [_i64, _int, _int, _int]),
    "m3d_conv_bwd_weight_f32": (_int, [_vp, _int, _vp, _int, _int, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "m3d_conv_bwd_weight_bf16x3": (_int, [_vp, _int, _vp, _int, _int, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "m3d_bn_workspace_bytes": (_sz, [_i64, _int]),
    "m3d_bn_stats": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _sz, _vp]),
    "m3d_f16_status": (_int, [_vp, _int, _vp]),
    "m3d_bn_act_fwd": (_int, [_vp, _i64, _int, _vp, _vp, _f32, _vp, _vp, _vp, _int, _vp, _vp]),
    "m3d_bn_act_backward_reduce": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _sz, _vp]),
    "m3d_bn_act_backward": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "m3d_bn_stats_static_max_bytes": (_sz, []),
    "m3d_bn_stats_static": (_int, [_vp, _i64, _int, _vp, _vp, _vp]),
    "m3d_bn_backward_reduce": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _sz, _vp]),
    "m3d_sf32_bytes": (_sz, [_i64, _int, _int]),
    "m3d_split_f32": (_int, [_vp, _i64, _int, _int, _vp, _vp]),
    "m3d_conv_packed_weight_bytes": (_sz, [_int, _int, _int, _int]),
    "m3d_conv_pack_weight": (_int, [_vp, _int, _int, _int, _int, _vp, _vp]),
    "m3d_conv_pack_weight_transposed": (_int, [_vp, _int, _int, _int, _int, _int, _vp, _vp]),
    "m3d_argmax_rows": (_int, [_vp, _i64, _int, _i64, _vp, _vp]),
    "m3d_gather_labels": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "m3d_avgpool_fwd": (_int, [_vp, _int, _vp, _int, _i64, _i64, _vp, _vp, _vp]),
    "m3d_head_text_bytes": (_sz, [_int, _int]),
    "m3d_head_pack_text": (_int, [_vp, _int, _int, _int, _vp, _vp]),
    "m3d_head_text_bytes_terms": (_sz, [_int, _int, _int]),
    "m3d_head_pack_text_terms": (_int, [_vp, _int, _int, _int, _int, _vp, _vp]),
    "m3d_head_fused_terms": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _int, _vp, _int, _int, _vp, _i64, _vp, _vp]),
    "m3d_head_fused_sf32": (_int, [_vp, _int, _i64, _vp, _vp, _vp, _int, _vp, _int, _int, _vp, _i64, _vp, _vp]),
    "m3d_net_create": (_int, [_vp, _int, _vp, _int, _int, ctypes.POINTER(_vp)]),
    "m3d_net_destroy": (None, [_vp]),
    "m3d_net_arena_bytes": (_sz, [_vp, ctypes.POINTER(_i64)]),
    "m3d_net_run": (_int, [_vp, _vp, ctypes.POINTER(_i64), _vp, _vp, _sz, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "m3d_segment_csr": (_int, [_vp, _int, _vp, _i64, _int, _vp, _vp]),
    "m3d_scatter_rows": (_int, [_vp, _vp, _i64, _int, _i64, _int, _vp, _vp, _vp]),
    "m3d_levels_workspace_bytes": (_sz, [_vp, _int, _int]),
    "m3d_levels_begin": (_int, [_vp, _int, _int, _int, _vp, _vp, _sz, _vp]),
    "m3d_levels_finish": (_int, [_vp, _int, _int, ctypes.c_uint, _vp, _sz, _vp]),
}

_i32 = ctypes.c_int32


class NetValue(ctypes.Structure):      # m3d_net_value_t
    _fields_ = [("level", _i32), ("channels", _i32), ("format", _i32)]


class NetLayer(ctypes.Structure):      # m3d_net_layer_t
    _fields_ = [("op", _i32), ("in0", _i32), ("in1", _i32), ("residual", _i32), ("out_f32", _i32), ("out_split", _i32),
                ("sched", _i32), ("k", _i32), ("cout", _i32), ("c0", _i32), ("c1", _i32), ("relu", _i32),
                ("precision", _i32), ("n_classes", _i32), ("classes_padded", _i32),
                ("weight", _vp), ("scale", _vp), ("shift", _vp), ("text", _vp)]


class NetSched(ctypes.Structure):      # m3d_net_sched_t
    _fields_ = [("nbr", _vp), ("out_row", _vp), ("tile_mask", _vp), ("tile_order", _vp), ("n", _i64), ("npad", _i64),
                ("k", _i32)]


class NetMaps(ctypes.Structure):       # m3d_net_maps_t
    _fields_ = [("subm", NetSched), ("down", NetSched), ("up", NetSched)]

_lib = None


def load():
    """Load the shared library once and declare every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C segmentation_amd/csrc`.  segmentation_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class M3DError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = load().m3d_last_error().decode("utf-8", "replace")
        raise M3DError(f"{what} failed (code {rc}): {msg}")


class LevelSched(ctypes.Structure):    # the schedule outputs inside m3d_level_t
    _fields_ = [("nbr", _vp), ("out_row", _vp), ("tile_mask", _vp), ("tile_order", _vp)]


class Level(ctypes.Structure):         # m3d_level_t
    _fields_ = [("indices", _vp), ("n", _i64), ("shape", _i64 * 3), ("next_indices", _vp), ("pair_fwd", _vp),
                ("pair_bwd", _vp), ("ld", _i64), ("pair_subm", _vp), ("subm_mode", _i32),
                ("subm", LevelSched), ("down", LevelSched), ("up", LevelSched)]
