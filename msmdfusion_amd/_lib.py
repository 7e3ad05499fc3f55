"""ctypes binding of libmsmd_hip.so (C ABI: include/msmd_hip.h).

The library is the product: if it has not been built this module raises at
import -- there is no CPU or PyTorch fallback anywhere in msmdfusion_amd.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C msmdfusion_amd/csrc``.
"""
import ctypes as C
import os

# torch first: its wheel bundles the HIP runtime (torch/lib/libamdhip64.so,
# SONAME libamdhip64.so.7).  Loading libmsmd_hip.so before torch would pull in
# /opt/rocm's copy as a second runtime instance that cannot see torch's
# device context ("no ROCm-capable device is detected" on the first launch).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MSMD_LIB: another build of the same library -- tools/kprof.py loads the instrumented
# `make PROF=1 OUT=../libmsmd_hip_prof.so` one; the product never sets it)
LIB_PATH = os.environ.get("MSMD_LIB") or os.path.join(_HERE, "libmsmd_hip.so")

if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f"{LIB_PATH} is missing: the HIP library is not built. Run "
        "`make -C msmdfusion_amd/csrc` (hipcc, --offload-arch=gfx950). "
        "msmdfusion_amd has no fallback path.")

# MSMD_PYDLL=1 (experiment): keep the interpreter lock across library calls -- no hand-over to
# the other Python thread at every launch; see DESIGN.md 8.5 for what it measured
lib = (C.PyDLL if os.environ.get("MSMD_PYDLL") == "1" else C.CDLL)(LIB_PATH)

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_i64 = C.c_int64
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes).  Mirrors include/msmd_hip.h one to one;
# tests/test_boundary.py checks the two stay in sync.
SIGNATURES = {
    "msmd_status_string": (C.c_char_p, [_i]),
    "msmd_last_launch_error": (C.c_char_p, []),
    "msmd_abi_version": (_i, []),
    "msmd_device_ok": (_i, []),
    "msmd_voxelize_workspace_bytes": (_sz, [_i, _i, _i]),
    "msmd_hard_voxelize": (_i, [_vp, _i, _i, _fp, _fp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "msmd_hard_voxelize_many_workspace_bytes": (_sz, [_vp, _i]),
    "msmd_hard_voxelize_many": (_i, [_vp, _i, _vp, _sz, _vp]),
    "msmd_voxel_mean": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "msmd_rulebook_subm_workspace_bytes": (_sz, [_i]),
    "msmd_rulebook_subm3d": (_i, [_vp, _i, _i, _ip, _ip, _vp, _vp, _sz, _vp]),
    "msmd_rulebook_subm_bitmap_workspace_bytes": (_sz, [_i, _i, _ip]),
    "msmd_rulebook_subm3d_bitmap": (_i, [_vp, _i, _i, _ip, _ip, _vp, _vp, _sz, _vp]),
    "msmd_rulebook_conv_workspace_bytes": (_sz, [_i, _ip]),
    "msmd_rulebook_conv_chain_workspace_bytes": (_sz, [_i, _i, _ip]),
    "msmd_rulebook_conv3d_count_chain": (_i, [_vp, _i, _i, _i, _ip, _ip, _ip, _ip, _vp, _vp, _sz, _vp]),
    "msmd_rulebook_add_conv_chain_workspace_bytes": (_sz, [_i, _i, _ip, _ip]),
    "msmd_rulebook_add_conv_count_chain": (_i, [_vp, _ip, _i, _i, _ip, _ip, _ip, _ip, _ip, _vp, _vp,
                                                _sz, _vp]),
    "msmd_rulebook_conv3d_count": (_i, [_vp, _i, _i, _ip, _ip, _ip, _ip, _vp, _vp, _sz, _vp]),
    "msmd_rulebook_conv3d_fill": (_i, [_vp, _i, _i, _ip, _ip, _ip, _ip, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "msmd_rulebook_pairs_workspace_bytes": (_sz, [_i, _i]),
    "msmd_rulebook_pairs": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "msmd_spconv_packed_weight_elems": (_sz, [_i, _i, _i]),
    "msmd_spconv_pack_weight": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "msmd_spconv_fwd_f32": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "msmd_rulebook_row_masks": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "msmd_rulebook_tile_costs": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    "msmd_rulebook_tiling_workspace_bytes": (_sz, [_i, _i]),
    "msmd_rulebook_tiling": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "msmd_rulebook_plan_workspace_bytes": (_sz, [_i, _i, _i]),
    "msmd_rulebook_plan": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "msmd_rulebook_subm3d_many_workspace_bytes": (_sz, [_vp, _i]),
    "msmd_rulebook_subm3d_many": (_i, [_vp, _i, _vp, _sz, _vp]),
    "msmd_rulebook_plan_many_workspace_bytes": (_sz, [_vp, _i]),
    "msmd_rulebook_plan_many": (_i, [_vp, _i, _vp, _sz, _vp]),
    "msmd_spconv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "msmd_spconv_wgrad_f32": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "msmd_spconv_packed_split_bytes": (_sz, [_i, _i, _i, _i]),
    "msmd_spconv_pack_weight_split": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "msmd_spconv_pack_weight_split_pair": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "msmd_spconv_fwd_split_supported": (_i, [_i, _i, _i]),
    "msmd_spconv_fwd_split_workspace_bytes": (_sz, [_i, _i]),
    "msmd_spconv_fwd_split": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i,
                                   _vp, _sz, _vp, _vp]),
    "msmd_spconv_fwd_split_stats": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "msmd_rulebook_tile_prefix": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "msmd_spconv_fwd_split_tile_rows": (_i, [_i]),
    "msmd_spconv_fwd_split_instantiation": (_i, [_i, _ip]),
    "msmd_spconv_fwd_split_stats_blocks": (_i, [_i, _i]),
    "msmd_spconv_wgrad_split_supported": (_i, [_i, _i]),
    "msmd_spconv_wgrad_split": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "msmd_spconv_pack_weight_split_many": (_i, [_vp, _i, C.c_long, _i, _vp]),
    "msmd_rulebook_pair_segments_ints": (_sz, [_i, _i]),
    "msmd_rulebook_pair_segments": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "msmd_spconv_wgrad_segments_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "msmd_spconv_wgrad_split_segments": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp,
                                              _i, _vp, _sz, _vp]),
    "msmd_rulebook_permute_cols": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "msmd_bn_workspace_bytes": (_sz, [_i, _i]),
    "msmd_bn_act_fwd_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "msmd_bn_act_fwd_from_partials_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _vp,
                                               _vp, _vp, _vp, _i, _vp]),
    "msmd_bn_act_bwd_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "msmd_bn_relu_bwd_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "msmd_dense_scatter_f32": (_i, [_vp, _vp, _i, _i, _i, _ip, _vp, _vp]),
    "msmd_dense_gather_f32": (_i, [_vp, _vp, _i, _i, _i, _ip, _vp, _vp]),
    "msmd_bev_scatter_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _ip, _vp, _i, _i, _vp]),
    "msmd_bev_gather_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _ip, _vp, _i, _i, _vp]),
    "msmd_fg_gather_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, C.c_double, _vp, _i, _vp, _i,
                                _vp, _vp, _vp, _vp, _vp]),
    "msmd_fg_gather_scored_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, C.c_double, _vp, _i, _vp,
                                       _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "msmd_fg_scatter_add_f32": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "msmd_depth_canvas_workspace_bytes": (_sz, [_i, _i, _i]),
    "msmd_boxes_overlap_bev_f32": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "msmd_boxes_iou3d_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "msmd_heatmap_gaussian_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "msmd_gaussian_focal_workspace_bytes": (_sz, [_i64]),
    "msmd_gaussian_focal_f32": (_i, [_vp, _vp, _i64, _f, _vp, _vp, _vp, _sz, _vp]),
    "msmd_depth_canvas_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "msmd_sparse_add_workspace_bytes": (_sz, [_i, _ip]),
    "msmd_sparse_add_count": (_i, [_vp, _i, _vp, _i, _i, _ip, _vp, _vp, _sz, _vp]),
    "msmd_sparse_add_fill": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _ip, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "msmd_sparse_add_rows": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "msmd_rows_inverse": (_i, [_vp, _i, _i, _vp, _vp]),
    "msmd_sparse_add_rows_gather": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "msmd_gma_assemble_fwd_f32": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp,
                                       _i, _i, _vp, _vp]),
    "msmd_gma_assemble_bwd_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp,
                                       _vp, _vp, _vp, _vp, _vp]),
    "msmd_gma_assemble_bwd_workspace_floats": (_sz, [_i]),
    "msmd_rows_linear_supported": (_i, [_i, _i]),
    "msmd_rows_linear_fwd_f32": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "msmd_rows_linear_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "msmd_rows_linear_bwd_f32": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _sz,
                                      _vp]),
    "msmd_modality_split_workspace_bytes": (_sz, [_i, _ip]),
    "msmd_modality_split": (_i, [_vp, _i, _vp, _i, _i, _ip, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "msmd_modality_split_stats": (_i, [_vp, _i, _vp, _i, _i, _ip, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                       _vp]),
    "msmd_modality_split_float_keys_workspace_bytes": (_sz, [_i, _i, _i]),
    "msmd_modality_split_float_keys": (_i, [_vp, _i, _vp, _i, _i, _ip, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                            _vp, _sz, _vp]),
    "msmd_rows_where_workspace_bytes": (_sz, [_i]),
    "msmd_rows_where_eq": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "msmd_rows_where_eq_many_workspace_bytes": (_sz, [_vp, _i]),
    "msmd_rows_where_eq_many": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "msmd_furthest_point_sample": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "msmd_furthest_point_sample_ragged": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "msmd_ball_query": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp]),
    "msmd_nn_search": (_i, [_vp, _i, _vp, _i, _f, _vp, _vp, _vp]),
    "msmd_nn_assign": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == the library is stale
    _fn.restype = _res
    _fn.argtypes = _args

ABI_VERSION = 1
if lib.msmd_abi_version() != ABI_VERSION:
    raise RuntimeError("libmsmd_hip.so ABI version mismatch: rebuild msmdfusion_amd/csrc")


class MsmdError(RuntimeError):
    """A C-ABI call returned a non-zero msmd_status (the reference raises
    RuntimeError from TV_ASSERT_RT_ERR / TORCH_CHECK the same way)."""


def check(status, what=""):
    if status != 0:
        msg = lib.msmd_status_string(status).decode()
        if status == -4:
            msg += " [" + lib.msmd_last_launch_error().decode() + "]"
        raise MsmdError(f"{what}: {msg} (msmd_status {status})")


def int3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def float_arr(v):
    return (C.c_float * len(v))(*[float(x) for x in v])
