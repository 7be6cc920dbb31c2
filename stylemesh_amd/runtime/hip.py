"""ctypes binding of ``libstylemesh_hip.so`` (the C ABI declared in ``include/stylemesh_hip.h``).

PyTorch supplies device memory and streams only: every call passes raw ``tensor.data_ptr()`` device pointers
and the current HIP stream. There is NO fallback: if the library is missing the import of any product
module raises, and every non-zero return code raises ``RuntimeError`` (the reference surface reports
errors as Python exceptions, SURVEY.md section 8 b).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libstylemesh_hip.so")

ABI_VERSION = 11         # sm_abi_version() of the library this binding was written against
SM_MAX_TEX_LAYERS = 8
SM_FMAP_GUARD = 4096
EPI_BIAS_RELU, EPI_RELU_MASK, EPI_ADD, EPI_POOL, EPI_GRAM = 1, 2, 4, 8, 16
LIST_QUADS = 32          # SM_LIST_QUADS: the tile list holds vertical quads of segments (resident-input kernel)

_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t

# name -> argtypes; must list every symbol include/stylemesh_hip.h declares (tests/test_abi.py checks it)
SIGNATURES = {
    "sm_fmap_row_stride": [_i],
    "sm_fmap_plane": [_i, _i],
    "sm_abi_version": [],
    "sm_sizeof_problem": [_i],
    "sm_tex_sample_fwd": [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp],
    "sm_tex_sample_fwd_grouped": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "sm_tex_sample_bwd": [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp],
    "sm_tex_touch_flags": [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _vp],
    "sm_tex_scatter_plan_temp_bytes": [_sz, _i],
    "sm_tex_scatter_plan_cross_bytes": [_sz],
    "sm_radix_sort_pairs": [_vp, _vp, _vp, _vp, _sz, _i, _vp, _sz, _vp, _vp],
    "sm_tex_scatter_plan": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp, _vp],
    "sm_tex_scatter_planned": [_vp, _vp, _sz, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp],
    "sm_adam_fused": [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _i, _f, _d, _d, _f, _d, _d, _f, _f, _f, _i, _vp, _vp, _vp, _i, _vp],
    "sm_adam_hyper_step": [_vp, _d, _d, _vp, _vp],
    "sm_step_begin": [_vp, _vp, _i, _vp, _vp, _sz, _vp, _sz, _vp],
    "sm_flags_or": [_vp, _vp, _sz, _vp],
    "sm_clamp_sumsq": [_vp, _sz, _vp, _i, _f, _f, _vp, _vp],
    "sm_conv3x3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp],
    "sm_conv3x3_grouped": [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp, _vp],
    "sm_conv3x3_grouped_split2": [_vp, _i, _vp, _f, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp, _vp, _vp],
    "sm_amax_floats": [],
    "sm_fmap_amax": [_vp, _i, _i, _i, _vp, _vp],
    "sm_conv_tile_positions": [_i, _i],
    "sm_conv_split2_tile_positions": [_i],
    "sm_conv3x3_dgrad_c3": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "sm_conv3x3_dgrad_c3_grouped": [_vp, _i, _vp, _i, _vp],
    "sm_maxpool2x2_fwd_grouped": [_vp, _i, _i, _vp],
    "sm_maxpool2x2_bwd_relu_grouped": [_vp, _i, _i, _vp],
    "sm_plane_tile_positions": [_i],
    "sm_conv3x3_dgrad_c3_tiles": [_vp, _i, _vp, _i, _vp, _i, _vp],
    "sm_maxpool2x2_fwd_tiles": [_vp, _i, _i, _vp, _i, _vp],
    "sm_maxpool2x2_bwd_relu_tiles": [_vp, _i, _i, _vp, _i, _vp],
    "sm_maxpool2x2_fwd_codes_tiles": [_vp, _vp, _i, _i, _vp, _i, _vp],
    "sm_maxpool2x2_fwd": [_vp, _vp, _i, _i, _i, _vp],
    "sm_maxpool2x2_bwd_relu": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sm_gram_num_slabs": [_i, _i, _i],
    "sm_gram_workspace_slabs": [_i, _i, _i],
    "sm_gram_masked": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sm_gram_split_num_slabs": [],
    "sm_gram_masked_split": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "sm_gram_masked_split_acc": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "sm_gram_masked_split2_grouped": [_vp, _i, _vp],
    "sm_style_loss_grouped": [_vp, _i, _vp, _vp],
    "sm_gram_backward_split2_grouped": [_vp, _i, _vp],
    "sm_style_loss": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "sm_gram_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sm_mse_masked": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _vp],
    "sm_level_masks": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "sm_level_maps": [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sm_layer_masks": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sm_level_factors": [_vp, _vp, _i, _vp, _vp],
    "sm_need_step": [_vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp],
    "sm_tile_flags": [_vp, _i, _i, _i, _vp, _vp],
    "sm_cover_segments_ws_bytes": [_vp, _i],
    "sm_cover_segments": [_vp, _i, _vp, _sz, _vp],
    "sm_call_id": [C.c_char_p],
    "sm_call_n_args": [_i],
    "sm_call_replay": [_vp, _i, _vp, _vp],
    "sm_copy_floats": [_vp, _vp, _sz, _vp],
    "sm_zero_floats": [_vp, _sz, _vp],
    "sm_view_masks": [_vp, _vp],
    "sm_view_lists_ws_bytes": [_vp],
    "sm_view_lists": [_vp, _vp],
    "sm_fmap_resize_bilinear": [_vp, _i, _i, _i, _vp, _i, _i, _vp],
    "sm_image_to_fmap": [_vp, _i, _i, _i, _vp, _i, _i, _vp],
    "sm_fmap_to_image": [_vp, _i, _i, _i, _vp, _vp],
    "sm_gram_backward_split_ws_bytes": [_i],
    "sm_gram_backward_split": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sm_reproject_blocks": [_i, _i],
    "sm_reproject": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp],
    "sm_raster_maps": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "sm_mip_downsample": [_vp, _vp, _i, _i, _i, _vp],
    "sm_tex_sample_mip": [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp],
    "sm_flags_compact_ws_ints": [_sz],
    "sm_flags_compact": [_vp, _sz, _vp, _vp, _vp, _vp],
    "sm_chunks_gather": [_vp, _vp, _vp, _sz, _i, _vp, _vp],
    "sm_chunks_scatter": [_vp, _vp, _vp, _sz, _i, _vp, _f, _vp],
    "sm_stream_create_cu_subset": [_i, _vp],
    "sm_stream_destroy": [_vp],
    "sm_comm_unique_id_bytes": [],
    "sm_comm_get_unique_id": [_vp],
    "sm_comm_init": [_vp, _i, _vp, _i],
    "sm_comm_destroy": [_vp],
    "sm_allreduce_grad": [_vp, _vp, _sz, _vp],
    "sm_allreduce_flags_max": [_vp, _vp, _sz, _vp],
    "sm_comm_info": [_vp, _vp],
    "sm_device_link": [_i, _i, _vp, _vp],
}


# entry points that are pure host functions (sizes, layout constants, ids): no stream, nothing to replay
PURE_HOST = {"sm_fmap_row_stride", "sm_fmap_plane", "sm_abi_version", "sm_sizeof_problem", "sm_tex_scatter_plan_temp_bytes",
             "sm_tex_scatter_plan_cross_bytes", "sm_amax_floats", "sm_conv_tile_positions",
             "sm_conv_split2_tile_positions", "sm_plane_tile_positions", "sm_gram_num_slabs", "sm_gram_workspace_slabs",
             "sm_gram_split_num_slabs", "sm_gram_backward_split_ws_bytes", "sm_reproject_blocks", "sm_flags_compact_ws_ints",
             "sm_comm_unique_id_bytes", "sm_cover_segments_ws_bytes", "sm_view_lists_ws_bytes", "sm_call_id",
             "sm_call_n_args", "sm_device_link"}


class ConvProblem(C.Structure):
    """sm_conv_problem of include/stylemesh_hip.h"""
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("gate", C.c_void_p), ("H", C.c_int), ("W", C.c_int),
                ("unpool_code", C.c_void_p), ("pool_out", C.c_void_p), ("pool_code", C.c_void_p),
                ("gram_ws", C.c_void_p), ("gram_mask0", C.c_void_p), ("gram_mask1", C.c_void_p),
                ("gram_amax_feat", C.c_void_p), ("gram_amax_d", C.c_void_p)]


class GramProblem(C.Structure):
    """sm_gram_problem of include/stylemesh_hip.h"""
    _fields_ = [("feat", C.c_void_p), ("mask0", C.c_void_p), ("mask1", C.c_void_p), ("S0", C.c_void_p),
                ("S1", C.c_void_p), ("amax_feat", C.c_void_p), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int)]


class StyleProblem(C.Structure):
    """sm_style_problem of include/stylemesh_hip.h"""
    _fields_ = [("S0", C.c_void_p), ("S1", C.c_void_p), ("counts", C.c_void_p), ("factor", C.c_void_p),
                ("targets", C.c_void_p * 4), ("term_mask", C.c_int * 4), ("n_terms", C.c_int),
                ("skip_if_empty", C.c_int * 2), ("weight", C.c_float), ("C", C.c_int), ("D0", C.c_void_p),
                ("D1", C.c_void_p), ("history", C.c_void_p), ("hist_len", C.c_int), ("hist_slot", C.c_int),
                ("n_slabs", C.c_int), ("amax_d_out", C.c_void_p)]


class GramBwdProblem(C.Structure):
    """sm_gram_bwd_problem of include/stylemesh_hip.h"""
    _fields_ = [("feat", C.c_void_p), ("mask0", C.c_void_p), ("mask1", C.c_void_p), ("D0", C.c_void_p),
                ("D1", C.c_void_p), ("dfeat", C.c_void_p), ("ws", C.c_void_p), ("amax_feat", C.c_void_p),
                ("amax_d", C.c_void_p), ("amax_out", C.c_void_p), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("relu_gate", C.c_int)]


class CoverProblem(C.Structure):
    """sm_cover_problem of include/stylemesh_hip.h"""
    _fields_ = [("need", C.c_void_p), ("starts", C.c_void_p), ("count", C.c_void_p), ("h", C.c_int), ("w", C.c_int),
                ("tag", C.c_int), ("cap", C.c_int), ("pair_w", C.c_int), ("quad", C.c_int)]


VIEW_MAX_LEVELS, VIEW_MAX_LAYERS, VIEW_MAX_LISTS = 8, 24, 48


class ViewLevel(C.Structure):
    """sm_view_level of include/stylemesh_hip.h"""
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("has_maps", C.c_int), ("M", C.c_void_p), ("pixel_weight", C.c_void_p),
                ("passed", C.c_void_p), ("m_sum", C.c_void_p)]


class ViewLayerMask(C.Structure):
    """sm_view_layer_mask"""
    _fields_ = [("level", C.c_int), ("loss_layer", C.c_int), ("hl", C.c_int), ("wl", C.c_int), ("mask_planes", C.c_void_p),
                ("counts", C.c_void_p), ("factor", C.c_void_p)]


class ViewResize(C.Structure):
    """sm_view_resize"""
    _fields_ = [("src", C.c_void_p), ("C", C.c_int), ("h", C.c_int), ("w", C.c_int), ("dst", C.c_void_p), ("H", C.c_int),
                ("W", C.c_int)]


class ViewMasksDesc(C.Structure):
    """sm_view_masks_desc"""
    _fields_ = [("mask", C.c_void_p), ("angle_guidance", C.c_void_p), ("angle_degrees", C.c_void_p), ("rounded", C.c_void_p),
                ("other", C.c_void_p), ("interp_w", C.c_void_p), ("h", C.c_int), ("w", C.c_int), ("angle_threshold", C.c_float),
                ("E", C.c_void_p), ("Wt", C.c_void_p), ("n_levels", C.c_int), ("levels", ViewLevel * VIEW_MAX_LEVELS),
                ("n_masks", C.c_int), ("masks", C.POINTER(ViewLayerMask)), ("n_resizes", C.c_int),
                ("resizes", C.POINTER(ViewResize))]


class ViewList(C.Structure):
    """sm_view_list"""
    _fields_ = [("layer", C.c_int), ("mode", C.c_int), ("bn", C.c_int), ("pair_layer", C.c_int), ("group", C.c_int),
                ("out", C.c_void_p), ("cap", C.c_int), ("staging", C.c_void_p), ("staging_cap", C.c_int)]


class ViewListsDesc(C.Structure):
    """sm_view_lists_desc"""
    _fields_ = [("n_levels", C.c_int), ("M", C.c_void_p * VIEW_MAX_LEVELS), ("H", C.c_int * VIEW_MAX_LEVELS),
                ("W", C.c_int * VIEW_MAX_LEVELS), ("n_layers", C.c_int), ("node_is_pool", C.c_int * VIEW_MAX_LAYERS),
                ("node_src", C.c_int * VIEW_MAX_LAYERS), ("injected", C.c_int * VIEW_MAX_LAYERS),
                ("need", (C.c_void_p * VIEW_MAX_LAYERS) * VIEW_MAX_LEVELS), ("lh", (C.c_int * VIEW_MAX_LAYERS) * VIEW_MAX_LEVELS),
                ("lw", (C.c_int * VIEW_MAX_LAYERS) * VIEW_MAX_LEVELS), ("n_lists", C.c_int), ("lists", C.POINTER(ViewList)),
                ("summary", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t)]


CALL_MAX_ARGS = 24


class Call(C.Structure):
    """sm_call of include/stylemesh_hip.h"""
    _fields_ = [("fn", C.c_int), ("n_args", C.c_int), ("skip", C.c_int), ("reserved", C.c_int),
                ("args", C.c_uint64 * CALL_MAX_ARGS)]


class PlaneProblem(C.Structure):
    """sm_plane_problem of include/stylemesh_hip.h"""
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("out", C.c_void_p), ("H", C.c_int),
                ("W", C.c_int)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or stylemesh_amd/csrc/build.sh")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _sz if name.endswith(("_bytes", "_ws_ints")) else _i
    return lib


lib = _load()
if lib.sm_abi_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH} has ABI version {lib.sm_abi_version()}, this binding needs {ABI_VERSION}: rebuild it "
                      "(stylemesh_amd/csrc/build.sh)")


def check(code: int, what: str = ""):
    if code != 0:
        raise RuntimeError(f"libstylemesh_hip: {what} failed with HIP error code {code}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> int:
    """The current HIP stream's handle. Called once per kernel launch: ``torch.cuda.current_stream().cuda_stream`` builds a
    Stream object every time (10 us, 0.4 ms of host time per single-level step); the raw getter is a plain C call."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    """Device pointer of a tensor / FMap (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    if hasattr(t, "ptr"):
        return t.ptr
    assert t.is_cuda and t.is_contiguous(), "device, contiguous tensors only"
    return t.data_ptr()


def ptr_array(ptrs):
    return (C.c_void_p * len(ptrs))(*[ptr(p) for p in ptrs])


def int_array(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def float_array(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


def size_array(vals):
    return (C.c_size_t * len(vals))(*[int(v) for v in vals])


def row_stride(W: int) -> int:
    return lib.sm_fmap_row_stride(W)


def plane(H: int, W: int) -> int:
    return lib.sm_fmap_plane(H, W)
