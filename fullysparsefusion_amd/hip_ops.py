"""Thin torch-tensor wrappers, one per C-ABI entry point of libfsf_hip.so (include/fsf_hip.h).

These functions only marshal pointers/sizes and allocate caller-owned outputs with torch; all arithmetic
happens in the HIP kernels.  The reference-shaped Python API (scatter_v2, Voxelization, SparseConvTensor,
...) is built on top of these in `fullysparsefusion_amd.mmdet3d_plugin`.
"""
import ctypes
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib, switches
from ._lib import FsfHipError, c_f32, c_i32, c_i64, c_p, check, f32_array, i32_array, i64_array, ptr, require_cuda, stream_ptr

ERR_KEY_RANGE = -3  # FSF_ERR_KEY_RANGE (include/fsf_hip.h)

_P = c_p
_ARGTYPES = {
    "fsf_assemble_sweeps_workspace_bytes": [c_i64],
    "fsf_assemble_sweeps": [_P, c_i64, c_i32, _P, c_i32, _P, _P, _P, c_f32, _P, c_i32, c_f32, c_f32, _P, _P, _P, _P, c_i64, _P],
    "fsf_voxelize_dynamic": [_P, c_i64, c_i32, c_i32, _P, _P, _P, _P, _P, _P],
    "fsf_vfe_decorate": [_P, c_i64, c_i32, c_i32, _P, c_i32, _P, _P, _P, _P, c_i32, c_i32, _P, c_i32, _P],
    "fsf_vote_centers_keys": [_P, c_i32, _P, c_i32, _P, c_i32, _P, _P, _P, c_i64, c_i32, c_i32, _P, _P, _P, c_i32, _P, _P, _P, _P],
    "fsf_voxelize_divfloor": [_P, c_i64, c_i32, _P, _P, c_i32, _P, _P, _P],
    "fsf_unique_rows_workspace_bytes": [c_i64, c_i32],
    "fsf_unique_rows": [_P, c_i64, c_i32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_i64, _P],
    "fsf_segment_plan_workspace_bytes": [c_i64, c_i64],
    "fsf_segment_plan_from_inverse": [_P, c_i64, c_i64, _P, _P, _P, _P, c_i64, _P],
    "fsf_segment_reduce_workspace_bytes": [c_i64, c_i64, c_i32],
    "fsf_segment_reduce": [_P, c_i64, c_i64, c_i32, _P, _P, _P, c_i64, c_i32, _P, _P, _P, c_i64, _P],
    "fsf_segment_reduce_short": [_P, _P, _P, c_i32, c_i64, _P, _P, c_i64, c_i32, _P, _P, _P],
    "fsf_segment_reduce_backward": [_P, c_i64, c_i32, _P, _P, c_i64, c_i32, _P, _P, _P],
    "fsf_gather_rows": [_P, c_i64, c_i32, _P, c_i64, _P, c_i64, _P],
    "fsf_gather_rows_strided": [_P, c_i64, c_i64, c_i32, _P, c_i64, _P, c_i64, _P],
    "fsf_gather_rows_add": [_P, c_i64, c_i64, c_i32, _P, c_i64, _P, c_i64, _P, c_i64, _P],
    "fsf_norm_act": [_P, c_i64, c_i32, _P, _P, c_f32, c_i32, c_i32, _P, c_i64, _P],
    "fsf_voxel2point": [_P, c_i32, _P, _P, c_i64, c_i32, _P, c_i64, _P, _P, c_f32, _P, _P, _P],
    "fsf_voxel2point_strided": [_P, c_i32, _P, _P, c_i64, c_i32, _P, c_i64, _P, _P, c_f32, _P, c_i64, _P, _P],
    "fsf_project_gather_mask": [_P, c_i64, c_i32, _P, c_i32, _P, c_i32, c_i32, c_i32, c_i32, _P, _P, _P],
    "fsf_cam_select_score": [_P, c_i64, c_i32, c_i32, _P, c_i32, c_i32, c_i32, _P, _P, _P],
    "fsf_project_score": [_P, c_i64, c_i32, _P, c_i32, _P, c_i32, c_i32, c_i32, c_i32, _P, c_i32, c_i32, c_i32, _P, _P, _P, _P, _P, _P],
    "fsf_concat_mul": [_P, c_i64, c_i32, _P, _P, c_i64, c_i32, _P, c_i64, c_i32, c_f32, _P, c_i64, _P, _P],
    "fsf_concat_mul_backward": [_P, c_i64, c_i32, _P, _P, c_i64, c_i32, _P, c_i64, c_i32, c_f32, _P, _P, c_i64, _P, _P, _P, _P],
    "fsf_group_pairs_workspace_bytes": [c_i64, c_i32],
    "fsf_group_pairs": [_P, c_i64, c_i32, c_i64, _P, c_i32, _P, c_i32, _P, _P, c_i64, _P, _P, c_i64, _P],
    "fsf_overlap_plan_workspace_bytes": [c_i64],
    "fsf_overlap_plan": [_P, _P, c_i64, c_i32, _P, _P, c_i64, _P],
    "fsf_overlap_rows": [_P, c_i64, c_i32, _P, c_i32, _P, c_i32, c_i32, c_i32, c_i32, _P, _P, _P, c_i64, c_i64, c_i64, c_i64, _P, _P, _P],
    "fsf_project_gather_bilinear": [_P, c_i64, c_i32, _P, c_i32, _P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, _P, _P, _P],
    "fsf_rulebook_workspace_bytes": [c_i64, c_i32],
    "fsf_rulebook_subm": [_P, c_i64, c_i32, _P, _P, _P, _P, _P, c_i64, _P],
    "fsf_rulebook_strided": [_P, c_i64, c_i32, _P, _P, _P, _P, _P, _P, c_i64, _P, _P, _P, _P, _P, c_i64, _P],
    "fsf_rulebook_to_pairs_workspace_bytes": [c_i64, c_i32],
    "fsf_rulebook_to_pairs": [_P, c_i64, c_i32, _P, c_i64, _P, _P, c_i64, _P],
    "fsf_spconv_transpose_weight": [_P, c_i32, c_i32, c_i32, _P, _P],
    "fsf_spconv_workspace_bytes": [c_i64, c_i32, c_i32, c_i32],
    "fsf_spconv_forward": [_P, c_i64, c_i32, _P, c_i32, c_i32, _P, c_i64, _P, _P, _P, c_i32, _P, _P, c_i64, _P],
    "fsf_spconv_backward_weight_workspace_bytes": [c_i64, c_i32, c_i32, c_i32],
    "fsf_spconv_backward_weight": [_P, c_i64, c_i32, _P, c_i64, c_i32, _P, _P, c_i64, c_i32, _P, _P, c_i64, _P],
    "fsf_connected_components_workspace_bytes": [c_i64],
    "fsf_connected_components": [_P, c_i64, c_i32, _P, c_f32, _P, _P, _P, c_i64, _P],
    "fsf_column_stats_workspace_bytes": [c_i32],
    "fsf_column_stats": [_P, c_i64, c_i32, _P, _P, _P, c_i64, _P],
    "fsf_batch_norm_train_stats": [_P, c_i64, c_i32, _P, _P, c_f32, c_f32, c_f32, c_f32, _P, _P, _P, _P, _P, _P, _P, _P, c_i64, _P],
    "fsf_batch_norm_act_forward": [_P, c_i64, c_i32, _P, _P, c_i32, _P, _P],
    "fsf_batch_norm_act_backward": [_P, _P, c_i64, c_i32, _P, _P, _P, _P, c_i32, _P, _P, _P, _P, c_i64, _P],
    "fsf_norm_act_backward_workspace_bytes": [c_i32],
    "fsf_norm_act_backward": [_P, _P, c_i64, c_i32, _P, _P, c_f32, c_i32, _P, _P, _P, _P, c_i64, _P],
    "fsf_row_topk_desc": [_P, c_i64, c_i32, c_i32, _P, _P],
    "fsf_sir_input_gather": [_P, c_i64, c_i32, _P, _P, _P, _P, c_i32, _P, c_i32, _P, c_i64, c_i32, c_f32, _P, c_i64, c_i32, c_f32, _P, _P, _P, c_i32,
                             _P, _P, _P, c_i32, _P, _P, _P, c_f32, c_i32, c_i64, _P, c_i64, _P],
    "fsf_sir_input": [_P, c_i64, c_i32, _P, _P, c_i64, c_i32, _P, c_i64, c_i32, c_f32, _P, c_i64, c_i32, c_f32, _P, _P, _P, c_i32,
                      _P, _P, _P, c_i32, _P, _P, _P, c_f32, c_i32, c_i64, _P, c_i64, _P],
    "fsf_spconv_split_weight_bytes": [c_i32, c_i32, c_i32],
    "fsf_spconv_prepare_weight_split": [_P, c_i32, c_i32, c_i32, _P, _P],
    "fsf_spconv_split_workspace_bytes": [c_i64, c_i32, c_i32, c_i32],
    "fsf_spconv_forward_split": [_P, c_i64, c_i32, _P, c_i32, c_i32, _P, c_i64, _P, _P, _P, c_i32, _P, _P, c_i64, _P],
    "fsf_planes_bytes": [c_i64, c_i32],
    "fsf_planes_scale_count": [c_i64, c_i32],
    "fsf_to_planes": [_P, c_i64, c_i32, c_i64, _P, _P, _P],
    "fsf_to_planes_rows": [_P, c_i64, c_i32, c_i64, _P, _P, _P, _P],
    "fsf_spconv_planes_weight_bytes": [c_i32, c_i32, c_i32],
    "fsf_spconv_prepare_weight_planes": [_P, c_i32, c_i32, c_i32, _P, _P],
    "fsf_spconv_forward_planes": [_P, _P, c_i32, _P, _P, c_i32, c_i64, _P, c_i32, c_i32, _P, c_i64, _P, _P, _P, c_i32, _P, _P, _P,
                                  _P],
    "fsf_channel_group_sum_add": [_P, c_i64, c_i32, c_i32, _P, _P, _P],
    "fsf_channel_pair_sum_add2": [_P, c_i32, _P, c_i32, c_i64, _P, _P, _P],
    "fsf_channel_pair_sum_add2_planes": [_P, c_i32, _P, c_i32, c_i64, _P, _P, _P, _P],
    "fsf_linear_prepared_weight_bytes": [c_i32, c_i32],
    "fsf_linear_prepare_weight": [_P, c_i32, c_i32, _P, _P],
    "fsf_linear_norm_act": [_P, c_i64, c_i32, c_i64, _P, c_i32, _P, c_i32, _P, _P, c_f32, c_i32, _P, c_i64, _P],
    "fsf_linear_prepared_weight_sliced_bytes": [c_i32, c_i32, c_i32],
    "fsf_linear_prepare_weight_sliced": [_P, c_i32, c_i32, c_i32, _P, _P],
    "fsf_linear_norm_act_sliced": [_P, c_i64, c_i32, c_i64, c_i64, _P, c_i32, c_i32, _P, c_i32, _P, _P, c_f32, c_i32, _P, c_i64, _P],
    "fsf_linear_norm_act_grouped": [_P, c_i64, c_i32, c_i64, _P, c_i32, _P, _P, _P, c_i64, c_i32, _P, _P, c_f32, c_i32, _P,
                                    c_i64, _P],
    "fsf_linear_f16w_norm_act_grouped": [_P, c_i64, c_i32, c_i64, _P, c_i32, _P, _P, _P, c_i64, c_i32, _P, _P, c_f32, c_i32, _P,
                                         c_i64, _P],
    "fsf_linear_f16w_norm_act_segmax": [_P, c_i64, c_i32, c_i64, _P, c_i32, _P, _P, _P, c_i64, c_i32, _P, _P, c_f32, c_i32, _P, c_i64,
                                        _P, c_i64, _P, c_i64, _P],
    "fsf_dynamic_point_pool_workspace_bytes": [c_i64, c_i64],
    "fsf_dynamic_point_pool": [_P, c_i64, c_i32, c_i32, c_i32, _P, c_i64, c_i32, _P, _P, c_i32, c_i64, _P, _P, _P, _P, _P,
                               _P, c_i64, _P],
    "fsf_nms_bev_workspace_bytes": [c_i64],
    "fsf_nms_bev": [_P, c_i64, c_f32, c_i32, _P, _P, _P, _P, c_i64, _P],
    "fsf_nms_bev_multiclass_workspace_bytes": [c_i64, c_i32],
    "fsf_nms_bev_multiclass": [_P, c_i64, c_i32, _P, _P, c_f32, c_i32, _P, _P, _P, c_i64, _P],
    "fsf_nms_bev_multiclass_capped": [_P, c_i64, c_i32, _P, _P, c_f32, c_i32, c_i64, _P, _P, _P, _P, c_i64, _P],
    "fsf_nms_bev_multiclass_capped_workspace_bytes": [c_i64, c_i32, c_i64],
    "fsf_decode_cluster_boxes": [_P, c_i64, _P, c_i64, _P, c_i64, c_i64, c_i32, c_i32, c_f32, _P, _P, _P, _P],
    "fsf_class_rank_desc_workspace_bytes": [c_i64, c_i32],
    "fsf_class_rank_desc": [_P, c_i64, c_i32, c_f32, _P, _P, _P, _P, c_i64, _P],
    "fsf_linear_norm_act_segmax": [_P, c_i64, c_i32, c_i64, _P, c_i32, _P, _P, _P, c_i64, c_i32, _P, _P, c_f32, c_i32, _P, c_i64,
                                   _P, c_i64, _P, c_i64, _P],
    "fsf_nms_select_capacity": [],
    "fsf_box_tail_max_classes": [],
    "fsf_nms_select": [_P, c_i32, _P, _P, _P, c_i64, _P, c_i64, c_i32, c_i64, c_i32, _P, _P, _P, _P, _P],
    "fsf_connected_components_grouped": [_P, c_i64, c_i32, _P, _P, c_i32, _P, _P, _P, c_i64, _P],
    "fsf_cluster_key_survival_workspace_bytes": [c_i64, c_i64],
    "fsf_cluster_key_survival": [_P, c_i32, _P, c_i64, _P, c_i64, c_i64, c_i64, c_i32, _P, _P, _P, _P, _P, _P, c_i64, _P],
    "fsf_cluster_point_ids": [_P, _P, c_i64, _P, _P, _P, c_i64, c_i32, _P, _P, c_i64, _P],
    "fsf_ingroup_rank_workspace_bytes": [c_i64],
    "fsf_ingroup_rank": [_P, c_i64, _P, _P, c_i64, _P],
    "fsf_order_by_neighbor_mask_workspace_bytes": [c_i64],
    "fsf_order_by_neighbor_mask": [_P, c_i64, c_i32, _P, _P, _P, _P, c_i64, _P],
    "fsf_remap_indices": [_P, c_i64, _P, _P, _P],
    "fsf_set_option": [c_i32, c_i64],
    "fsf_get_option": [c_i32],
    "fsf_row_planes_bytes": [c_i64, c_i32],
    "fsf_rows_to_planes": [_P, c_i64, c_i32, c_i64, c_i32, _P, _P, c_f32, c_i32, _P, _P, _P, c_i64, _P],
    "fsf_linear_prepared_weight_f16_bytes": [c_i32, c_i32, c_i32],
    "fsf_linear_prepare_weight_f16": [_P, c_i32, c_i32, c_i32, _P, _P],
    "fsf_linear_planes_norm_act": [_P, _P, c_i64, c_i32, _P, c_i32, c_i32, _P, c_i32, _P, _P, c_f32, c_i32, _P, c_i64, _P],
    "fsf_spconv_split_weight_f16_bytes": [c_i32, c_i32, c_i32],
    "fsf_spconv_prepare_weight_split_f16": [_P, c_i32, c_i32, c_i32, _P, _P],
    "fsf_spconv_forward_split_planes": [_P, _P, c_i64, c_i32, _P, c_i32, c_i32, _P, c_i64, _P, _P, _P, c_i32, _P, _P, c_i64, _P],
    "fsf_sir_stack_arena_bytes": [_P, c_i32, c_i64, c_i64],
    "fsf_sir_stack_forward": [_P, c_i32, _P, c_i64, c_i32, _P, _P, _P, c_i32, _P, c_i32, _P, c_i64, c_i32, c_f32, _P, c_i64, c_i32, _P, c_i64,
                              c_i64, _P, c_i64, _P, _P, c_i64, _P],
    "fsf_sorted_rows": [_P, _P, c_i64, _P, c_i64, c_i32, _P, c_i64, _P, c_i64, _P, _P, _P, _P, _P, _P, c_i64, c_f32, _P],
    "fsf_compact_pairs": [_P, c_i64, _P, c_i64, _P, _P, _P, _P, _P, _P, c_i64, _P, _P, _P, _P, _P],
    "fsf_combine_queries": [_P, c_i64, _P, c_i64, _P, _P, _P, c_i32, c_i64, _P, _P, _P, _P],
    "fsf_decode_rois": [_P, c_i64, c_i32, _P, c_i64, _P, c_i64, c_i64, c_f32, _P, _P],
    "fsf_refine_rows": [_P, _P, c_i64, c_i32, _P, _P, _P, c_i64, c_i64, _P, _P, _P],
    "fsf_encode_preds_2d": [_P, c_i64, c_i32, _P, c_i64, c_i32, c_f32, c_f32, _P, _P, c_i64, _P],
    "fsf_weighted_xyz": [_P, c_i64, _P, c_i64, c_f32, _P, _P],
    "fsf_centroid_divide": [_P, c_i64, _P, _P],
    "fsf_lidar_cluster_frontend_arena_bytes": [c_i64, c_i32, c_i32],
    "fsf_lidar_cluster_frontend": [_P, c_i64, c_i32, c_i64, _P, c_i32, _P, _P, c_i32, _P, c_i32, _P, c_i32, c_i32, _P, _P, _P, _P, _P, c_i64, _P,
                                   _P, c_i64, _P, _P],
}
_configured = False


def _L():
    global _configured
    h = _lib.lib()
    if not _configured:
        for name, argtypes in _ARGTYPES.items():
            getattr(h, name).argtypes = argtypes
        _configured = True
    return h


def _rows_view(t):
    """(tensor, row stride in elements) for a 2-D tensor whose rows are contiguous (a column slice of a wider
    row-major buffer qualifies); anything else is made contiguous."""
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.size(1):
        return t, t.stride(0)
    t = t.contiguous()
    return t, t.size(1)


MODE_SUM, MODE_MEAN, MODE_MAX = 0, 1, 2
_MODES = {"sum": MODE_SUM, "mean": MODE_MEAN, "avg": MODE_MEAN, "max": MODE_MAX}


# ------------------------------------------------------------------------------------------- voxelize
def assemble_sweeps(raw: torch.Tensor, offsets, params, transform, remove_close, close_radius=1.0, pc_range=None, norm_col=3,
                    norm_mean=0.0, norm_std=255.0):
    """fsf_assemble_sweeps: raw f32 [n, load_dim] (key frame rows, then every sweep's) on the device + per-sweep host metadata
    -> f32 [count, load_dim + 3] = LoadPointsFromMultiSweeps + SaveNoAugPoints (+ PointsRangeFilter) + NormalizePoints."""
    require_cuda(raw)
    assert raw.dtype == torch.float32 and raw.dim() == 2
    raw = raw.contiguous()
    n, load_dim = raw.shape
    ns = len(offsets) - 1
    out = torch.empty((n, load_dim + 3), dtype=torch.float32, device=raw.device)
    count = ctypes.c_int64(0)
    h = _L()
    ws = _lib.workspace(h.fsf_assemble_sweeps_workspace_bytes(n), raw.device)
    par = (ctypes.c_double * (ns * 13))(*[float(v) for row in params for v in row])
    check(h.fsf_assemble_sweeps(ptr(raw), n, load_dim, _lib.i64_array(offsets), ns, par,
                                (ctypes.c_uint8 * ns)(*[int(bool(v)) for v in transform]),
                                (ctypes.c_uint8 * ns)(*[int(bool(v)) for v in remove_close]), float(close_radius),
                                _lib.f32_array(pc_range) if pc_range is not None else c_p(None), int(norm_col), float(norm_mean),
                                float(norm_std), ptr(out), None, ctypes.cast(ctypes.pointer(count), c_p), ptr(ws), ws.numel(),
                                stream_ptr()), "fsf_assemble_sweeps")
    return out[: int(count.value)]


def voxelize_dynamic(points: torch.Tensor, voxel_size, pc_range, grid, batch_idx: int = 0, want_zyx=True,
                     want_bzyx=False):
    """fsf_voxelize_dynamic.  points f32 [n, C>=3] -> (coors_zyx i32 [n,3] | None, coors_bzyx i64 [n,4] | None)."""
    require_cuda(points)
    assert points.dtype == torch.float32 and points.dim() == 2 and points.size(1) >= 3
    points = points.contiguous()
    n = points.size(0)
    zyx = torch.empty((n, 3), dtype=torch.int32, device=points.device) if want_zyx else None
    bzyx = torch.empty((n, 4), dtype=torch.int64, device=points.device) if want_bzyx else None
    check(_L().fsf_voxelize_dynamic(ptr(points), n, points.size(1), int(batch_idx), f32_array(voxel_size),
                                    f32_array(pc_range), i32_array(grid), ptr(zyx), ptr(bzyx), stream_ptr()),
          "fsf_voxelize_dynamic")
    return zyx, bzyx


def voxelize_divfloor(points: torch.Tensor, voxel_size, range_min, order="zyx", batch_idx: Optional[torch.Tensor] = None):
    """fsf_voxelize_divfloor: torch.div(p - min, v, rounding_mode='floor').long() keys, i64 [n, 3|4]."""
    require_cuda(points, batch_idx)
    assert points.dtype == torch.float32 and points.dim() == 2 and points.size(1) >= 3
    points = points.contiguous()
    n = points.size(0)
    k = 4 if batch_idx is not None else 3
    if batch_idx is not None:
        batch_idx = batch_idx.to(torch.int64).contiguous()
    coors = torch.empty((n, k), dtype=torch.int64, device=points.device)
    check(_L().fsf_voxelize_divfloor(ptr(points), n, points.size(1), f32_array(voxel_size), f32_array(range_min),
                                     {"xyz": 0, "zyx": 1}[order], ptr(batch_idx), ptr(coors), stream_ptr()),
          "fsf_voxelize_divfloor")
    return coors


def vfe_decorate(features, voxel_mean, inv, coors_bzyx, voxel_size, offset, with_cluster_center=True, with_voxel_center=True):
    """fsf_vfe_decorate: features f32 [n, P] -> f32 [n, P (+3) (+3)] as a view of a buffer with 16-byte rows."""
    require_cuda(features, voxel_mean, inv, coors_bzyx)
    f, fstride = _rows_view(features)
    n, p = f.shape
    width = p + (3 if with_cluster_center else 0) + (3 if with_voxel_center else 0)
    stride = (width + 3) // 4 * 4
    buf = torch.empty((n, stride), dtype=torch.float32, device=f.device)
    vm, vstride = _rows_view(voxel_mean) if with_cluster_center else (None, 3)
    inv = inv.to(torch.int64).contiguous() if with_cluster_center else None
    coors = coors_bzyx.to(torch.int64).contiguous() if with_voxel_center else None
    check(_L().fsf_vfe_decorate(c_p(f.data_ptr()) if n else c_p(None), n, int(fstride), p, ptr(vm), int(vstride), ptr(inv), ptr(coors),
                                f32_array(voxel_size), f32_array(offset), int(bool(with_cluster_center)), int(bool(with_voxel_center)),
                                ptr(buf), stride, stream_ptr()), "fsf_vfe_decorate")
    return buf[:, :width]


def vote_centers_keys(logits, offsets, points, batch_idx, g_ids, p_ids, num_classes, group_class_masks, group_voxel_sizes,
                      range_min, batch_size):
    """fsf_vote_centers_keys: for the (group, point) pairs (g_ids, p_ids i64 [n]) -> (centers f32 [n,3], keys i64 [n,4] =
    (g * batch_size + batch, vx, vy, vz), batch i64 [n]).  logits f32 [P, >= num_classes], offsets f32 [P, >= 3 * num_classes],
    points f32 [P, >= 3] (row-strided views allowed); group_class_masks: one bit mask of member classes per group."""
    require_cuda(logits, offsets, points, batch_idx, g_ids, p_ids)
    lg, lstride = _rows_view(logits)
    of, ostride = _rows_view(offsets)
    pt, pstride = _rows_view(points)
    assert lg.dtype == of.dtype == pt.dtype == torch.float32
    g_ids, p_ids = g_ids.to(torch.int64).contiguous(), p_ids.to(torch.int64).contiguous()
    if batch_idx is not None:
        batch_idx = batch_idx.to(torch.int64).contiguous()
    n, ng = p_ids.numel(), len(group_class_masks)
    dev = pt.device
    centers = torch.empty((n, 3), dtype=torch.float32, device=dev)
    keys = torch.empty((n, 4), dtype=torch.int64, device=dev)
    bout = torch.empty((n,), dtype=torch.int64, device=dev)
    masks = (ctypes.c_uint32 * ng)(*[int(m) for m in group_class_masks])
    vs = (ctypes.c_float * (3 * ng))(*[float(v) for row in group_voxel_sizes for v in row])
    check(_L().fsf_vote_centers_keys(c_p(lg.data_ptr()), int(lstride), c_p(of.data_ptr()), int(ostride), c_p(pt.data_ptr()), int(pstride),
                                     ptr(batch_idx), ptr(g_ids), ptr(p_ids), n, int(num_classes), ng, masks, vs, f32_array(range_min),
                                     int(batch_size), ptr(centers), ptr(keys), ptr(bout), stream_ptr()), "fsf_vote_centers_keys")
    return centers, keys, bout


# --------------------------------------------------------------------------------------------- unique
@dataclass
class SegmentPlan:
    """Sort-once segment plan shared by every segmented reduction over the same key."""
    inv: torch.Tensor          # i64 [n]   point -> segment
    order: torch.Tensor        # i32 [n]   point indices stably sorted by segment
    seg_offsets: torch.Tensor  # i32 [m+1] CSR offsets into `order`
    m: int
    cnt: Optional[torch.Tensor] = None  # i64 [m]

    @property
    def n(self):
        return self.inv.numel()


def unique_rows(coors: torch.Tensor, col_min: Optional[Sequence[int]] = None, col_max: Optional[Sequence[int]] = None,
                return_counts=True):
    """fsf_unique_rows: (new_coors i64 [m,k] ascending lexicographic, SegmentPlan)."""
    require_cuda(coors)
    squeeze = coors.dim() == 1
    if squeeze:
        coors = coors[:, None]
    assert coors.dim() == 2 and 1 <= coors.size(1) <= 4
    coors = coors.to(torch.int64).contiguous()
    n, k = coors.shape
    dev = coors.device
    new_coors = torch.empty((max(n, 1), k), dtype=torch.int64, device=dev)
    inv = torch.empty((n,), dtype=torch.int64, device=dev)
    cnt = torch.empty((max(n, 1),), dtype=torch.int64, device=dev) if return_counts else None
    order = torch.empty((n,), dtype=torch.int32, device=dev)
    seg_offsets = torch.empty((n + 1,), dtype=torch.int32, device=dev)
    m_dev = torch.empty((1,), dtype=torch.int64, device=dev)
    m_host = c_i64(0)
    h = _L()
    ws_bytes = h.fsf_unique_rows_workspace_bytes(n, k)
    ws = _lib.workspace(ws_bytes, dev)
    cmin = i64_array(col_min) if col_min is not None else None
    cmax = i64_array(col_max) if col_max is not None else None
    import ctypes
    check(h.fsf_unique_rows(ptr(coors), n, k, cmin, cmax, ptr(new_coors), ptr(inv), ptr(cnt), ptr(order),
                            ptr(seg_offsets), ptr(m_dev), ctypes.cast(ctypes.pointer(m_host), c_p), ptr(ws),
                            ws.numel(), stream_ptr()), "fsf_unique_rows")
    m = int(m_host.value)
    new_coors = new_coors[:m]
    if squeeze:
        new_coors = new_coors[:, 0]
    plan = SegmentPlan(inv=inv, order=order, seg_offsets=seg_offsets[: m + 1], m=m,
                       cnt=cnt[:m] if cnt is not None else None)
    return new_coors, plan


def segment_plan_from_inverse(inv: torch.Tensor, m: int, return_counts=False) -> SegmentPlan:
    require_cuda(inv)
    inv = inv.to(torch.int64).contiguous()
    n = inv.numel()
    dev = inv.device
    order = torch.empty((n,), dtype=torch.int32, device=dev)
    seg_offsets = torch.empty((m + 1,), dtype=torch.int32, device=dev)
    cnt = torch.empty((m,), dtype=torch.int64, device=dev) if return_counts else None
    h = _L()
    ws = _lib.workspace(h.fsf_segment_plan_workspace_bytes(n, m), dev)
    check(h.fsf_segment_plan_from_inverse(ptr(inv), n, m, ptr(order), ptr(seg_offsets), ptr(cnt), ptr(ws), ws.numel(),
                                          stream_ptr()), "fsf_segment_plan_from_inverse")
    return SegmentPlan(inv=inv, order=order, seg_offsets=seg_offsets, m=m, cnt=cnt)


# ------------------------------------------------------------------------------------- segment reduce
def segment_reduce(feat: torch.Tensor, plan: SegmentPlan, mode: str, return_argmax=False):
    """fsf_segment_reduce: feat f32 [n,c] -> out f32 [m,c] (+ argmax i64 [m,c] for mode='max')."""
    require_cuda(feat)
    assert feat.dtype == torch.float32 and feat.dim() == 2 and feat.size(0) == plan.n
    feat, feat_stride = _rows_view(feat)
    n, c = feat.shape
    dev = feat.device
    out = torch.empty((plan.m, c), dtype=torch.float32, device=dev)
    md = _MODES[mode]
    argmax = torch.empty((plan.m, c), dtype=torch.int64, device=dev) if (return_argmax and md == MODE_MAX) else None
    h = _L()
    ws = _lib.workspace(h.fsf_segment_reduce_workspace_bytes(n, plan.m, c), dev)
    check(h.fsf_segment_reduce(c_p(feat.data_ptr()), feat_stride, n, c, ptr(plan.order), ptr(plan.inv), ptr(plan.seg_offsets), plan.m, md,
                               ptr(out), ptr(argmax), ptr(ws), ws.numel(), stream_ptr()), "fsf_segment_reduce")
    return (out, argmax) if return_argmax else out


def segment_reduce_short(feats, plan: SegmentPlan, mode: str, return_argmax=False):
    """fsf_segment_reduce_short: the reduction of up to 8 tensors f32 [n, c_t] over ONE plan whose segments are short (voxels)
    in one launch -> list of f32 [m, c_t] (+ argmax i64 [m, c] with a single tensor and mode='max')."""
    feats = list(feats)
    require_cuda(*feats)
    assert 1 <= len(feats) <= 8 and all(f.dtype == torch.float32 and f.dim() == 2 and f.size(0) == plan.n for f in feats)
    views = [_rows_view(f) for f in feats]
    dev = feats[0].device
    outs = [torch.empty((plan.m, f.size(1)), dtype=torch.float32, device=dev) for f in feats]
    md = _MODES[mode]
    argmax = None
    if return_argmax and md == MODE_MAX:
        assert len(feats) == 1
        argmax = torch.empty((plan.m, feats[0].size(1)), dtype=torch.int64, device=dev)
    nt = len(feats)
    fp = (ctypes.c_void_p * nt)(*[v[0].data_ptr() if v[0].numel() else None for v in views])
    op = (ctypes.c_void_p * nt)(*[o.data_ptr() if o.numel() else None for o in outs])
    st = (ctypes.c_int64 * nt)(*[int(v[1]) for v in views])
    cs = (ctypes.c_int32 * nt)(*[int(f.size(1)) for f in feats])
    if plan.m > 0:
        check(_L().fsf_segment_reduce_short(fp, st, cs, nt, plan.n, ptr(plan.order), ptr(plan.seg_offsets), plan.m, md, op,
                                            ptr(argmax), stream_ptr()), "fsf_segment_reduce_short")
    return (outs, argmax) if return_argmax else outs


def segment_reduce_backward(grad_out: torch.Tensor, plan: SegmentPlan, mode: str, argmax: Optional[torch.Tensor] = None):
    require_cuda(grad_out)
    grad_out = grad_out.contiguous()
    m, c = grad_out.shape
    n = plan.n
    grad_feat = torch.empty((n, c), dtype=torch.float32, device=grad_out.device)
    check(_L().fsf_segment_reduce_backward(ptr(grad_out), n, c, ptr(plan.inv), ptr(plan.seg_offsets), m, _MODES[mode],
                                           ptr(argmax), ptr(grad_feat), stream_ptr()), "fsf_segment_reduce_backward")
    return grad_feat


def gather_rows(src: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None):
    """fsf_gather_rows[_strided]: out[i,:] = src[idx[i],:].  `src` and `out` may be column slices of wider row-major buffers."""
    require_cuda(src, idx, out)
    assert src.dtype == torch.float32 and src.dim() == 2
    if src.stride(1) != 1 or (src.size(0) > 1 and src.stride(0) < src.size(1)):
        src = src.contiguous()
    idx = idx.to(torch.int64).contiguous()
    n, (m, c) = idx.numel(), src.shape
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=src.device)
    assert out.shape == (n, c) and out.dtype == torch.float32 and out.stride(1) == 1 and out.stride(0) >= c
    check(_L().fsf_gather_rows_strided(c_p(src.data_ptr()) if src.numel() else c_p(None), src.stride(0) if m > 1 else c, m, c, ptr(idx), n,
                                       c_p(out.data_ptr()), out.stride(0), stream_ptr()), "fsf_gather_rows_strided")
    return out


def gather_rows_add(src: torch.Tensor, idx: torch.Tensor, add: torch.Tensor):
    """fsf_gather_rows_add: add[i,:] + src[idx[i],:] -> f32 [n, c] (new tensor)."""
    require_cuda(src, idx, add)
    assert src.dtype == torch.float32 and src.dim() == 2 and add.dtype == torch.float32
    src, add = src.contiguous(), add.contiguous()
    idx = idx.to(torch.int64).contiguous()
    n, (m, c) = idx.numel(), src.shape
    assert add.shape == (n, c)
    out = torch.empty((n, c), dtype=torch.float32, device=src.device)
    check(_L().fsf_gather_rows_add(ptr(src), c, m, c, ptr(idx), n, ptr(add), c, ptr(out), c, stream_ptr()), "fsf_gather_rows_add")
    return out


def channel_group_sum_add(feat: torch.Tensor, cout: int, add: Optional[torch.Tensor] = None):
    """fsf_channel_group_sum_add: feat f32 [n, cin] -> add + feat.view(n, cout, cin // cout).sum(2), f32 [n, cout]."""
    require_cuda(feat, add)
    feat = feat.contiguous()
    n, cin = feat.shape
    if add is not None:
        add = add.contiguous()
        assert add.shape == (n, cout)
    out = torch.empty((n, cout), dtype=torch.float32, device=feat.device)
    check(_L().fsf_channel_group_sum_add(ptr(feat), n, cin, cout, ptr(add), ptr(out), stream_ptr()), "fsf_channel_group_sum_add")
    return out


def channel_pair_sum_add2_supported(a: torch.Tensor, b: torch.Tensor) -> bool:
    return (a.is_cuda and a.dtype == b.dtype == torch.float32 and a.dim() == b.dim() == 2 and a.size(0) == b.size(0)
            and a.size(1) % 8 == 0 and b.size(1) % 8 == 0 and a.is_contiguous() and b.is_contiguous())


def channel_pair_sum_add2(a: torch.Tensor, b: torch.Tensor, add: Optional[torch.Tensor] = None):
    """fsf_channel_pair_sum_add2: add + cat([a, b], 1).view(n, (ca + cb) / 2, 2).sum(2) without the concatenation."""
    require_cuda(a, b, add)
    n, ca, cb = a.size(0), a.size(1), b.size(1)
    cout = (ca + cb) // 2
    if add is not None:
        add = add.contiguous()
        assert add.shape == (n, cout)
    out = torch.empty((n, cout), dtype=torch.float32, device=a.device)
    check(_L().fsf_channel_pair_sum_add2(ptr(a), ca, ptr(b), cb, n, ptr(add), ptr(out), stream_ptr()), "fsf_channel_pair_sum_add2")
    return out


def channel_pair_sum_add2_planes(a: torch.Tensor, b: torch.Tensor, add: Optional[torch.Tensor] = None) -> "Planes":
    """fsf_channel_pair_sum_add2_planes: to_planes(channel_pair_sum_add2(a, b, add)) in one launch, the fp32 rows never written."""
    require_cuda(a, b, add)
    n, ca, cb = a.size(0), a.size(1), b.size(1)
    cout = (ca + cb) // 2
    assert ca % 16 == 0 and cb % 16 == 0 and a.is_contiguous() and b.is_contiguous()
    if add is not None:
        add = add.contiguous()
        assert add.shape == (n, cout)
    out = planes_empty(n, cout, a.device)
    check(_L().fsf_channel_pair_sum_add2_planes(ptr(a), ca, ptr(b), cb, n, ptr(add), ptr(out.data), ptr(out.scales), stream_ptr()),
          "fsf_channel_pair_sum_add2_planes")
    return out


def voxel2point(points, coors_bzyx, voxel_feats, inv, voxel_size, range_min, padding=-1.0):
    """fsf_voxel2point: fused gather + local-xyz decoration + padding mask (Voxel2PointScatterNeck)."""
    require_cuda(points, coors_bzyx, voxel_feats, inv)
    points = points.contiguous()
    coors_bzyx = coors_bzyx.to(torch.int64).contiguous()
    voxel_feats = voxel_feats.contiguous()
    inv = inv.to(torch.int64).contiguous()
    n = points.size(0)
    m, c = voxel_feats.shape
    # rows padded to a multiple of 4 floats (16-byte aligned rows): the [n, c + 3] view is then a legal K22 operand as it is
    stride = (c + 3 + 3) // 4 * 4
    buf = torch.empty((n, stride), dtype=torch.float32, device=points.device)
    valid = torch.empty((n,), dtype=torch.uint8, device=points.device)
    check(_L().fsf_voxel2point_strided(ptr(points), points.size(1), ptr(coors_bzyx), ptr(voxel_feats), m, c, ptr(inv), n,
                                       f32_array(voxel_size), f32_array(range_min), float(padding), ptr(buf), stride, ptr(valid),
                                       stream_ptr()), "fsf_voxel2point_strided")
    return buf[:, :c + 3], valid.bool()


# ---------------------------------------------------------------------------------------- projection
def project_gather_mask(xyz: torch.Tensor, lidar2img: torch.Tensor, mask: torch.Tensor, return_pts_2d=False):
    """fsf_project_gather_mask for ONE sample: xyz f32 [n,>=3], lidar2img f32 [ncam,4,4],
    mask u8|i32 [ncam,ncls,H,W] -> obj_id i64 [n,ncam,ncls] (+ pts_2d f32 [ncam,n,2])."""
    require_cuda(xyz, lidar2img, mask)
    assert xyz.dtype == torch.float32 and lidar2img.dtype == torch.float32
    assert mask.dtype in (torch.uint8, torch.int32) and mask.dim() == 4
    xyz = xyz.contiguous()
    lidar2img = lidar2img.contiguous()
    mask = mask.contiguous()
    n = xyz.size(0)
    ncam, ncls, H, W = mask.shape
    assert lidar2img.shape == (ncam, 4, 4)
    obj_id = torch.empty((n, ncam, ncls), dtype=torch.int64, device=xyz.device)
    pts_2d = torch.empty((ncam, n, 2), dtype=torch.float32, device=xyz.device) if return_pts_2d else None
    check(_L().fsf_project_gather_mask(ptr(xyz), n, xyz.size(1), ptr(lidar2img), ncam, ptr(mask), mask.element_size(),
                                       ncls, H, W, ptr(obj_id), ptr(pts_2d), stream_ptr()), "fsf_project_gather_mask")
    return (obj_id, pts_2d) if return_pts_2d else obj_id


PROJECT_SCORE_MAX_CLS = 16


def project_score(xyz: torch.Tensor, lidar2img: torch.Tensor, mask: torch.Tensor, mask_anno: torch.Tensor, score_col=4,
                  return_ids=False, return_fg=True, return_overlap=False):
    """fsf_project_score for ONE sample: xyz f32 [n,>=3], mask u8|i32 [ncam,ncls,H,W], mask_anno f32 [A,D] ->
    score f32 [n,ncls] (+ ids i64 [n,ncls] of the selected camera) (+ fg bool [n]: inside any mask)
    (+ with return_overlap: (fg u8 [n], count u8 [n], max_id i32 [n]) — what overlap_plan / overlap_rows take)."""
    require_cuda(xyz, lidar2img, mask, mask_anno)
    assert xyz.dtype == torch.float32 and lidar2img.dtype == torch.float32 and mask.dtype in (torch.uint8, torch.int32)
    xyz, lidar2img, mask = xyz.contiguous(), lidar2img.contiguous(), mask.contiguous()
    mask_anno = mask_anno.to(torch.float32).contiguous()
    n = xyz.size(0)
    ncam, ncls, H, W = mask.shape
    score = torch.empty((n, ncls), dtype=torch.float32, device=xyz.device)
    ids = torch.empty((n, ncls), dtype=torch.int64, device=xyz.device) if return_ids else None
    fg = torch.empty((n,), dtype=torch.uint8, device=xyz.device) if return_fg or return_overlap else None
    count = torch.empty((n,), dtype=torch.uint8, device=xyz.device) if return_overlap else None
    max_id = torch.empty((n,), dtype=torch.int32, device=xyz.device) if return_overlap else None
    check(_L().fsf_project_score(ptr(xyz), n, xyz.size(1), ptr(lidar2img), ncam, ptr(mask), mask.element_size(), ncls, H, W,
                                 ptr(mask_anno), mask_anno.size(0), mask_anno.size(1), int(score_col), ptr(score), ptr(ids),
                                 ptr(fg), ptr(count), ptr(max_id), stream_ptr()), "fsf_project_score")
    out = (score,)
    if return_ids:
        out += (ids,)
    if return_fg:
        out += (fg.bool(),)
    if return_overlap:
        out += ((fg, count, max_id),)
    return out if len(out) > 1 else score


def _concat_mul_args(points, feats, extra, xyz_normalizer, extra_div):
    assert points.dtype == torch.float32 and points.dim() == 2 and points.stride(1) == 1 and points.size(1) >= 3
    n = points.size(0)
    for t in (feats, extra):
        assert t is None or (t.dtype == torch.float32 and t.dim() == 2 and t.size(0) == n and t.stride(1) == 1)
    def rows(t):
        return (c_p(t.data_ptr()) if t is not None and t.numel() else c_p(None), (t.stride(0) if n > 1 else t.size(1)) if t is not None else 0,
                t.size(1) if t is not None else 0)
    return (c_p(points.data_ptr()) if n else c_p(None), points.stride(0) if n > 1 else points.size(1), points.size(1),
            f32_array(xyz_normalizer), *rows(feats), *rows(extra), float(extra_div))


def concat_mul(points, feats, extra, h, xyz_normalizer, extra_div=1.0):
    """fsf_concat_mul (K28): cat([points[:, :3] / normalizer, points[:, 3:], feats, extra / extra_div], 1) * h -> f32 [n, c]."""
    require_cuda(points, feats, extra, h)
    n = points.size(0)
    c = points.size(1) + (feats.size(1) if feats is not None else 0) + (extra.size(1) if extra is not None else 0)
    assert h.shape == (n, c) and h.dtype == torch.float32 and h.is_contiguous()
    out = torch.empty((n, c), dtype=torch.float32, device=points.device)
    check(_L().fsf_concat_mul(*_concat_mul_args(points, feats, extra, xyz_normalizer, extra_div), ptr(h), n, ptr(out), stream_ptr()),
          "fsf_concat_mul")
    return out


def concat_mul_backward(points, feats, extra, h, grad_out, xyz_normalizer, extra_div=1.0, want_feats=True, want_extra=False):
    """fsf_concat_mul_backward -> (grad_h [n, c], grad_feats [n, cf] | None, grad_extra [n, ce] | None)."""
    require_cuda(points, feats, extra, h, grad_out)
    n, c = h.shape
    assert grad_out.shape == (n, c) and grad_out.dtype == torch.float32 and grad_out.is_contiguous() and h.is_contiguous()
    g_h = torch.empty((n, c), dtype=torch.float32, device=h.device)
    g_f = torch.empty((n, feats.size(1)), dtype=torch.float32, device=h.device) if (want_feats and feats is not None) else None
    g_e = torch.empty((n, extra.size(1)), dtype=torch.float32, device=h.device) if (want_extra and extra is not None) else None
    check(_L().fsf_concat_mul_backward(*_concat_mul_args(points, feats, extra, xyz_normalizer, extra_div), ptr(h), ptr(grad_out), n,
                                       ptr(g_h), ptr(g_f), ptr(g_e), stream_ptr()), "fsf_concat_mul_backward")
    return g_h, g_f, g_e


def group_pairs(score: torch.Tensor, thresh: torch.Tensor, keep_one=True, group_cols=None):
    """fsf_group_pairs: score f32 [n, ng], thresh f32 [ng] -> (g_ids i64 [P], p_ids i64 [P]) = ((score > thresh) with point 0 kept
    for a group nobody passes).t().nonzero() columns; one host wait.  With `group_cols` (per group the list of its one or two class
    columns) `score` is the class score matrix [n, classes] and the group scores are formed inside."""
    require_cuda(score, thresh)
    assert score.dtype == torch.float32 and thresh.dtype == torch.float32 and score.dim() == 2 and score.stride(1) == 1
    n, cols = score.shape
    ng = thresh.numel()
    masks = None
    if group_cols is not None:
        assert len(group_cols) == ng and all(1 <= len(c) <= 2 and max(c) < cols for c in group_cols)
        masks = (ctypes.c_uint32 * ng)(*[sum(1 << int(c) for c in cs) for cs in group_cols])
    else:
        assert cols == ng
    assert thresh.is_contiguous()
    cap = max(n * ng, 1)
    buf = torch.empty((2, cap), dtype=torch.int64, device=score.device)
    h = _L()
    ws = _lib.workspace(h.fsf_group_pairs_workspace_bytes(n, ng), score.device)
    count = c_i64(0)
    check(h.fsf_group_pairs(c_p(score.data_ptr()) if n else c_p(None), n, ng, score.stride(0) if n > 1 else cols, ptr(thresh),
                            int(bool(keep_one)), ctypes.cast(masks, c_p) if masks is not None else c_p(None), cols, ptr(buf[0]),
                            ptr(buf[1]), cap, ctypes.cast(ctypes.pointer(count), c_p), ptr(ws), ws.numel(), stream_ptr()),
          "fsf_group_pairs")
    k = int(count.value)
    if 2 * k < cap:  # the upper-bound buffer would stay alive behind two short views (~56 MB at 5e5 points x 7 groups): copy them out
        return buf[0, :k].clone(), buf[1, :k].clone()
    return buf[0, :k], buf[1, :k]


def overlap_plan(fg: torch.Tensor, count: torch.Tensor, max_cells: int):
    """fsf_overlap_plan: fg u8 [n], count u8 [n] -> (F, M, T, workspace): foreground points, points inside >= 2 masks, appended rows;
    the workspace goes to overlap_rows unchanged.  One host wait."""
    require_cuda(fg, count)
    assert fg.dtype == torch.uint8 and count.dtype == torch.uint8 and fg.is_contiguous() and count.is_contiguous()
    n = fg.size(0)
    # (its own buffer, not the per-stream scratch: the index lists must survive until overlap_rows)
    ws = torch.empty((max(int(_L().fsf_overlap_plan_workspace_bytes(n)), 256),), dtype=torch.uint8, device=fg.device)
    counts = (ctypes.c_int64 * 3)()
    check(_L().fsf_overlap_plan(ptr(fg), ptr(count), n, int(max_cells), ctypes.cast(counts, c_p), ptr(ws), ws.numel(), stream_ptr()), "fsf_overlap_plan")
    return int(counts[0]), int(counts[1]), int(counts[2]), ws


def overlap_rows(xyz: torch.Tensor, lidar2img: torch.Tensor, mask: torch.Tensor, max_id: torch.Tensor, batch_idx, ws, num_fg, num_multi,
                 num_extra):
    """fsf_overlap_rows -> (src_pt i64 [F + T], sir_coors i64 [F + T, 3]) in the order extract_fg_pts + double_overlap_pts produce."""
    require_cuda(xyz, lidar2img, mask, max_id)
    assert xyz.dtype == torch.float32 and lidar2img.dtype == torch.float32 and mask.dtype in (torch.uint8, torch.int32)
    assert max_id.dtype == torch.int32 and (batch_idx is None or (batch_idx.dtype == torch.int64 and batch_idx.is_contiguous()))
    xyz, lidar2img, mask = xyz.contiguous(), lidar2img.contiguous(), mask.contiguous()
    n = xyz.size(0)
    ncam, ncls, H, W = mask.shape
    rows = int(num_fg) + int(num_extra)
    src_pt = torch.empty((rows,), dtype=torch.int64, device=xyz.device)
    sir_coors = torch.empty((rows, 3), dtype=torch.int64, device=xyz.device)
    check(_L().fsf_overlap_rows(ptr(xyz), n, xyz.size(1), ptr(lidar2img), ncam, ptr(mask), mask.element_size(), ncls, H, W, ptr(max_id),
                                ptr(batch_idx), ptr(ws), ws.numel(), int(num_fg), int(num_multi), int(num_extra), ptr(src_pt),
                                ptr(sir_coors), stream_ptr()), "fsf_overlap_rows")
    return src_pt, sir_coors


def project_gather_bilinear(xyz: torch.Tensor, lidar2img: torch.Tensor, feat: torch.Tensor, img_hw, channels_last=False,
                            reduce_cams=False, return_count=False):
    """fsf_project_gather_bilinear for ONE sample: xyz f32 [n,>=3], lidar2img f32 [ncam,4,4], feat f32 [ncam,C,Hf,Wf]
    (or [ncam,Hf,Wf,C] with channels_last) -> f32 [n,ncam,C] (or [n,C] summed over the cameras that see the point);
    grid_sample(bilinear, align_corners=False, zeros) at the projection of FSF.prj_points_2d."""
    require_cuda(xyz, lidar2img, feat)
    assert xyz.dtype == torch.float32 and lidar2img.dtype == torch.float32 and feat.dtype == torch.float32 and feat.dim() == 4
    xyz, lidar2img, feat = xyz.contiguous(), lidar2img.contiguous(), feat.contiguous()
    n = xyz.size(0)
    if channels_last:
        ncam, hf, wf, c = feat.shape
    else:
        ncam, c, hf, wf = feat.shape
    assert lidar2img.shape == (ncam, 4, 4)
    out = torch.empty((n, c) if reduce_cams else (n, ncam, c), dtype=torch.float32, device=xyz.device)
    count = torch.empty((n,), dtype=torch.uint8, device=xyz.device) if return_count else None
    check(_L().fsf_project_gather_bilinear(ptr(xyz), n, xyz.size(1), ptr(lidar2img), ncam, ptr(feat), c, hf, wf,
                                           int(bool(channels_last)), int(img_hw[0]), int(img_hw[1]), int(bool(reduce_cams)),
                                           ptr(out), ptr(count), stream_ptr()), "fsf_project_gather_bilinear")
    return (out, count) if return_count else out


def cam_select_score(obj_id: torch.Tensor, mask_anno: torch.Tensor, score_col=4, return_ids=False):
    """fsf_cam_select_score for ONE sample: obj_id i64 [n,ncam,ncls], mask_anno f32 [A,D] -> score f32 [n,ncls]."""
    require_cuda(obj_id, mask_anno)
    obj_id = obj_id.contiguous()
    mask_anno = mask_anno.to(torch.float32).contiguous()
    n, ncam, ncls = obj_id.shape
    score = torch.empty((n, ncls), dtype=torch.float32, device=obj_id.device)
    ids = torch.empty((n, ncls), dtype=torch.int64, device=obj_id.device) if return_ids else None
    check(_L().fsf_cam_select_score(ptr(obj_id), n, ncam, ncls, ptr(mask_anno), mask_anno.size(0), mask_anno.size(1),
                                    int(score_col), ptr(ids), ptr(score), stream_ptr()), "fsf_cam_select_score")
    return (score, ids) if return_ids else score


# ------------------------------------------------------------------------------------------ rulebooks
def rulebook_subm(indices: torch.Tensor, batch_size: int, spatial_shape, ksize=(3, 3, 3), dilation=(1, 1, 1)):
    """fsf_rulebook_subm: indices i32 [m,4] (b,z,y,x) -> nbr i32 [m, kvol]."""
    require_cuda(indices)
    assert indices.dtype == torch.int32 and indices.dim() == 2 and indices.size(1) == 4
    indices = indices.contiguous()
    m = indices.size(0)
    kvol = int(ksize[0] * ksize[1] * ksize[2])
    nbr = torch.empty((m, kvol), dtype=torch.int32, device=indices.device)
    h = _L()
    ws = _lib.workspace(h.fsf_rulebook_workspace_bytes(m, kvol), indices.device)
    check(h.fsf_rulebook_subm(ptr(indices), m, int(batch_size), i32_array(spatial_shape), i32_array(ksize),
                              i32_array(dilation), ptr(nbr), ptr(ws), ws.numel(), stream_ptr()), "fsf_rulebook_subm")
    return nbr


def conv_out_shape(spatial_shape, ksize, stride, padding, dilation):
    return [(spatial_shape[j] + 2 * padding[j] - dilation[j] * (ksize[j] - 1) - 1) // stride[j] + 1 for j in range(3)]


def rulebook_strided(indices: torch.Tensor, batch_size: int, spatial_shape, ksize, stride, padding, dilation=(1, 1, 1),
                     want_inverse=True):
    """fsf_rulebook_strided -> (out_indices i32 [m_out,4], nbr i32 [m_out,kvol], nbr_inv i32 [m,kvol] | None, out_shape)."""
    import ctypes
    require_cuda(indices)
    assert indices.dtype == torch.int32 and indices.dim() == 2 and indices.size(1) == 4
    indices = indices.contiguous()
    m = indices.size(0)
    dev = indices.device
    kvol = int(ksize[0] * ksize[1] * ksize[2])
    out_shape = conv_out_shape(spatial_shape, ksize, stride, padding, dilation)
    cells = int(batch_size) * out_shape[0] * out_shape[1] * out_shape[2]
    cap = max(1, min(m * kvol, cells))
    out_indices = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    nbr = torch.empty((cap, kvol), dtype=torch.int32, device=dev)
    nbr_inv = torch.empty((m, kvol), dtype=torch.int32, device=dev) if want_inverse else None
    m_out_host = c_i64(0)
    h = _L()
    ws = _lib.workspace(h.fsf_rulebook_workspace_bytes(m, kvol), dev)
    check(h.fsf_rulebook_strided(ptr(indices), m, int(batch_size), i32_array(spatial_shape), i32_array(ksize),
                                 i32_array(stride), i32_array(padding), i32_array(dilation), ptr(out_indices), cap,
                                 ptr(nbr), ptr(nbr_inv), None, ctypes.cast(ctypes.pointer(m_out_host), c_p), ptr(ws),
                                 ws.numel(), stream_ptr()), "fsf_rulebook_strided")
    m_out = int(m_out_host.value)
    return out_indices[:m_out], nbr[:m_out], nbr_inv, out_shape


def order_by_neighbor_mask(indices: torch.Tensor, batch_size: int, spatial_shape):
    """fsf_order_by_neighbor_mask: coordinates i32 [m, 4] (b, z, y, x) of a level -> (perm i32 [m], inv_perm i32 [m]): position i of the
    new order holds row perm[i]: descending 16-bit neighbourhood key (in-plane 3x3 mask | counts below / above), then coordinate parity, stable."""
    require_cuda(indices)
    assert indices.dtype == torch.int32 and indices.dim() == 2 and indices.size(1) == 4 and indices.is_contiguous()
    m = indices.size(0)
    perm = torch.empty((m,), dtype=torch.int32, device=indices.device)
    inv = torch.empty((m,), dtype=torch.int32, device=indices.device)
    h = _L()
    ws = _lib.workspace(h.fsf_order_by_neighbor_mask_workspace_bytes(m), indices.device)
    check(h.fsf_order_by_neighbor_mask(ptr(indices), m, int(batch_size), i32_array(spatial_shape), ptr(perm), ptr(inv), ptr(ws),
                                       ws.numel(), stream_ptr()), "fsf_order_by_neighbor_mask")
    return perm, inv


def remap_indices(table: torch.Tensor, index_map: torch.Tensor):
    """fsf_remap_indices: out = table >= 0 ? index_map[table] : -1 (i32, any shape)."""
    require_cuda(table, index_map)
    assert table.dtype == torch.int32 and index_map.dtype == torch.int32 and table.is_contiguous() and index_map.is_contiguous()
    out = torch.empty_like(table)
    check(_L().fsf_remap_indices(ptr(table), table.numel(), ptr(index_map), ptr(out), stream_ptr()), "fsf_remap_indices")
    return out


def rulebook_to_pairs(nbr: torch.Tensor):
    """fsf_rulebook_to_pairs: nbr i32 [m_out,kvol] -> (indice_pairs i32 [kvol,2,m_out], indice_num i32 [kvol])."""
    require_cuda(nbr)
    nbr = nbr.contiguous()
    m_out, kvol = nbr.shape
    cap = max(m_out, 1)
    pairs = torch.full((kvol, 2, cap), -1, dtype=torch.int32, device=nbr.device)
    num = torch.empty((kvol,), dtype=torch.int32, device=nbr.device)
    h = _L()
    ws = _lib.workspace(h.fsf_rulebook_to_pairs_workspace_bytes(m_out, kvol), nbr.device)
    check(h.fsf_rulebook_to_pairs(ptr(nbr), m_out, kvol, ptr(pairs), cap, ptr(num), ptr(ws), ws.numel(), stream_ptr()),
          "fsf_rulebook_to_pairs")
    return pairs, num


# --------------------------------------------------------------------------------------- sparse conv
def spconv_transpose_weight(weight: torch.Tensor):
    """weight f32 [kvol, cin, cout] (spconv v1 layout flattened) -> [kvol, cout, cin]."""
    require_cuda(weight)
    weight = weight.contiguous()
    kvol, cin, cout = weight.shape
    wt = torch.empty((kvol, cout, cin), dtype=torch.float32, device=weight.device)
    check(_L().fsf_spconv_transpose_weight(ptr(weight), kvol, cin, cout, ptr(wt), stream_ptr()),
          "fsf_spconv_transpose_weight")
    return wt


def spconv_forward(feat: torch.Tensor, weight_t: torch.Tensor, nbr: torch.Tensor, scale=None, shift=None, residual=None,
                   relu=False):
    """fsf_spconv_forward: feat f32 [m_in,cin], weight_t f32 [kvol,cout,cin], nbr i32 [m_out,kvol] -> f32 [m_out,cout]."""
    require_cuda(feat, weight_t, nbr)
    feat = feat.contiguous()
    weight_t = weight_t.contiguous()
    nbr = nbr.contiguous()
    m_in, cin = feat.shape
    kvol, cout, cin_w = weight_t.shape
    assert cin_w == cin and nbr.size(1) == kvol and nbr.dtype == torch.int32
    m_out = nbr.size(0)
    out = torch.empty((m_out, cout), dtype=torch.float32, device=feat.device)
    if scale is not None:
        scale = scale.contiguous()
    if shift is not None:
        shift = shift.contiguous()
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == out.shape
    h = _L()
    ws = _lib.workspace(h.fsf_spconv_workspace_bytes(m_out, cin, cout, kvol), feat.device)
    check(h.fsf_spconv_forward(ptr(feat), m_in, cin, ptr(weight_t), kvol, cout, ptr(nbr), m_out, ptr(scale), ptr(shift),
                               ptr(residual), int(bool(relu)), ptr(out), ptr(ws), ws.numel(), stream_ptr()),
          "fsf_spconv_forward")
    return out


def spconv_prepare_weight_split(weight: torch.Tensor):
    """fsf_spconv_prepare_weight_split: spconv v1 weight f32 [kvol, cin, cout] -> opaque split-bf16 fragment planes."""
    require_cuda(weight)
    weight = weight.detach().contiguous()
    kvol, cin, cout = weight.shape
    h = _L()
    planes = torch.empty(h.fsf_spconv_split_weight_bytes(kvol, cin, cout), dtype=torch.uint8, device=weight.device)
    check(h.fsf_spconv_prepare_weight_split(ptr(weight), kvol, cin, cout, ptr(planes), stream_ptr()),
          "fsf_spconv_prepare_weight_split")
    return planes


def spconv_forward_split(feat: torch.Tensor, planes: torch.Tensor, kvol: int, cout: int, nbr: torch.Tensor, scale=None,
                         shift=None, residual=None, relu=False):
    """fsf_spconv_forward_split (K9b): feat f32 [m_in,cin], planes from spconv_prepare_weight_split, nbr i32 [m_out,kvol]."""
    require_cuda(feat, planes, nbr)
    feat = feat.contiguous()
    nbr = nbr.contiguous()
    m_in, cin = feat.shape
    assert nbr.size(1) == kvol and nbr.dtype == torch.int32
    m_out = nbr.size(0)
    out = torch.empty((m_out, cout), dtype=torch.float32, device=feat.device)
    scale = scale.contiguous() if scale is not None else None
    shift = shift.contiguous() if shift is not None else None
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == out.shape
    h = _L()
    ws = _lib.workspace(h.fsf_spconv_split_workspace_bytes(m_out, cin, cout, kvol), feat.device)
    check(h.fsf_spconv_forward_split(ptr(feat), m_in, cin, ptr(planes), kvol, cout, ptr(nbr), m_out, ptr(scale), ptr(shift),
                                     ptr(residual), int(bool(relu)), ptr(out), ptr(ws), ws.numel(), stream_ptr()),
          "fsf_spconv_forward_split")
    return out


def spconv_prepare_weight_split_f16(weight: torch.Tensor):
    """fsf_spconv_prepare_weight_split_f16: spconv v1 weight f32 [kvol, cin, cout] -> header + f16 hi | lo fragment planes (K9b-XP)."""
    require_cuda(weight)
    weight = weight.detach().contiguous()
    kvol, cin, cout = weight.shape
    h = _L()
    planes = torch.empty(h.fsf_spconv_split_weight_f16_bytes(kvol, cin, cout), dtype=torch.uint8, device=weight.device)
    check(h.fsf_spconv_prepare_weight_split_f16(ptr(weight), kvol, cin, cout, ptr(planes), stream_ptr()),
          "fsf_spconv_prepare_weight_split_f16")
    return planes


def spconv_split_planes_supported(cin: int, cout: int) -> bool:
    return cin % 32 == 0 and cout % 4 == 0 and cout > 64


def spconv_forward_split_planes(xp, planes: torch.Tensor, kvol: int, cout: int, nbr: torch.Tensor, scale=None, shift=None,
                                residual=None, relu=False):
    """fsf_spconv_forward_split_planes (K9b-XP): xp = RowPlanes of the input rows (rows_to_planes), planes from
    spconv_prepare_weight_split_f16, nbr i32 [m_out, kvol] -> out f32 [m_out, cout]."""
    require_cuda(xp.data, planes, nbr, scale, shift, residual)
    nbr = nbr.contiguous()
    assert nbr.size(1) == kvol and nbr.dtype == torch.int32
    m_in, cin, m_out = xp.n, xp.c, nbr.size(0)
    out = torch.empty((m_out, cout), dtype=torch.float32, device=nbr.device)
    scale = scale.contiguous() if scale is not None else None
    shift = shift.contiguous() if shift is not None else None
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == (m_out, cout)
    h = _L()
    ws = _lib.workspace(h.fsf_spconv_split_workspace_bytes(m_out, cin, cout, kvol), nbr.device)
    check(h.fsf_spconv_forward_split_planes(ptr(xp.data), ptr(xp.inv_scales), m_in, cin, ptr(planes), kvol, cout, ptr(nbr), m_out,
                                            ptr(scale), ptr(shift), ptr(residual), int(bool(relu)), ptr(out), ptr(ws), ws.numel(),
                                            stream_ptr()), "fsf_spconv_forward_split_planes")
    return out


class Planes:
    """A feature tensor in K9c's plane form (fsf_to_planes): `data` u8 [(m + 1) * c * 4] = [m + 1][c / 8][2][8] f16 hi / lo of
    the row-scaled values (row m = zeros), `scales` f32 [m + 1, ceil(c / 128)] = the inverse row scales."""

    __slots__ = ("data", "scales", "m", "c")

    def __init__(self, data, scales, m, c):
        self.data, self.scales, self.m, self.c = data, scales, int(m), int(c)


def planes_empty(m: int, c: int, device):
    h = _L()
    data = torch.empty(max(h.fsf_planes_bytes(m, c), 16), dtype=torch.uint8, device=device)
    scales = torch.empty((m + 1, (c + 127) // 128), dtype=torch.float32, device=device)
    return Planes(data, scales, m, c)


def to_planes(feat: torch.Tensor, row_index: Optional[torch.Tensor] = None) -> Planes:
    """fsf_to_planes: f32 [m, c] (rows may be strided, c % 8 == 0) -> Planes; with `row_index` (i64 [m']) fsf_to_planes_rows: the planes
    of feat[row_index] without the gathered rows."""
    require_cuda(feat, row_index)
    assert feat.dtype == torch.float32 and feat.dim() == 2 and feat.size(1) % 8 == 0
    feat, stride = _rows_view(feat)
    if stride % 4 or feat.data_ptr() % 16:
        feat, stride = feat.contiguous(), feat.size(1)
    if row_index is not None:
        assert row_index.dtype == torch.int64 and row_index.dim() == 1 and row_index.is_contiguous()
        m, c = row_index.numel(), feat.size(1)
        out = planes_empty(m, c, feat.device)
        check(_L().fsf_to_planes_rows(c_p(feat.data_ptr()) if m else c_p(None), m, c, stride, ptr(row_index), ptr(out.data), ptr(out.scales),
                                      stream_ptr()), "fsf_to_planes_rows")
        return out
    m, c = feat.shape
    out = planes_empty(m, c, feat.device)
    check(_L().fsf_to_planes(c_p(feat.data_ptr()) if m else c_p(None), m, c, stride, ptr(out.data), ptr(out.scales), stream_ptr()),
          "fsf_to_planes")
    return out


def spconv_prepare_weight_planes(weight: torch.Tensor):
    """fsf_spconv_prepare_weight_planes: spconv v1 weight f32 [kvol, cin, cout] -> opaque f16 fragment planes (K9c)."""
    require_cuda(weight)
    weight = weight.detach().contiguous()
    kvol, cin, cout = weight.shape
    h = _L()
    planes = torch.empty(h.fsf_spconv_planes_weight_bytes(kvol, cin, cout), dtype=torch.uint8, device=weight.device)
    check(h.fsf_spconv_prepare_weight_planes(ptr(weight), kvol, cin, cout, ptr(planes), stream_ptr()),
          "fsf_spconv_prepare_weight_planes")
    return planes


def spconv_planes_supported(cins, cout: int, kvol: int) -> bool:
    """Shapes K9c takes: one or two sources of 32..128 channels (multiples of 32), cout 64 or a multiple of 128, kvol <= 27."""
    return (1 <= len(cins) <= 2 and all(32 <= c <= 128 and c % 32 == 0 for c in cins) and (cout == 64 or cout % 128 == 0)
            and kvol <= 27)


def spconv_forward_planes(sources, wplanes: torch.Tensor, kvol: int, cout: int, nbr: torch.Tensor, scale=None, shift=None,
                          residual=None, relu=False, want_out=True, want_planes=False):
    """fsf_spconv_forward_planes (K9c): `sources` = [Planes] or [Planes, Planes] (channel concatenation), wplanes from
    spconv_prepare_weight_planes, nbr i32 [m_out, kvol] -> (out f32 [m_out, cout] or None, Planes of the output or None)."""
    require_cuda(wplanes, nbr, scale, shift, residual)
    nbr = nbr.contiguous()
    assert nbr.size(1) == kvol and nbr.dtype == torch.int32 and 1 <= len(sources) <= 2 and (want_out or want_planes)
    a = sources[0]
    b = sources[1] if len(sources) > 1 else None
    m_in, m_out = a.m, nbr.size(0)
    assert b is None or b.m == m_in
    dev = nbr.device
    out = torch.empty((m_out, cout), dtype=torch.float32, device=dev) if want_out else None
    op = planes_empty(m_out, cout, dev) if want_planes else None
    scale = scale.contiguous() if scale is not None else None
    shift = shift.contiguous() if shift is not None else None
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == (m_out, cout)
    check(_L().fsf_spconv_forward_planes(ptr(a.data), ptr(a.scales), a.c, ptr(b.data) if b else c_p(None),
                                         ptr(b.scales) if b else c_p(None), b.c if b else 0, m_in, ptr(wplanes), kvol, cout,
                                         ptr(nbr), m_out, ptr(scale), ptr(shift), ptr(residual), int(bool(relu)), ptr(out),
                                         ptr(op.data) if op else c_p(None), ptr(op.scales) if op else c_p(None), stream_ptr()),
          "fsf_spconv_forward_planes")
    return out, op


def linear_backward_weight(x: torch.Tensor, grad_out: torch.Tensor):
    """X^T dY through fsf_spconv_backward_weight's identity pairing: x f32 [n,cin], grad_out f32 [n,cout] -> f32 [cin,cout]."""
    require_cuda(x, grad_out)
    x, grad_out = x.contiguous(), grad_out.contiguous()
    n, cin = x.shape
    cout = grad_out.size(1)
    gw = torch.empty((1, cin, cout), dtype=torch.float32, device=x.device)
    h = _L()
    ws = _lib.workspace(h.fsf_spconv_backward_weight_workspace_bytes(n, cin, cout, 1), x.device)
    check(h.fsf_spconv_backward_weight(ptr(x), n, cin, ptr(grad_out), n, cout, None, None, n, 1, ptr(gw), ptr(ws), ws.numel(),
                                       stream_ptr()), "fsf_spconv_backward_weight")
    return gw[0]


def spconv_backward_weight(feat: torch.Tensor, grad_out: torch.Tensor, pairs: torch.Tensor, num: torch.Tensor):
    """fsf_spconv_backward_weight: feat f32 [m_in,cin], grad_out f32 [m_out,cout], (pairs, num) from
    rulebook_to_pairs -> grad_weight f32 [kvol,cin,cout]."""
    require_cuda(feat, grad_out, pairs, num)
    feat = feat.contiguous()
    grad_out = grad_out.contiguous()
    assert pairs.is_contiguous() and pairs.dtype == torch.int32 and num.dtype == torch.int32
    kvol, _, cap = pairs.shape
    m_in, cin = feat.shape
    m_out, cout = grad_out.shape
    if m_in == 0 or m_out == 0:
        return torch.zeros((kvol, cin, cout), dtype=torch.float32, device=feat.device)
    gw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=feat.device)
    h = _L()
    ws = _lib.workspace(h.fsf_spconv_backward_weight_workspace_bytes(cap, cin, cout, kvol), feat.device)
    check(h.fsf_spconv_backward_weight(ptr(feat), m_in, cin, ptr(grad_out), m_out, cout, ptr(pairs), ptr(num), cap, kvol,
                                       ptr(gw), ptr(ws), ws.numel(), stream_ptr()), "fsf_spconv_backward_weight")
    return gw


# ------------------------------------------------------------------------------------- in-group rank
def ingroup_rank(group_inds: torch.Tensor):
    """fsf_ingroup_rank: stable rank of each element inside its group (TorchEx ingroup_indices contract)."""
    require_cuda(group_inds)
    g = group_inds.to(torch.int64).contiguous()
    n = g.numel()
    out = torch.empty((n,), dtype=torch.int64, device=g.device)
    h = _L()
    ws = _lib.workspace(h.fsf_ingroup_rank_workspace_bytes(n), g.device)
    check(h.fsf_ingroup_rank(ptr(g), n, ptr(out), ptr(ws), ws.numel(), stream_ptr()), "fsf_ingroup_rank")
    return out


def row_topk_desc(x: torch.Tensor, k: int):
    """fsf_row_topk_desc: x i64 [n, w<=128] -> the k largest values of each row, descending, i64 [n, k]."""
    require_cuda(x)
    assert x.dtype == torch.int64 and x.dim() == 2
    x = x.contiguous()
    n, w = x.shape
    out = torch.empty((n, k), dtype=torch.int64, device=x.device)
    check(_L().fsf_row_topk_desc(ptr(x), n, w, int(k), ptr(out), stream_ptr()), "fsf_row_topk_desc")
    return out


# ------------------------------------------------------------------------------------- SIR-layer input
def sir_input(points, feats, f_cluster, xyz_normalizer, layers, act: str, rel_div: float, extra=None, extra_div: float = 1.0,
              feats_index=None, direct_parts=()):
    """fsf_sir_input[_gather]: cat(points / normalizer, feats[, extra / extra_div]) * rel_mlp(f_cluster / rel_div) -> f32 [n, C].
    `layers` = three (linear_weight, ln_weight, ln_bias) triples of the position MLP; one eps (taken by the caller).
    `feats` may be a list of up to three tensors standing side by side, and with `feats_index` (i64 [n]) row i of the input takes row
    feats_index[i] of every part — the gather and the concat happen in the kernel's loads."""
    parts = list(feats) if isinstance(feats, (list, tuple)) else [feats]
    require_cuda(points, f_cluster, extra, feats_index, *parts)
    n = points.size(0)
    (w1, g1, b1), (w2, g2, b2), (w3, g3, b3), eps = layers
    fcols = sum(t.size(1) for t in parts)
    c = points.size(1) + fcols + (extra.size(1) if extra is not None else 0)
    assert w3.size(0) == c and w1.size(1) == f_cluster.size(1) and w2.size(1) == w1.size(0) and w3.size(1) == w2.size(0)
    for t in [points, f_cluster, extra] + parts:
        assert t is None or (t.dtype == torch.float32 and t.dim() == 2 and (t.size(0) == 0 or t.stride(1) == 1))
    direct_mask = sum(1 << int(p) for p in direct_parts)  # parts whose rows are the layer's rows already (not read through the index)
    assert 1 <= len(parts) <= 3 and all(t.size(0) == n for i, t in enumerate(parts) if feats_index is None or (direct_mask >> i) & 1)
    if feats_index is not None:
        feats_index = feats_index.to(torch.int64).contiguous()
        assert feats_index.numel() == n
    cpad = (c + 3) // 4 * 4  # rows start 16-byte aligned: the consumer is the fused Linear kernel (K22)
    out_full = torch.empty((n, cpad), dtype=torch.float32, device=points.device)
    out = out_full[:, :c]
    rp = lambda t: c_p(t.data_ptr()) if t is not None and t.numel() else c_p(None)  # noqa: E731  row-strided views pass as is
    st = lambda t: t.stride(0) if t is not None and t.size(0) > 1 else (t.size(1) if t is not None else 0)  # noqa: E731
    k = len(parts)
    fp = (ctypes.c_void_p * k)(*[t.data_ptr() if t.numel() else None for t in parts])
    fs = (ctypes.c_int64 * k)(*[int(st(t)) for t in parts])
    fc = (ctypes.c_int32 * k)(*[int(t.size(1)) for t in parts])
    check(_L().fsf_sir_input_gather(rp(points), st(points), points.size(1), f32_array(xyz_normalizer), fp, fs, fc, k, ptr(feats_index), direct_mask,
                                    rp(extra), st(extra), extra.size(1) if extra is not None else 0, float(extra_div),
                                    rp(f_cluster), st(f_cluster), f_cluster.size(1), float(rel_div),
                                    ptr(w1.contiguous()), ptr(g1), ptr(b1), w1.size(0), ptr(w2.contiguous()), ptr(g2), ptr(b2),
                                    w2.size(0), ptr(w3.contiguous()), ptr(g3), ptr(b3), float(eps),
                                    {"none": 0, "relu": 1, "gelu": 2}[act], n, ptr(out_full), cpad, stream_ptr()),
          "fsf_sir_input_gather")
    return out


_FMT_ATTR = "_fsf_weight_format"


def linear_weight_is_f16(planes: torch.Tensor) -> bool:
    """Which of `linear_prepare_weight`'s two formats a prepared weight is in: the tag the preparing call attached to the tensor
    (until round 5 this was inferred from the buffer's size — one layout change away from dispatching the wrong kernel).
    A buffer without a tag (cloned, built by hand) is refused; the f16 kernels check the tag word of the buffer's own header as well."""
    fmt = getattr(planes, _FMT_ATTR, None)
    if fmt is None:
        raise FsfHipError("prepared Linear weight without a format tag: use the tensor hip_ops.linear_prepare_weight returned")
    return fmt == "f16x3"


def linear_prepare_weight(weight: torch.Tensor, fmt: Optional[str] = None):
    """Linear weight f32 [c, k] -> the opaque fragment planes `linear_norm_act` / `linear_norm_act_segmax` take (u8 tensor).
    fmt "bf16x6": fsf_linear_prepare_weight (exact 3-way bf16 split, six MFMA passes per product); "f16x3": fsf_linear_prepare_weight_f16
    (f16 hi | lo of W * s_w behind a header; the kernel then splits x the same way with a per-row scale: three passes, K22f) — the
    default for more than 32 output channels while `switches.K22F` is on.  The returned tensor carries its format as an attribute
    (`linear_weight_is_f16`); the f16 buffer also holds a tag word in its header that the f16 kernels verify."""
    require_cuda(weight)
    weight = weight.detach().contiguous()
    c, k = weight.shape
    if fmt is None:
        fmt = "f16x3" if switches.K22F and c > 32 and c % 4 == 0 else "bf16x6"
    if fmt == "f16x3":
        return linear_prepare_weight_f16(weight, min(128, c))
    h = _L()
    planes = torch.empty(h.fsf_linear_prepared_weight_bytes(k, c), dtype=torch.uint8, device=weight.device)
    check(h.fsf_linear_prepare_weight(ptr(weight), k, c, ptr(planes), stream_ptr()), "fsf_linear_prepare_weight")
    setattr(planes, _FMT_ATTR, "bf16x6")
    return planes


def linear_norm_act_supported(x: torch.Tensor, out_features: int) -> bool:
    """Shapes the K22 kernel takes: fp32 rows that start 16-byte aligned, output channels a multiple of 4 (LayerNorm inside
    the kernel only up to 128 channels)."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and out_features % 4 == 0
            and (x.size(0) <= 1 or x.stride(0) % 4 == 0) and x.stride(1) == 1 and x.data_ptr() % 16 == 0)


def linear_norm_act(x: torch.Tensor, planes: torch.Tensor, out_features: int, bias=None, norm: str = "none", gamma=None,
                    beta=None, eps: float = 0.0, act: str = "none", out=None, row_add=None, row_add_index=None):
    """fsf_linear_norm_act[_grouped]: act(norm(x @ W^T + bias [+ row_add[row_add_index]])) -> f32 [n, c]; `planes` from
    linear_prepare_weight; norm 'none' | 'ln' | 'affine'.  x (and `out`, if given) may be row-strided views (stride a
    multiple of 4 floats, 16-byte aligned base); row_add f32 [g, c] contiguous, row_add_index i64 [n]."""
    require_cuda(x, planes, bias, gamma, beta, out, row_add, row_add_index)
    n, k = x.shape
    c = int(out_features)
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    assert out.shape == (n, c) and out.dtype == torch.float32 and out.stride(1) == 1
    xs = x.stride(0) if n > 1 else (k + 3) // 4 * 4
    os_ = out.stride(0) if n > 1 else (c + 3) // 4 * 4
    norm_code, act_code = {"none": 0, "ln": 1, "affine": 2}[norm], {"none": 0, "relu": 1, "gelu": 2}[act]
    f16w = linear_weight_is_f16(planes)
    if row_add is None and not f16w:
        check(_L().fsf_linear_norm_act(c_p(x.data_ptr()) if n else c_p(None), n, k, xs, ptr(planes), c, ptr(bias), norm_code,
                                       ptr(gamma), ptr(beta), float(eps), act_code,
                                       c_p(out.data_ptr()) if n else c_p(None), os_, stream_ptr()), "fsf_linear_norm_act")
        return out
    if row_add is not None:
        assert (row_add.dtype == torch.float32 and row_add.dim() == 2 and row_add.size(1) == c and row_add.is_contiguous()
                and row_add_index.dtype == torch.int64 and row_add_index.shape == (n,) and row_add_index.is_contiguous())
    if f16w:  # K22f
        check(_L().fsf_linear_f16w_norm_act_grouped(c_p(x.data_ptr()) if n else c_p(None), n, k, xs, ptr(planes), c, ptr(bias),
                                                    ptr(row_add), ptr(row_add_index), row_add.stride(0) if row_add is not None else 0,
                                                    norm_code, ptr(gamma), ptr(beta), float(eps), act_code,
                                                    c_p(out.data_ptr()) if n else c_p(None), os_, stream_ptr()),
              "fsf_linear_f16w_norm_act_grouped")
        return out
    check(_L().fsf_linear_norm_act_grouped(c_p(x.data_ptr()) if n else c_p(None), n, k, xs, ptr(planes), c, ptr(bias),
                                           ptr(row_add), ptr(row_add_index), row_add.stride(0), norm_code, ptr(gamma),
                                           ptr(beta), float(eps), act_code, c_p(out.data_ptr()) if n else c_p(None), os_,
                                           stream_ptr()), "fsf_linear_norm_act_grouped")
    return out


def linear_norm_act_segmax(x: torch.Tensor, planes: torch.Tensor, out_features: int, seg_ids: torch.Tensor, seg_out: torch.Tensor,
                           bias=None, norm: str = "ln", gamma=None, beta=None, eps: float = 0.0, act: str = "gelu", row_add=None,
                           row_add_index=None, want_rows=True):
    """fsf_linear_norm_act_segmax (K22s): y = act(LayerNorm(x @ W^T + bias [+ row_add[row_add_index]])) for rows SORTED by segment
    (seg_ids i64 [n] nondecreasing, values < m) and, in the same pass, seg_out[s] = max over the rows of segment s of y — `seg_out`
    f32 [m, c] (a column slice of a wider buffer qualifies) must hold -inf on entry.  Returns y f32 [n, c], or None with
    `want_rows=False` (the rows are then never written)."""
    require_cuda(x, planes, bias, gamma, beta, row_add, row_add_index, seg_ids, seg_out)
    n, k = x.shape
    c = int(out_features)
    m = seg_out.size(0)
    assert seg_ids.dtype == torch.int64 and seg_ids.shape == (n,) and seg_ids.is_contiguous() and m >= (1 if n else 0)
    assert seg_out.dtype == torch.float32 and seg_out.shape == (m, c) and seg_out.stride(1) == 1
    out = torch.empty((n, c), dtype=torch.float32, device=x.device) if want_rows else None
    xs = x.stride(0) if n > 1 else (k + 3) // 4 * 4
    norm_code, act_code = {"none": 0, "ln": 1, "affine": 2}[norm], {"none": 0, "relu": 1, "gelu": 2}[act]
    if row_add is not None:
        assert (row_add.dtype == torch.float32 and row_add.dim() == 2 and row_add.size(1) == c and row_add.is_contiguous()
                and row_add_index.dtype == torch.int64 and row_add_index.shape == (n,) and row_add_index.is_contiguous())
    fn = _L().fsf_linear_f16w_norm_act_segmax if linear_weight_is_f16(planes) else _L().fsf_linear_norm_act_segmax
    check(fn(c_p(x.data_ptr()) if n else c_p(None), n, k, xs, ptr(planes), c, ptr(bias), ptr(row_add),
                                          ptr(row_add_index), row_add.stride(0) if row_add is not None else 0, norm_code, ptr(gamma),
                                          ptr(beta), float(eps), act_code, ptr(seg_ids), m,
                                          c_p(seg_out.data_ptr()) if m else c_p(None), seg_out.stride(0) if m > 1 else (c + 3) // 4 * 4,
                                          ptr(out), c, stream_ptr()), "fsf_linear_norm_act_segmax")
    return out


def linear_prepare_weight_sliced(weight: torch.Tensor, nslice: int, slice_c: int):
    """fsf_linear_prepare_weight_sliced: the stacked weights f32 [nslice * slice_c, k] of nslice independent layers -> planes."""
    require_cuda(weight)
    weight = weight.detach().contiguous()
    assert weight.dim() == 2 and weight.size(0) == nslice * slice_c and 1 <= slice_c <= 128
    k = weight.size(1)
    h = _L()
    planes = torch.empty(h.fsf_linear_prepared_weight_sliced_bytes(k, nslice, slice_c), dtype=torch.uint8, device=weight.device)
    check(h.fsf_linear_prepare_weight_sliced(ptr(weight), k, nslice, slice_c, ptr(planes), stream_ptr()),
          "fsf_linear_prepare_weight_sliced")
    return planes


def linear_norm_act_sliced(x: torch.Tensor, k: int, x_slice_offset: int, planes: torch.Tensor, nslice: int, slice_c: int, bias=None,
                           norm: str = "none", gamma=None, beta=None, eps: float = 0.0, act: str = "none", out=None):
    """fsf_linear_norm_act_sliced: nslice independent `act(norm(x_s @ W_s^T + b_s))` layers in one launch -> f32
    [n, nslice * slice_c]; layer s reads the k columns of x from s * x_slice_offset (0: all layers share the input), its
    LayerNorm spans its own slice_c channels.  bias / gamma / beta f32 [nslice * slice_c]."""
    require_cuda(x, planes, bias, gamma, beta, out)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    n, c = x.size(0), nslice * slice_c
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    assert out.shape == (n, c) and out.dtype == torch.float32 and out.stride(1) == 1
    xs = x.stride(0) if n > 1 else (x.size(1) + 3) // 4 * 4
    os_ = out.stride(0) if n > 1 else (c + 3) // 4 * 4
    norm_code, act_code = {"none": 0, "ln": 1, "affine": 2}[norm], {"none": 0, "relu": 1, "gelu": 2}[act]
    check(_L().fsf_linear_norm_act_sliced(c_p(x.data_ptr()) if n else c_p(None), n, int(k), xs, int(x_slice_offset), ptr(planes),
                                          int(nslice), int(slice_c), ptr(bias), norm_code, ptr(gamma), ptr(beta), float(eps),
                                          act_code, c_p(out.data_ptr()) if n else c_p(None), os_, stream_ptr()),
          "fsf_linear_norm_act_sliced")
    return out


class RowPlanes:
    """A matrix f32 [n, c] as K22h's operand: `data` u8 [n * c * 4] = [n][c / 8][2][8] f16 hi | lo of the row-scaled values,
    `inv_scales` f32 [n] (fsf_rows_to_planes)."""

    __slots__ = ("data", "inv_scales", "n", "c")

    def __init__(self, data, inv_scales, n, c):
        self.data, self.inv_scales, self.n, self.c = data, inv_scales, int(n), int(c)

    def rows(self):
        """The matrix back as f32 [n, c]: (hi + lo) * inv_scale, exact (the two halves do not overlap, the scale is a power of two) —
        i.e. the 22-bit rounding of the values the planes were made from.  For a consumer that is not a wide Linear (a
        head whose first layer does not take planes used to raise here); never on the built paths."""
        n, c = self.n, self.c
        d = self.data[: n * c * 4].view(torch.float16).view(n, c // 8, 2, 8).to(torch.float32)
        return (d[:, :, 0, :] + d[:, :, 1, :]).reshape(n, c) * self.inv_scales[:n, None]


def rows_to_planes_supported(x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.size(1) % 8 == 0 and 8 <= x.size(1) <= 2048
            and x.stride(1) == 1 and (x.size(0) <= 1 or x.stride(0) % 4 == 0) and x.data_ptr() % 16 == 0)


def rows_to_planes(x: torch.Tensor, norm: str = "none", gamma=None, beta=None, eps: float = 0.0, act: str = "none", want_rows=False):
    """fsf_rows_to_planes: act(LayerNorm(x)) (or x itself) -> RowPlanes (+ the fp32 rows with `want_rows`)."""
    require_cuda(x, gamma, beta)
    assert rows_to_planes_supported(x)
    n, c = x.shape
    h = _L()
    data = torch.empty(max(int(h.fsf_row_planes_bytes(n, c)), 16), dtype=torch.uint8, device=x.device)
    inv = torch.empty((max(n, 1),), dtype=torch.float32, device=x.device)
    out = torch.empty((n, c), dtype=torch.float32, device=x.device) if want_rows else None
    xs = x.stride(0) if n > 1 else c
    check(h.fsf_rows_to_planes(c_p(x.data_ptr()) if n else c_p(None), n, c, xs, {"none": 0, "ln": 1}[norm], ptr(gamma), ptr(beta),
                               float(eps), {"none": 0, "relu": 1, "gelu": 2}[act], ptr(data), ptr(inv), ptr(out), c, stream_ptr()),
          "fsf_rows_to_planes")
    rp = RowPlanes(data, inv, n, c)
    return (rp, out) if want_rows else rp


def linear_prepare_weight_f16(weight: torch.Tensor, slice_c: int = 128):
    """fsf_linear_prepare_weight_f16: Linear weight f32 [c, k] -> header + f16 hi | lo fragment planes (K22h)."""
    require_cuda(weight)
    weight = weight.detach().contiguous()
    c, k = weight.shape
    h = _L()
    planes = torch.empty(h.fsf_linear_prepared_weight_f16_bytes(k, c, slice_c), dtype=torch.uint8, device=weight.device)
    check(h.fsf_linear_prepare_weight_f16(ptr(weight), k, c, int(slice_c), ptr(planes), stream_ptr()), "fsf_linear_prepare_weight_f16")
    setattr(planes, _FMT_ATTR, "f16x3")
    return planes


def linear_planes_supported(k: int, c: int, slice_c: int = 128) -> bool:
    return k % 32 == 0 and c % 4 == 0 and 64 < slice_c <= 128 and slice_c % 4 == 0


def linear_planes_norm_act(xp: RowPlanes, wplanes: torch.Tensor, out_features: int, slice_c: int = 128, bias=None, norm: str = "none",
                           gamma=None, beta=None, eps: float = 0.0, act: str = "none", out=None):
    """fsf_linear_planes_norm_act (K22h): act(norm(x W^T + bias)) -> f32 [n, c] from x in plane form and f16-prepared weights."""
    require_cuda(xp.data, wplanes, bias, gamma, beta, out)
    n, k, c = xp.n, xp.c, int(out_features)
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=xp.data.device)
    assert out.shape == (n, c) and out.dtype == torch.float32 and out.stride(1) == 1
    os_ = out.stride(0) if n > 1 else (c + 3) // 4 * 4
    check(_L().fsf_linear_planes_norm_act(ptr(xp.data), ptr(xp.inv_scales), n, k, ptr(wplanes), c, int(slice_c), ptr(bias),
                                          {"none": 0, "ln": 1, "affine": 2}[norm], ptr(gamma), ptr(beta), float(eps),
                                          {"none": 0, "relu": 1, "gelu": 2}[act], c_p(out.data_ptr()) if n else c_p(None), os_,
                                          stream_ptr()), "fsf_linear_planes_norm_act")
    return out


# ----------------------------------------------------------------------------------- refine-stage ops
OPT_POOL_BRUTE = 1


def set_option(option: int, value: int) -> int:
    """fsf_set_option: a process-wide algorithm switch of the library; returns the previous value."""
    h = _L()
    old = int(h.fsf_get_option(int(option)))
    check(h.fsf_set_option(int(option), int(value)), "fsf_set_option")
    return old


def dynamic_point_pool(rois: torch.Tensor, pts: torch.Tensor, extra_wlh, max_inbox_point: int, max_all_pts: int = 50000,
                       roi_batch_col: int = -1, box_col: int = 0, pts_batch: Optional[torch.Tensor] = None):
    """fsf_dynamic_point_pool: rois f32 [R, >=7] (box at box_col.., optional batch column), pts f32 [P, >=3] ->
    (pts_idx i64 [k], roi_idx i64 [k], feats f32 [k,13]) in ascending (roi, point) order; k = 0 possible."""
    require_cuda(rois, pts, pts_batch)
    assert rois.dtype == torch.float32 and pts.dtype == torch.float32 and rois.dim() == 2 and pts.dim() == 2
    if rois.stride(1) != 1:
        rois = rois.contiguous()
    if pts.stride(1) != 1:
        pts = pts.contiguous()
    if pts_batch is not None:
        pts_batch = pts_batch.to(torch.int64).contiguous()
    dev = pts.device
    out_pts = torch.empty((max_all_pts,), dtype=torch.int64, device=dev)
    out_roi = torch.empty((max_all_pts,), dtype=torch.int64, device=dev)
    out_feat = torch.empty((max_all_pts, 13), dtype=torch.float32, device=dev)
    count = ctypes.c_int64(0)
    h = _L()
    n_rois, n_pts = rois.size(0), pts.size(0)
    ws = _lib.workspace(h.fsf_dynamic_point_pool_workspace_bytes(n_pts, n_rois), dev)
    # row-strided views (e.g. the xyz columns of [P,5] points) go through as they are: base pointer + row stride
    check(h.fsf_dynamic_point_pool(c_p(rois.data_ptr()), n_rois, rois.stride(0) if n_rois else rois.size(1), box_col,
                                   roi_batch_col, c_p(pts.data_ptr()), n_pts, pts.stride(0) if n_pts else pts.size(1),
                                   ptr(pts_batch),
                                   (ctypes.c_float * 3)(*[float(v) for v in extra_wlh]), int(max_inbox_point),
                                   int(max_all_pts), ptr(out_pts), ptr(out_roi), ptr(out_feat), None,
                                   ctypes.cast(ctypes.pointer(count), c_p), ptr(ws), ws.numel(), stream_ptr()),
          "fsf_dynamic_point_pool")
    k = int(count.value)
    return out_pts[:k], out_roi[:k], out_feat[:k]


def nms_bev(boxes_sorted: torch.Tensor, thresh: float, rotated: bool = True):
    """fsf_nms_bev: boxes f32 [n,5] (x1,y1,x2,y2,yaw) in descending score order -> positions kept (i64, ascending)."""
    require_cuda(boxes_sorted)
    assert boxes_sorted.dtype == torch.float32 and boxes_sorted.dim() == 2 and boxes_sorted.size(1) == 5
    b = boxes_sorted.contiguous()
    n = b.size(0)
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=b.device)
    num = ctypes.c_int64(0)
    h = _L()
    ws = _lib.workspace(h.fsf_nms_bev_workspace_bytes(n), b.device)
    check(h.fsf_nms_bev(ptr(b), n, float(thresh), int(bool(rotated)), ptr(keep), None,
                        ctypes.cast(ctypes.pointer(num), c_p), ptr(ws), ws.numel(), stream_ptr()), "fsf_nms_bev")
    return keep[: int(num.value)]


def nms_bev_multiclass(boxes: torch.Tensor, rank: torch.Tensor, count: torch.Tensor, thresh: float, rotated: bool = True,
                       max_keep: int = 0, windowed: bool = False):
    """fsf_nms_bev_multiclass[_capped]: boxes f32 [n,5] (caller's order), rank i32 [C,n], count i32 [C] ->
    (keep i64 [C,n] kept ranks per class, num_keep i64 [C]), all on the device (no sync); max_keep > 0 stops a class after
    that many kept boxes.  `windowed` (with max_keep): the per-class masks hold only each class's best max(4 max_keep, 2048)
    boxes and a third result, `incomplete` i32 [1], says whether some class ran out of window first (then call again without)."""
    require_cuda(boxes, rank, count)
    assert boxes.dtype == torch.float32 and boxes.dim() == 2 and boxes.size(1) == 5
    assert rank.dtype == torch.int32 and count.dtype == torch.int32 and rank.dim() == 2 and rank.size(1) == boxes.size(0)
    b, rank, count = boxes.contiguous(), rank.contiguous(), count.contiguous()
    n, c = b.size(0), rank.size(0)
    keep = torch.empty((c, max(n, 1)), dtype=torch.int64, device=b.device)
    num = torch.empty((c,), dtype=torch.int64, device=b.device)
    h = _L()
    windowed = bool(windowed) and max_keep > 0
    flag = torch.empty((1,), dtype=torch.int32, device=b.device) if windowed else None
    nbytes = (h.fsf_nms_bev_multiclass_capped_workspace_bytes(n, c, int(max_keep)) if windowed
              else h.fsf_nms_bev_multiclass_workspace_bytes(n, c))
    ws = _lib.workspace(nbytes, b.device)
    check(h.fsf_nms_bev_multiclass_capped(ptr(b), n, c, ptr(rank), ptr(count), float(thresh), int(bool(rotated)), int(max_keep),
                                          ptr(keep), ptr(num), ptr(flag), ptr(ws), ws.numel(), stream_ptr()),
          "fsf_nms_bev_multiclass_capped")
    return (keep, num, flag) if windowed else (keep, num)


def decode_cluster_boxes(cls_logits: torch.Tensor, reg_preds: torch.Tensor, cluster_xyz: torch.Tensor, eps: float):
    """fsf_decode_cluster_boxes: cls_logits f32 [n, C], reg_preds f32 [n, 8 | 10], cluster_xyz f32 [n, 3] ->
    (boxes f32 [n, code - 1], boxes_nms f32 [n, 5] = (x1, y1, x2, y2, yaw), scores_t f32 [C, n] = sigmoid scores, class-major)."""
    require_cuda(cls_logits, reg_preds, cluster_xyz)
    assert cls_logits.dtype == reg_preds.dtype == cluster_xyz.dtype == torch.float32
    n, c, code = cls_logits.size(0), cls_logits.size(1), reg_preds.size(1)
    assert reg_preds.size(0) == n and cluster_xyz.shape == (n, 3)
    cls_logits, reg_preds, cluster_xyz = (t if t.stride(1) == 1 else t.contiguous() for t in (cls_logits, reg_preds, cluster_xyz))
    dev = cls_logits.device
    boxes = torch.empty((n, code - 1), dtype=torch.float32, device=dev)
    boxes_nms = torch.empty((n, 5), dtype=torch.float32, device=dev)
    scores_t = torch.empty((c, n), dtype=torch.float32, device=dev)
    check(_L().fsf_decode_cluster_boxes(c_p(cls_logits.data_ptr()), cls_logits.stride(0), c_p(reg_preds.data_ptr()),
                                        reg_preds.stride(0), c_p(cluster_xyz.data_ptr()), cluster_xyz.stride(0), n, c, code,
                                        float(eps), ptr(boxes), ptr(boxes_nms), ptr(scores_t), stream_ptr()), "fsf_decode_cluster_boxes")
    return boxes, boxes_nms, scores_t


def class_rank_desc(scores_t: torch.Tensor, score_thr: float):
    """fsf_class_rank_desc: scores_t f32 [C, n] -> (order i32 [C, n], rank i32 [C, n], count i32 [C]) — per class the stable
    descending score order of the boxes above score_thr (then the others, by index), each box's position in it (-1 under the
    threshold) and the number of boxes above it."""
    require_cuda(scores_t)
    assert scores_t.dtype == torch.float32 and scores_t.dim() == 2 and scores_t.is_contiguous()
    c, n = scores_t.shape
    dev = scores_t.device
    order = torch.empty((c, n), dtype=torch.int32, device=dev)
    rank = torch.empty((c, n), dtype=torch.int32, device=dev)
    count = torch.empty((c,), dtype=torch.int32, device=dev)
    h = _L()
    ws = _lib.workspace(h.fsf_class_rank_desc_workspace_bytes(n, c), dev)
    check(h.fsf_class_rank_desc(ptr(scores_t), n, c, float(score_thr), ptr(order), ptr(rank), ptr(count), ptr(ws), ws.numel(),
                                stream_ptr()), "fsf_class_rank_desc")
    return order, rank, count


def cluster_key_survival(new_keys: torch.Tensor, cnt: torch.Tensor, inv: torch.Tensor, batch_size: int, min_points: int, num_groups: int):
    """fsf_cluster_key_survival (K25): ClusterAssigner's density filter for all class groups at once.  new_keys i64 [m, k] (ascending, column 0 =
    group * batch_size + sample), cnt i64 [m], inv i64 [n] -> (k_idx i64 [mk], k_group i32 [mk], v_idx i64 [nv], vox_inv i64 [nv]): surviving
    keys and their class groups, surviving pairs (both ascending) and each surviving pair's position among the surviving keys.  One host sync."""
    require_cuda(new_keys, cnt, inv)
    assert new_keys.dtype == torch.int64 and new_keys.dim() == 2 and new_keys.is_contiguous()
    assert cnt.dtype == torch.int64 and inv.dtype == torch.int64
    cnt, inv = cnt.contiguous(), inv.contiguous()
    m, n = new_keys.size(0), inv.numel()
    assert cnt.numel() == m
    dev = new_keys.device
    k_idx = torch.empty((max(m, 1),), dtype=torch.int64, device=dev)
    k_group = torch.empty((max(m, 1),), dtype=torch.int32, device=dev)
    v_idx = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    vox_inv = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    counts = (ctypes.c_int64 * 2)(0, 0)
    h = _L()
    ws = _lib.workspace(h.fsf_cluster_key_survival_workspace_bytes(m, n), dev)
    check(h.fsf_cluster_key_survival(ptr(new_keys), new_keys.size(1), ptr(cnt), m, ptr(inv), n, int(batch_size), int(min_points),
                                     int(num_groups), ptr(k_idx), ptr(k_group), ptr(v_idx), ptr(vox_inv), ctypes.cast(counts, c_p), ptr(ws),
                                     ws.numel(), stream_ptr()), "fsf_cluster_key_survival")
    mk, nv = int(counts[0]), int(counts[1])
    return k_idx[:mk], k_group[:mk], v_idx[:nv], vox_inv[:nv]


def cluster_point_ids(labels: torch.Tensor, vox_group: torch.Tensor, vox_inv: torch.Tensor, g_ids: torch.Tensor, b_pts: torch.Tensor,
                      num_groups: int):
    """fsf_cluster_point_ids: (group, sample, cluster id within the group) i64 [nv, 3] of every surviving pair from the grouped
    connected-component labels i32 [m] of the (group-sorted) cluster voxels."""
    require_cuda(labels, vox_group, vox_inv, g_ids, b_pts)
    assert labels.dtype == torch.int32 and vox_group.dtype == torch.int32 and labels.numel() == vox_group.numel()
    assert vox_inv.dtype == torch.int64 and g_ids.dtype == torch.int64 and b_pts.dtype == torch.int64
    labels, vox_group, vox_inv, g_ids, b_pts = (t.contiguous() for t in (labels, vox_group, vox_inv, g_ids, b_pts))
    nv = vox_inv.numel()
    assert g_ids.numel() == nv and b_pts.numel() == nv
    out = torch.empty((nv, 3), dtype=torch.int64, device=labels.device)
    ws = _lib.workspace(256, labels.device)
    check(_L().fsf_cluster_point_ids(ptr(labels), ptr(vox_group), labels.numel(), ptr(vox_inv), ptr(g_ids), ptr(b_pts), nv, int(num_groups),
                                     ptr(out), ptr(ws), ws.numel(), stream_ptr()), "fsf_cluster_point_ids")
    return out


def nms_select_capacity() -> int:
    return int(_L().fsf_nms_select_capacity())


def box_tail_max_classes() -> int:
    return int(_L().fsf_box_tail_max_classes())


def nms_select(boxes: torch.Tensor, scores_t: torch.Tensor, order: torch.Tensor, keep: torch.Tensor, num_keep: torch.Tensor,
               max_keep: int, max_num: int, label_lut: Optional[torch.Tensor] = None, incomplete: Optional[torch.Tensor] = None):
    """fsf_nms_select: the kept boxes of every class (keep / num_keep of nms_bev_multiclass run with the cap max_keep) -> ONE f32
    buffer of max_num * (box_dim + 2) + 4 words: rows (box | score | label) and, in the last four words (as i32), rows written, boxes
    kept over all classes, the `incomplete` flag.  No sync: bring it to the host with one copy."""
    require_cuda(boxes, scores_t, order, keep, num_keep, label_lut, incomplete)
    assert boxes.dtype == torch.float32 and boxes.dim() == 2 and boxes.is_contiguous() and scores_t.is_contiguous()
    assert order.dtype == torch.int32 and order.is_contiguous() and keep.dtype == torch.int64 and keep.stride(1) == 1
    assert num_keep.dtype == torch.int64 and (label_lut is None or (label_lut.dtype == torch.int64 and label_lut.is_contiguous()))
    c, n = scores_t.shape
    d = boxes.size(1)
    w = d + 2
    buf = torch.empty((int(max_num) * w + 4,), dtype=torch.float32, device=boxes.device)
    meta = buf[int(max_num) * w:].view(torch.int32)
    check(_L().fsf_nms_select(ptr(boxes), d, ptr(scores_t), ptr(order), ptr(keep), keep.stride(0), ptr(num_keep), n, c, int(max_keep),
                              int(max_num), ptr(label_lut), ptr(incomplete), ptr(buf), ptr(meta), stream_ptr()), "fsf_nms_select")
    return buf


# ------------------------------------------------------------------------------ connected components
def connected_components(points: torch.Tensor, dist: float, batch_idx: Optional[torch.Tensor] = None):
    """fsf_connected_components: labels i32 [n] (0..K-1 in order of each component's first member)."""
    require_cuda(points, batch_idx)
    assert points.dtype == torch.float32 and points.dim() == 2 and points.size(1) >= 2
    points = points.contiguous()
    n = points.size(0)
    if batch_idx is not None:
        batch_idx = batch_idx.to(torch.int32).contiguous()
    labels = torch.empty((n,), dtype=torch.int32, device=points.device)
    h = _L()
    ws = _lib.workspace(h.fsf_connected_components_workspace_bytes(n), points.device)
    check(h.fsf_connected_components(ptr(points), n, points.size(1), ptr(batch_idx), float(dist), ptr(labels), None,
                                     ptr(ws), ws.numel(), stream_ptr()), "fsf_connected_components")
    return labels


def connected_components_grouped(points: torch.Tensor, group_idx: torch.Tensor, dist_table: torch.Tensor):
    """fsf_connected_components_grouped: per-group distance thresholds, no adjacency across groups; labels i32 [n]
    numbered by first member over all points."""
    require_cuda(points, group_idx, dist_table)
    assert points.dtype == torch.float32 and points.dim() == 2 and points.size(1) >= 2
    points = points.contiguous()
    group_idx = group_idx.to(torch.int32).contiguous()
    dist_table = dist_table.to(torch.float32).contiguous()
    n = points.size(0)
    labels = torch.empty((n,), dtype=torch.int32, device=points.device)
    h = _L()
    ws = _lib.workspace(h.fsf_connected_components_workspace_bytes(n), points.device)
    check(h.fsf_connected_components_grouped(ptr(points), n, points.size(1), ptr(group_idx), ptr(dist_table), dist_table.numel(),
                                             ptr(labels), None, ptr(ws), ws.numel(), stream_ptr()),
          "fsf_connected_components_grouped")
    return labels


# ------------------------------------------------------------------------------------ norm + activation
_ACTS = {None: 0, "none": 0, "relu": 1, "gelu": 2}


def norm_act(x: torch.Tensor, gamma, beta, eps: float, norm: str, act, inplace=True, out: Optional[torch.Tensor] = None):
    """fsf_norm_act: LayerNorm ('ln') or per-channel affine ('affine') fused with ReLU/GELU; x f32 [n,c].
    `out` may be a column slice of a wider row-major buffer."""
    require_cuda(x, out)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    n, c = x.shape
    if out is None:
        out = x if inplace else torch.empty_like(x)
    assert out.shape == (n, c) and out.dtype == torch.float32 and out.stride(1) == 1 and out.stride(0) >= c
    check(_L().fsf_norm_act(ptr(x), n, c, ptr(gamma), ptr(beta), float(eps), {"ln": 0, "affine": 1}[norm], _ACTS[act],
                            c_p(out.data_ptr()), out.stride(0), stream_ptr()), "fsf_norm_act")
    return out


def column_sum(x: torch.Tensor):
    """fsf_column_stats (sums only): x f32 [n,c] -> f32 [c], fixed summation order (the bias gradient of a per-point Linear)."""
    require_cuda(x)
    x = x.contiguous()
    n, c = x.shape
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    h = _L()
    ws = _lib.workspace(h.fsf_column_stats_workspace_bytes(c), x.device)
    check(h.fsf_column_stats(ptr(x), n, c, ptr(out), None, ptr(ws), ws.numel(), stream_ptr()), "fsf_column_stats")
    return out


def column_mean_var(x: torch.Tensor):
    """fsf_column_stats: x f32 [n,c] -> (mean [c], biased variance [c]), two passes."""
    require_cuda(x)
    x = x.contiguous()
    n, c = x.shape
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    var = torch.empty(c, dtype=torch.float32, device=x.device)
    h = _L()
    ws = _lib.workspace(h.fsf_column_stats_workspace_bytes(c), x.device)
    check(h.fsf_column_stats(ptr(x), n, c, ptr(mean), ptr(var), ptr(ws), ws.numel(), stream_ptr()), "fsf_column_stats")
    return mean, var


def batch_norm_train_stats(x: torch.Tensor, weight, bias, eps: float, momentum: float, running_mean=None, running_var=None):
    """fsf_batch_norm_train_stats: x f32 [n,c] -> (mean, invstd, scale, shift) [c] each; running_mean / running_var (if given) are
    updated in place with `momentum` (variance unbiased by n / (n - 1))."""
    require_cuda(x, weight, bias, running_mean, running_var)
    x = x.contiguous()
    n, c = x.shape
    assert n >= 1
    mean, invstd, scale, shift = (torch.empty(c, dtype=torch.float32, device=x.device) for _ in range(4))
    h = _L()
    ws = _lib.workspace(h.fsf_column_stats_workspace_bytes(c), x.device)
    for t in (weight, bias, running_mean, running_var):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == c)
    check(h.fsf_batch_norm_train_stats(ptr(x), n, c, ptr(weight), ptr(bias), float(eps), float(1.0 - momentum), float(momentum),
                                       float(momentum * n / max(n - 1, 1)), ptr(running_mean), ptr(running_var), ptr(mean), c_p(None),
                                       ptr(invstd), ptr(scale), ptr(shift), ptr(ws), ws.numel(), stream_ptr()),
          "fsf_batch_norm_train_stats")
    return mean, invstd, scale, shift


def batch_norm_act_forward(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, relu: bool):
    """fsf_batch_norm_act_forward: [relu] fma(x, scale, shift); x f32 [n,c]."""
    require_cuda(x, scale, shift)
    x = x.contiguous()
    n, c = x.shape
    out = torch.empty_like(x)
    check(_L().fsf_batch_norm_act_forward(ptr(x), n, c, ptr(scale.contiguous()), ptr(shift.contiguous()), int(bool(relu)),
                                          ptr(out), stream_ptr()), "fsf_batch_norm_act_forward")
    return out


def batch_norm_act_backward(x, grad_out, mean, invstd, scale, shift, relu: bool):
    """fsf_batch_norm_act_backward -> (grad_x [n,c], grad_gamma [c], grad_beta [c])."""
    require_cuda(x, grad_out, mean, invstd)
    x, grad_out = x.contiguous(), grad_out.contiguous()
    n, c = x.shape
    gx = torch.empty_like(x)
    dg = torch.empty(c, dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device)
    h = _L()
    ws = _lib.workspace(h.fsf_column_stats_workspace_bytes(c), x.device)
    check(h.fsf_batch_norm_act_backward(ptr(x), ptr(grad_out), n, c, ptr(mean), ptr(invstd),
                                        ptr(scale.contiguous()) if scale is not None else None,
                                        ptr(shift.contiguous()) if shift is not None else None, int(bool(relu)), ptr(gx),
                                        ptr(dg), ptr(db), ptr(ws), ws.numel(), stream_ptr()), "fsf_batch_norm_act_backward")
    return gx, dg, db


def norm_act_backward(x: torch.Tensor, grad_out: torch.Tensor, gamma, beta, eps: float, act):
    """fsf_norm_act_backward (LayerNorm form): -> (grad_x [n,c], grad_gamma [c] | None, grad_beta [c] | None)."""
    require_cuda(x, grad_out)
    x, grad_out = x.contiguous(), grad_out.contiguous()
    n, c = x.shape
    gx = torch.empty_like(x)
    dg = torch.empty((c,), dtype=torch.float32, device=x.device) if gamma is not None else None
    db = torch.empty((c,), dtype=torch.float32, device=x.device) if beta is not None else None
    h = _L()
    ws = _lib.workspace(h.fsf_norm_act_backward_workspace_bytes(c), x.device)
    check(h.fsf_norm_act_backward(ptr(x), ptr(grad_out), n, c, ptr(gamma), ptr(beta), float(eps), _ACTS[act], ptr(gx), ptr(dg),
                                  ptr(db), ptr(ws), ws.numel(), stream_ptr()), "fsf_norm_act_backward")
    return gx, dg, db


# ------------------------------------------------------------------------------- K29: query-stage glue (csrc/query_glue.hip)
def _f32_rows(t):
    """(data pointer, row stride) of an f32 [n, c] tensor whose rows are contiguous (column slices of wider buffers qualify)."""
    assert t.dtype == torch.float32 and t.dim() == 2
    if t.stride(1) != 1 and t.size(1) > 1:
        t = t.contiguous()
    return t, c_p(t.data_ptr()) if t.numel() else c_p(None), int(t.stride(0)) if t.size(0) > 1 else int(max(t.size(1), 1))


def sorted_rows(order, inv, points, f_cluster=None, centers=None, index=None, fill=None, fill_value=float("-inf")):
    """fsf_sorted_rows: the operands of a SIR stack in the unique's sort order -> (seg_ids i64 [n], points [n, c], f_cluster [n, 3],
    index i64 [n]); `fill` (a contiguous f32 tensor, e.g. the stack's group table) is set to `fill_value` in the same launch.
    f_cluster=None: points[:, :3] - centers[inv]."""
    require_cuda(order, inv, points, f_cluster, centers, index, fill)
    assert order.dtype == torch.int32 and inv.dtype == torch.int64 and order.is_contiguous() and inv.is_contiguous()
    n = order.numel()
    pts, pp, ps = _f32_rows(points)
    dev = points.device
    fc = cc = None
    fp, fs, cp_, cs = c_p(None), 3, c_p(None), 3
    if f_cluster is not None:
        fc, fp, fs = _f32_rows(f_cluster)
        assert fc.shape == (n, 3)
    else:
        cc, cp_, cs = _f32_rows(centers)
        assert cc.size(1) >= 3
    if index is not None:
        index = index.to(torch.int64).contiguous()
    seg_ids = torch.empty((n,), dtype=torch.int64, device=dev)
    idx_s = torch.empty((n,), dtype=torch.int64, device=dev)
    pts_s = torch.empty((n, pts.size(1)), dtype=torch.float32, device=dev)
    fcl_s = torch.empty((n, 3), dtype=torch.float32, device=dev)
    if fill is not None:
        assert fill.dtype == torch.float32 and fill.is_contiguous()
    check(_L().fsf_sorted_rows(ptr(order), ptr(inv), n, pp, ps, pts.size(1), fp, fs, cp_, cs, ptr(index), ptr(seg_ids), ptr(pts_s), ptr(fcl_s),
                               ptr(idx_s), ptr(fill), fill.numel() if fill is not None else 0, float(fill_value), stream_ptr()),
          "fsf_sorted_rows")
    return seg_ids, pts_s, fcl_s, idx_s


class _SirLayerC(ctypes.Structure):  # FsfSirLayer (include/fsf_hip.h)
    _fields_ = [("planes_left", ctypes.c_void_p), ("planes_right", ctypes.c_void_p), ("left_f16", ctypes.c_int32), ("right_f16", ctypes.c_int32),
                ("bias", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("eps", ctypes.c_float),
                ("norm", ctypes.c_int32), ("act", ctypes.c_int32), ("c", ctypes.c_int32)]


class _SirBlockC(ctypes.Structure):  # FsfSirBlock
    _fields_ = [("w1", ctypes.c_void_p), ("g1", ctypes.c_void_p), ("b1", ctypes.c_void_p), ("w2", ctypes.c_void_p), ("g2", ctypes.c_void_p),
                ("b2", ctypes.c_void_p), ("w3", ctypes.c_void_p), ("g3", ctypes.c_void_p), ("b3", ctypes.c_void_p),
                ("h1", ctypes.c_int32), ("h2", ctypes.c_int32), ("mlp_eps", ctypes.c_float), ("mlp_act", ctypes.c_int32),
                ("xyz_normalizer", ctypes.c_float * 3), ("rel_div", ctypes.c_float), ("in_cols", ctypes.c_int32),
                ("num_layers", ctypes.c_int32), ("layer", _SirLayerC * 4)]


class SirStackDescriptor:
    """The HOST side of fsf_sir_stack_forward's `blocks` argument: one FsfSirBlock per SIRLayer / DynamicClusterVFE of a stack.
    `blocks` = [dict(mlp=((w1, g1, b1), (w2, g2, b2), (w3, g3, b3)), mlp_eps, mlp_act, xyz_normalizer, rel_div, in_cols,
    layers=[dict(planes_left, planes_right | None, bias, gamma, beta, eps, norm, act, c)])].  Every tensor is kept alive here."""

    def __init__(self, blocks):
        codes_n, codes_a = {"none": 0, "ln": 1, "affine": 2}, {"none": 0, "relu": 1, "gelu": 2}
        self.keep = []
        self.num_blocks = len(blocks)
        self.widths = [[int(l["c"]) for l in b["layers"]] for b in blocks]
        arr = (_SirBlockC * len(blocks))()

        def dp(t):
            if t is None:
                return None
            assert t.is_cuda and t.is_contiguous()
            self.keep.append(t)
            return t.data_ptr()

        for k, b in zip(arr, blocks):
            (w1, g1, b1), (w2, g2, b2), (w3, g3, b3) = b["mlp"]
            for name, t in (("w1", w1), ("g1", g1), ("b1", b1), ("w2", w2), ("g2", g2), ("b2", b2), ("w3", w3), ("g3", g3), ("b3", b3)):
                assert t.dtype == torch.float32
                setattr(k, name, dp(t.contiguous()))
            assert w2.size(1) == w1.size(0) and w3.size(1) == w2.size(0) and w3.size(0) == b["in_cols"]
            k.h1, k.h2, k.mlp_eps, k.mlp_act = w1.size(0), w2.size(0), float(b["mlp_eps"]), codes_a[b["mlp_act"]]
            k.xyz_normalizer = (ctypes.c_float * 3)(*[float(v) for v in b["xyz_normalizer"]])
            k.rel_div, k.in_cols, k.num_layers = float(b["rel_div"]), int(b["in_cols"]), len(b["layers"])
            assert 1 <= len(b["layers"]) <= 4
            for i, l in enumerate(b["layers"]):
                L = k.layer[i]
                L.planes_left, L.left_f16 = dp(l["planes_left"]), int(linear_weight_is_f16(l["planes_left"]))
                if l.get("planes_right") is not None:
                    L.planes_right, L.right_f16 = dp(l["planes_right"]), int(linear_weight_is_f16(l["planes_right"]))
                L.bias, L.gamma, L.beta = dp(l.get("bias")), dp(l.get("gamma")), dp(l.get("beta"))
                L.eps, L.norm, L.act, L.c = float(l["eps"]), codes_n[l["norm"]], codes_a[l["act"]], int(l["c"])
        self.blocks = arr


def sir_stack_forward(desc: SirStackDescriptor, points, feats, f_cluster, seg_ids, groups, want_rows, extra=None, extra_div: float = 1.0,
                      feats_index=None, direct_parts=()):
    """fsf_sir_stack_forward (K31): every block of a SIR stack on rows sorted by group in ONE native call.  Arguments as
    `sir_input` takes them for the stack's first block (`feats`: tensor or up to three side by side, through `feats_index`), `seg_ids`
    i64 [n] nondecreasing, `groups` f32 [m, sum of all layers' widths] holding -inf.  Returns the last layer's rows f32 [n, c] or None."""
    parts = list(feats) if isinstance(feats, (list, tuple)) else [feats]
    require_cuda(points, f_cluster, extra, feats_index, seg_ids, groups, *parts)
    n, m = points.size(0), groups.size(0)
    for t in [points, f_cluster, extra] + parts:
        assert t is None or (t.dtype == torch.float32 and t.dim() == 2 and (t.size(0) == 0 or t.stride(1) == 1))
    assert n >= 1 and m >= 1 and seg_ids.dtype == torch.int64 and seg_ids.shape == (n,) and seg_ids.is_contiguous()
    assert groups.dtype == torch.float32 and groups.stride(1) == 1 and groups.size(1) >= sum(sum(w) for w in desc.widths)
    direct_mask = sum(1 << int(p) for p in direct_parts)
    assert 1 <= len(parts) <= 3 and all(t.size(0) == n for i, t in enumerate(parts) if feats_index is None or (direct_mask >> i) & 1)
    if feats_index is not None:
        feats_index = feats_index.to(torch.int64).contiguous()
        assert feats_index.numel() == n
    rp = lambda t: c_p(t.data_ptr()) if t is not None and t.numel() else c_p(None)  # noqa: E731
    st = lambda t: t.stride(0) if t is not None and t.size(0) > 1 else (t.size(1) if t is not None else 0)  # noqa: E731
    k = len(parts)
    fp = (ctypes.c_void_p * k)(*[t.data_ptr() if t.numel() else None for t in parts])
    fs = (ctypes.c_int64 * k)(*[int(st(t)) for t in parts])
    fc = (ctypes.c_int32 * k)(*[int(t.size(1)) for t in parts])
    h = _L()
    nbytes = int(h.fsf_sir_stack_arena_bytes(desc.blocks, desc.num_blocks, n, m))
    arena = torch.empty((nbytes,), dtype=torch.uint8, device=points.device)
    rows = torch.empty((n, desc.widths[-1][-1]), dtype=torch.float32, device=points.device) if want_rows else None
    check(h.fsf_sir_stack_forward(desc.blocks, desc.num_blocks, rp(points), st(points), points.size(1), fp, fs, fc, k, ptr(feats_index), direct_mask,
                                  rp(extra), st(extra), extra.size(1) if extra is not None else 0, float(extra_div), rp(f_cluster), st(f_cluster),
                                  f_cluster.size(1), ptr(seg_ids), n, m, c_p(groups.data_ptr()), groups.stride(0) if m > 1 else groups.size(1),
                                  ptr(rows), ptr(arena), nbytes, stream_ptr()), "fsf_sir_stack_forward")
    return rows


def compact_pairs(means, k_idx, g_ids, p_ids, b_pts, centers, v_idx):
    """fsf_compact_pairs -> (means[k_idx] [nk, 3], g_ids[v_idx], p_ids[v_idx], b_pts[v_idx], centers[v_idx] [nv, 3])."""
    require_cuda(means, k_idx, g_ids, p_ids, b_pts, centers, v_idx)
    mm, mp, ms = _f32_rows(means)
    centers = centers.contiguous()
    assert centers.dtype == torch.float32 and centers.size(1) == 3 and mm.size(1) >= 3
    k_idx, v_idx = k_idx.to(torch.int64).contiguous(), v_idx.to(torch.int64).contiguous()
    g_ids, p_ids, b_pts = (t.to(torch.int64).contiguous() for t in (g_ids, p_ids, b_pts))
    nk, nv, dev = k_idx.numel(), v_idx.numel(), means.device
    vox = torch.empty((nk, 3), dtype=torch.float32, device=dev)
    go, po, bo = (torch.empty((nv,), dtype=torch.int64, device=dev) for _ in range(3))
    co = torch.empty((nv, 3), dtype=torch.float32, device=dev)
    check(_L().fsf_compact_pairs(mp, ms, ptr(k_idx), nk, ptr(vox), ptr(g_ids), ptr(p_ids), ptr(b_pts), ptr(centers), ptr(v_idx), nv, ptr(go),
                                 ptr(po), ptr(bo), ptr(co), stream_ptr()), "fsf_compact_pairs")
    return vox, go, po, bo, co


def combine_queries(f_centers, l_centers, f_coors, l_coors, f_preds_2d, begin_idx):
    """fsf_combine_queries -> (obj_centers f32 [m, 3], obj_coors i64 [m, 3], preds_2d f32 [m, d]) of FSF.combine_frustum_and_fsd."""
    require_cuda(f_centers, l_centers, f_coors, l_coors, f_preds_2d)
    f_centers, l_centers, f_preds_2d = f_centers.contiguous(), l_centers.contiguous(), f_preds_2d.contiguous()
    f_coors, l_coors = f_coors.contiguous(), l_coors.contiguous()
    mf, ml, d, dev = f_centers.size(0), l_centers.size(0), f_preds_2d.size(1), f_centers.device
    centers = torch.empty((mf + ml, 3), dtype=torch.float32, device=dev)
    coors = torch.empty((mf + ml, 3), dtype=torch.int64, device=dev)
    preds = torch.empty((mf + ml, d), dtype=torch.float32, device=dev)
    check(_L().fsf_combine_queries(ptr(f_centers), mf, ptr(l_centers), ml, ptr(f_coors), ptr(l_coors), ptr(f_preds_2d), d, int(begin_idx),
                                   ptr(centers), ptr(coors), ptr(preds), stream_ptr()), "fsf_combine_queries")
    return centers, coors, preds


def decode_rois(reg_preds, centers, batch, eps):
    """fsf_decode_rois: rois f32 [m, code] = (batch, decode(reg_preds, centers)); `batch` i64 [m] (any stride)."""
    require_cuda(reg_preds, centers, batch)
    rg, rp, rs = _f32_rows(reg_preds)
    ct, cp_, cs = _f32_rows(centers)
    assert batch.dtype == torch.int64 and batch.dim() == 1 and batch.numel() == rg.size(0) == ct.size(0)
    m, code = rg.shape
    rois = torch.empty((m, code), dtype=torch.float32, device=rg.device)
    check(_L().fsf_decode_rois(rp, rs, code, cp_, cs, c_p(batch.data_ptr()) if m else c_p(None), int(batch.stride(0)) if m > 1 else 1, m,
                               float(eps), ptr(rois), stream_ptr()), "fsf_decode_rois")
    return rois


def refine_rows(info, points, pts_idx, roi_idx, roi_xyz):
    """fsf_refine_rows -> (points[pts_idx] f32 [k, c], f_cluster f32 [k, 13] = cat(info[:, 3:13], points[pts_idx, :3] - roi_xyz[roi_idx]))."""
    require_cuda(info, points, pts_idx, roi_idx, roi_xyz)
    info = info.contiguous()
    pts, pp, ps = _f32_rows(points)
    rx, rp, rs = _f32_rows(roi_xyz)
    assert info.dtype == torch.float32 and info.size(1) == 13 and rx.size(1) >= 3
    pts_idx, roi_idx = pts_idx.to(torch.int64).contiguous(), roi_idx.to(torch.int64).contiguous()
    k = info.size(0)
    out = torch.empty((k, pts.size(1)), dtype=torch.float32, device=info.device)
    fcl = torch.empty((k, 13), dtype=torch.float32, device=info.device)
    check(_L().fsf_refine_rows(ptr(info), pp, ps, pts.size(1), ptr(pts_idx), ptr(roi_idx), rp, rs, k, ptr(out), ptr(fcl), stream_ptr()),
          "fsf_refine_rows")
    return out, fcl


def encode_preds_2d(mask_anno, obj_coors, num_classes, img_w, img_h):
    """fsf_encode_preds_2d (ONE sample): mask_anno f32 [A, D], obj_coors i64 [m, 3] -> (preds_2d f32 [m, D], encoded f32 [m, 6 + classes])."""
    require_cuda(mask_anno, obj_coors)
    anno = mask_anno.to(torch.float32).contiguous()
    coors = obj_coors.to(torch.int64).contiguous()
    assert anno.dim() == 2 and coors.dim() == 2 and coors.size(1) == 3
    m, d = coors.size(0), anno.size(1)
    preds = torch.empty((m, d), dtype=torch.float32, device=anno.device)
    w = 6 + int(num_classes)
    enc = torch.empty((m, w), dtype=torch.float32, device=anno.device)
    check(_L().fsf_encode_preds_2d(ptr(anno), anno.size(0), d, ptr(coors), m, int(num_classes), float(img_w), float(img_h), ptr(preds),
                                   ptr(enc), w, stream_ptr()), "fsf_encode_preds_2d")
    return preds, enc


def weighted_xyz(points, weights, weight_min=1e-5):
    """fsf_weighted_xyz: cat([points[:, :3] * w, w], 1) with w = weights.clamp(min=weight_min); f32 [n, 4]."""
    require_cuda(points, weights)
    pts, pp, ps = _f32_rows(points)
    w = weights.reshape(-1).contiguous()
    assert w.dtype == torch.float32 and w.numel() == pts.size(0)
    out = torch.empty((pts.size(0), 4), dtype=torch.float32, device=pts.device)
    check(_L().fsf_weighted_xyz(pp, ps, ptr(w), pts.size(0), float(weight_min), ptr(out), stream_ptr()), "fsf_weighted_xyz")
    return out


def centroid_divide(mean):
    """fsf_centroid_divide: mean f32 [m, 4] -> mean[:, :3] / mean[:, 3:4]."""
    require_cuda(mean)
    mean = mean.contiguous()
    assert mean.dtype == torch.float32 and mean.dim() == 2 and mean.size(1) == 4
    out = torch.empty((mean.size(0), 3), dtype=torch.float32, device=mean.device)
    check(_L().fsf_centroid_divide(ptr(mean), mean.size(0), ptr(out), stream_ptr()), "fsf_centroid_divide")
    return out


# ------------------------------------------------------------------- K30: the LiDAR-query clustering front end (csrc/lidar_frontend.hip)
def lidar_cluster_frontend(scores, thresh, group_cols, logits, offsets, points, batch_idx, num_classes, group_voxel_sizes, range_min,
                           key_min, key_max, min_points, dist_table):
    """fsf_lidar_cluster_frontend (one sample): group_sample + ClusterAssigner + the SIR stack's unique + the cluster centroids as ONE
    native call.  Returns dict(p_ids i64 [V], centers f32 [V, 3], cluster_inds i64 [V, 3], points f32 [V, c], new_coors i64 [C, 3],
    plan SegmentPlan over cluster_inds, cluster_xyz f32 [C, 3], counts) — views of one arena tensor."""
    require_cuda(scores, thresh, logits, offsets, points, batch_idx, dist_table)
    sc, sp, ss = _f32_rows(scores)
    lg, lp, ls = _f32_rows(logits)
    of, op, os_ = _f32_rows(offsets)
    pt, pp, ps = _f32_rows(points)
    m, ng = sc.size(0), len(group_cols)
    assert thresh.dtype == torch.float32 and thresh.numel() == ng and dist_table.dtype == torch.float32 and dist_table.numel() == ng
    assert lg.size(0) == of.size(0) == pt.size(0) == m and all(1 <= len(c) <= 2 for c in group_cols)
    masks = (ctypes.c_uint32 * ng)(*[sum(1 << int(c) for c in cs) for cs in group_cols])
    vs = (ctypes.c_float * (3 * ng))(*[float(v) for row in group_voxel_sizes for v in row])
    if batch_idx is not None:
        batch_idx = batch_idx.to(torch.int64).contiguous()
    h = _L()
    pc = pt.size(1)
    nbytes = int(h.fsf_lidar_cluster_frontend_arena_bytes(m, ng, pc))
    arena = torch.empty((nbytes,), dtype=torch.uint8, device=pt.device)
    assert arena.data_ptr() % 256 == 0
    out = (ctypes.c_int64 * 16)()
    check(h.fsf_lidar_cluster_frontend(sp, m, int(num_classes), ss, ptr(thresh.contiguous()), ng, masks, lp, ls, op, os_, pp, ps, pc,
                                       ptr(batch_idx), vs, f32_array(range_min), i64_array(key_min), i64_array(key_max), int(min_points),
                                       ptr(dist_table.contiguous()), ptr(arena), nbytes, ctypes.cast(out, c_p), stream_ptr()),
          "fsf_lidar_cluster_frontend")
    P, K, Kk, V, C = (int(out[i]) for i in range(5))

    def view(slot, dtype, shape):
        off, n = int(out[slot]), 1
        for d in shape:
            n *= d
        return arena[off:off + n * torch.empty((), dtype=dtype).element_size()].view(dtype).view(shape)

    inv = view(10, torch.int64, (V,))
    plan = SegmentPlan(inv=inv, order=view(12, torch.int32, (V,)), seg_offsets=view(13, torch.int32, (C + 1,)), m=C, cnt=view(11, torch.int64, (C,)))
    return dict(p_ids=view(5, torch.int64, (V,)), centers=view(6, torch.float32, (V, 3)), cluster_inds=view(7, torch.int64, (V, 3)),
                points=view(8, torch.float32, (V, pc)), new_coors=view(9, torch.int64, (C, 3)), plan=plan,
                cluster_xyz=view(14, torch.float32, (C, 3)), counts=dict(pairs=P, keys=K, kept_keys=Kk, rows=V, clusters=C))
