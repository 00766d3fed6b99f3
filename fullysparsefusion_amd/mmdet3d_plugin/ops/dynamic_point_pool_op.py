"""`dynamic_point_pool`, same name / signature / return convention as the reference's op
(projects/mmdet3d_plugin/ops/dynamic_point_pool_op.py:10-51, exported at ops/__init__.py:1), which wraps the
un-vendored TorchEx `dynamic_point_pool_ext.forward`.

    out_pts_idx, out_roi_idx, out_pts_feats = dynamic_point_pool(rois, pts, extra_wlh, max_inbox_point, max_all_pts=50000)

* `rois` f32 [N, 7] (x, y, z_bottom, w, l, h, rz) of ONE sample, `pts` f32 [npoints, 3];
* outputs are int64 [k], int64 [k], f32 [k, 13] (xyz | local xyz | six boundary offsets | in-margin flag), at most
  `max_all_pts` rows, at most `max_inbox_point` rows per RoI;
* "fake a non-empty input": when no point falls into any RoI the reference returns ONE row (-1, -1, zeros) instead of
  empty tensors (`:36-40`), and callers test `ext_pts_inds[0] == -1` (dynamic_point_roi_extractor.py:60);
* the three outputs are marked non-differentiable and the backward returns None for every input (`:44-53`).

Here the kernel is `fsf_dynamic_point_pool` (csrc/point_pool.hip, K17): rows come back in ascending (roi, point) order
(upstream: "automatically sorted, but not strictly guaranteed"), the caps are applied in that order, and the count
comes back with the call, so there is no [max_all_pts] mask compaction.
"""
import torch
from torch.autograd import Function

from ... import hip_ops


class DynamicPointPoolFunction(Function):
    @staticmethod
    def forward(ctx, rois, pts, extra_wlh, max_inbox_point, max_all_pts=50000):
        assert len(rois) > 0
        assert rois.dim() == 2 and rois.size(1) == 7, "rois: [N, 7] (x, y, z_bottom, w, l, h, rz)"
        out_pts_idx, out_roi_idx, out_pts_feats = hip_ops.dynamic_point_pool(
            rois.float(), pts.float(), extra_wlh, max_inbox_point, max_all_pts)
        if out_pts_idx.numel() == 0:  # fake a non-empty input
            out_pts_idx = out_pts_idx.new_full((1,), -1)
            out_roi_idx = out_roi_idx.new_full((1,), -1)
            out_pts_feats = out_pts_feats.new_zeros((1, 13))
        ctx.mark_non_differentiable(out_pts_idx)
        ctx.mark_non_differentiable(out_roi_idx)
        ctx.mark_non_differentiable(out_pts_feats)
        return out_pts_idx, out_roi_idx, out_pts_feats

    @staticmethod
    def backward(ctx, g1, g2, g3):
        return None, None, None, None, None


dynamic_point_pool = DynamicPointPoolFunction.apply
