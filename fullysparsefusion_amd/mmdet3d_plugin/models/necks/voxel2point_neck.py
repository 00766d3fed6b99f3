"""`Voxel2PointScatterNeck` (NECKS): mirror of projects/mmdet3d_plugin/models/necks/voxel2point_neck.py:8-70.
The row gather, the "all == padding" scan and the local-xyz decoration are one fused HIP kernel; the boolean
compaction stays a torch op (it is the reference's return contract: compacted rows + mask)."""
import torch
from torch import nn

from .... import hip_ops
from ...ops.sst_ops import gather_by_inverse
from ...registry import NECKS


@NECKS.register_module()
class Voxel2PointScatterNeck(nn.Module):
    def __init__(self, point_cloud_range=None, voxel_size=None, with_xyz=True, normalize_local_xyz=False):
        super().__init__()
        self.point_cloud_range = point_cloud_range
        self.voxel_size = voxel_size
        self.with_xyz = with_xyz
        self.normalize_local_xyz = normalize_local_xyz

    takes_row_map = True  # (forward accepts `row_map`: voxel row r of the caller's order is row row_map[r] of `voxel_feats`)

    def forward(self, points, pts_coors, voxel_feats, voxel2point_inds, voxel_padding=-1, row_map=None):
        assert points.size(0) == pts_coors.size(0) == voxel2point_inds.size(-1)
        if row_map is not None:  # one small index gather (a point's voxel row in the given order) instead of a [voxels, C] copy
            nonneg = getattr(voxel_feats, "_fsf_nonnegative", False)
            voxel2point_inds = row_map.index_select(0, voxel2point_inds.long())
            voxel_feats._fsf_nonnegative = nonneg
        fused_ok = self.with_xyz and not self.normalize_local_xyz and not (torch.is_grad_enabled() and voxel_feats.requires_grad)
        if fused_ok:
            out, pts_mask = hip_ops.voxel2point(points, pts_coors, voxel_feats, voxel2point_inds, self.voxel_size,
                                                self.point_cloud_range[:3], float(voxel_padding))
            if self.training:
                vs = torch.tensor(self.voxel_size, dtype=out.dtype, device=out.device).reshape(1, 3)
                assert (out[pts_mask][:, -3:].abs() < vs / 2 + 1e-3).all(), \
                    "Holds in training. However, in test, this is not always True because of lack of point range clip"
            # no padded voxel row (the usual case; known without looking when the features are >= 0 and the padding value is not)
            if not self.training and ((voxel_padding < 0 and getattr(voxel_feats, "_fsf_nonnegative", False)) or bool(pts_mask.all())):
                pts_mask.fsf_all_true = True  # (the detector asks the same question: spare it the second round trip)
                return out, pts_mask
            return out[pts_mask], pts_mask
        dtype, device = voxel_feats.dtype, voxel_feats.device
        pts_feats = gather_by_inverse(voxel_feats, voxel2point_inds)
        pts_mask = ~((pts_feats == voxel_padding).all(1))
        if not self.with_xyz:
            return pts_feats[pts_mask], pts_mask
        pts_feats, pts_coors, points = pts_feats[pts_mask], pts_coors[pts_mask], points[pts_mask]
        voxel_size = torch.tensor(self.voxel_size, dtype=dtype, device=device).reshape(1, 3)
        pc_min_range = torch.tensor(self.point_cloud_range[:3], dtype=dtype, device=device).reshape(1, 3)
        centers = (pts_coors[:, [3, 2, 1]].to(dtype) + 0.5) * voxel_size + pc_min_range
        local_xyz = points[:, :3] - centers
        if self.normalize_local_xyz:
            local_xyz = local_xyz / (voxel_size / 2)
        if self.training and not self.normalize_local_xyz:
            assert (local_xyz.abs() < voxel_size / 2 + 1e-3).all()
        return torch.cat([pts_feats, local_xyz], 1), pts_mask
