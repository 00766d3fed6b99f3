"""`Voxelization` — mirror of the mmdet3d-fork op the reference imports at
projects/mmdet3d_plugin/models/detectors/single_stage_fsd.py:13 and builds at :176 [UNVENDORED].
Only dynamic voxelization (max_num_points == -1) is on the FSF path; hard voxelization raises."""
import torch
import torch.nn as nn

from ... import hip_ops


class Voxelization(nn.Module):
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        super().__init__()
        self.voxel_size = list(voxel_size)
        self.point_cloud_range = list(point_cloud_range)
        self.max_num_points = max_num_points
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else (max_voxels, max_voxels)
        pcr = torch.tensor(point_cloud_range, dtype=torch.float32)
        vs = torch.tensor(voxel_size, dtype=torch.float32)
        grid = torch.round((pcr[3:] - pcr[:3]) / vs).long()
        self.grid_size = grid  # (x, y, z)
        self.pcd_shape = [*grid.tolist()[:2], 1][::-1]

    def forward(self, points):
        """points f32 [n, C>=3] -> coors i32 [n, 3] in (z, y, x) order, -1 marks out-of-range."""
        if self.max_num_points != -1 and self.max_voxels[0] != -1:
            raise NotImplementedError("hard voxelization is not on the FSF path; use max_num_points=-1")
        zyx, _ = hip_ops.voxelize_dynamic(points, self.voxel_size, self.point_cloud_range, self.grid_size.tolist())
        return zyx

    def forward_batch(self, points_list):
        """Fused VoteSegmentor.voxelize (:206-226) + `.long()` (:231): per-sample voxelize with the batch index
        written by the same kernel -> (points f32 [N,C], coors i64 [N,4] (b,z,y,x))."""
        outs = []
        for i, p in enumerate(points_list):
            _, bzyx = hip_ops.voxelize_dynamic(p, self.voxel_size, self.point_cloud_range, self.grid_size.tolist(),
                                               batch_idx=i, want_zyx=False, want_bzyx=True)
            outs.append(bzyx)
        return (torch.cat(points_list, 0) if len(points_list) > 1 else points_list[0]), \
               (torch.cat(outs, 0) if len(outs) > 1 else outs[0])

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range={self.point_cloud_range}, "
                f"max_num_points={self.max_num_points}, max_voxels={self.max_voxels})")
