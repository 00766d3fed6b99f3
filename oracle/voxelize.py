"""Oracle: the two float->voxel-index formulas on the path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for `dynamic_voxelize` (mmdet3d fork `Voxelization`, un-vendored): restated from the published
mmdet3d v0.15 `dynamic_voxelize_kernel`.  `divfloor_coors` is pinned by construction: it calls torch itself.
"""
import numpy as np
import torch


def grid_size(voxel_size, pc_range):
    """mmdet3d Voxelization.__init__: grid = round((max - min) / voxel)  (x, y, z)."""
    v = np.asarray(voxel_size, dtype=np.float32)
    r = np.asarray(pc_range, dtype=np.float32)
    g = torch.round(torch.from_numpy((r[3:] - r[:3]) / v)).long().tolist()
    return [int(x) for x in g]


def dynamic_voxelize(points, voxel_size, pc_range, grid=None):
    """Follows the reference call `self.voxel_layer(res)` at
    projects/mmdet3d_plugin/models/detectors/single_stage_fsd.py:218 (Voxelization(max_num_points=-1)).

    Upstream kernel, per point, all in fp32:  c = floor((p - min) / v); early-out x -> y -> z;
    x OOB writes -1 to slot 0; y OOB to slots 0,1; z OOB to slots 0,1,2; output zero-initialised int32 (z,y,x).
    """
    pts = np.asarray(points, dtype=np.float32)
    v = np.asarray(voxel_size, dtype=np.float32)
    mn = np.asarray(pc_range[:3], dtype=np.float32)
    if grid is None:
        grid = grid_size(voxel_size, pc_range)
    n = pts.shape[0]
    out = np.zeros((n, 3), dtype=np.int32)
    with np.errstate(invalid="ignore"):
        cx = np.floor((pts[:, 0] - mn[0]) / v[0]).astype(np.int64)
        cy = np.floor((pts[:, 1] - mn[1]) / v[1]).astype(np.int64)
        cz = np.floor((pts[:, 2] - mn[2]) / v[2]).astype(np.int64)
    bad_x = (cx < 0) | (cx >= grid[0])
    bad_y = ~bad_x & ((cy < 0) | (cy >= grid[1]))
    bad_z = ~bad_x & ~bad_y & ((cz < 0) | (cz >= grid[2]))
    ok = ~(bad_x | bad_y | bad_z)
    out[ok, 0] = cz[ok]
    out[ok, 1] = cy[ok]
    out[ok, 2] = cx[ok]
    out[bad_x, 0] = -1
    out[bad_y, 0] = -1
    out[bad_y, 1] = -1
    out[bad_z, :] = -1
    return out


def voxelize_batch(points_list, voxel_size, pc_range):
    """VoteSegmentor.voxelize + `.long()` (single_stage_fsd.py:206-226, :231): concat + batch-index pad."""
    coors = []
    for i, p in enumerate(points_list):
        c = dynamic_voxelize(p, voxel_size, pc_range)
        coors.append(np.concatenate([np.full((c.shape[0], 1), i, dtype=np.int64), c.astype(np.int64)], axis=1))
    return np.concatenate([np.asarray(p, dtype=np.float32) for p in points_list], 0), np.concatenate(coors, 0)


def divfloor_coors(points_xyz, voxel_size, range_min, order="zyx", batch_idx=None):
    """`torch.div(points[:, :3] - pc_range[None, :3], voxel_size[None, :], rounding_mode='floor').long()`
    exactly as written at single_stage_fsd.py:270, :591-593 (zyx + batch column) and :948-950 (xyz)."""
    p = torch.as_tensor(np.asarray(points_xyz, dtype=np.float32))[:, :3]
    v = torch.tensor(list(voxel_size), dtype=torch.float32)
    mn = torch.tensor(list(range_min), dtype=torch.float32)
    coors = torch.div(p - mn[None, :], v[None, :], rounding_mode="floor").long()
    if order == "zyx":
        coors = coors[:, [2, 1, 0]]
    if batch_idx is not None:
        coors = torch.cat([torch.as_tensor(batch_idx).long()[:, None], coors], dim=1)
    return coors.numpy()
