"""Oracle: LiDAR->camera projection + per-point mask gather + camera select.
TEST INFRASTRUCTURE ONLY.

Restates FSF.prj_points_2d (projects/mmdet3d_plugin/models/detectors/FSF.py:169-200),
FSF.points_in_mask (:202-226), the cam-select of FSF.img_cross_attn (:716-718),
FSF.get_all_cls_preds_2d (:506-535) and the nuScenes branch of encode_preds_2d (:449-474) in numpy fp32.
Pinned: tests/golden/project_*.npz come from the reference's own methods lifted with `ast` and run on CPU torch.

The reference's `pts_4d @ lidar2img.permute(0,2,1)` is an fp32 GEMM; the in-container CPU result is
bit-identical to an fma chain in k order, acc = x*m0; acc = fma(y,m1,acc); acc = fma(z,m2,acc); acc = fma(1,m3,acc)
(probed: 0 mismatches in 3.6e6 values; plain mul+add differs on 23 %).  fma is emulated through float64: the
product of two fp32 is exact in fp64 and the one fp64 rounding before the final fp32 rounding can only matter
on an exact fp32 tie of the fp64-rounded sum.
"""
import numpy as np


def _fma32(a, b, c):
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32)


def prj_points_2d(points, lidar2img, img_h, img_w):
    """points f32 [n,3], lidar2img f32 [ncam,4,4] -> pts_2d f32 [ncam,n,2] in (-1,1), -2 where invalid."""
    p = np.asarray(points, dtype=np.float32)
    L = np.asarray(lidar2img, dtype=np.float32)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    one = np.ones_like(x)
    ncam = L.shape[0]
    out = np.empty((ncam, p.shape[0], 2), dtype=np.float32)
    for c in range(ncam):
        rows = []
        for j in range(3):
            m = L[c, j]
            acc = x * m[0]                      # FSF.py:179 (the matmul), k = 0
            acc = _fma32(y, m[1], acc)
            acc = _fma32(z, m[2], acc)
            acc = _fma32(one, m[3], acc)
            rows.append(acc)
        u, v, d = rows
        depth_valid = d > np.float32(1e-3)      # :180
        d = np.clip(d, np.float32(1e-5), np.float32(1e5))  # :182
        u = u / d                               # :183
        v = v / d                               # :184
        u = u / np.float32(img_w)               # :186
        v = v / np.float32(img_h)               # :187
        gu = (u - np.float32(0.5)) * np.float32(2)  # :190
        gv = (v - np.float32(0.5)) * np.float32(2)
        img_valid = (gu > -1.0) & (gu < 1.0) & (gv > -1.0) & (gv < 1.0)  # :192-195
        valid = depth_valid & img_valid
        gu = np.where(valid, gu, np.float32(-2.0))  # :199
        gv = np.where(valid, gv, np.float32(-2.0))
        out[c, :, 0] = gu
        out[c, :, 1] = gv
    return out


def grid_sample_nearest_ids(mask_cam, grid):
    """F.grid_sample(mask.float(), grid, mode='nearest') with align_corners=False, padding_mode='zeros'
    (FSF.py:221) followed by .long(), for one camera: mask_cam int [ncls,H,W], grid f32 [n,2] -> ids [ncls,n].
    ATen: ix = (g + 1) * (W / 2) - 0.5; nearest = nearbyint (round-half-even); outside -> 0."""
    ncls, H, W = mask_cam.shape
    gx = grid[:, 0].astype(np.float32)
    gy = grid[:, 1].astype(np.float32)
    ix = (gx + np.float32(1)) * (np.float32(W) / np.float32(2)) - np.float32(0.5)
    iy = (gy + np.float32(1)) * (np.float32(H) / np.float32(2)) - np.float32(0.5)
    rx = np.rint(ix)
    ry = np.rint(iy)
    inb = (rx >= 0) & (rx < W) & (ry >= 0) & (ry < H)
    xi = np.where(inb, rx, 0).astype(np.int64)
    yi = np.where(inb, ry, 0).astype(np.int64)
    ids = mask_cam[:, yi, xi].astype(np.int64)
    ids[:, ~inb] = 0
    return ids


def points_in_mask(points, mask_data, lidar2img):
    """FSF.points_in_mask (FSF.py:202-226): mask_data int [ncam,ncls,H,W] -> obj_id i64 [n,ncam,ncls]."""
    mask_data = np.asarray(mask_data)
    ncam, ncls, H, W = mask_data.shape
    pts_2d = prj_points_2d(points, lidar2img, H, W)
    out = np.empty((np.asarray(points).shape[0], ncam, ncls), dtype=np.int64)
    for c in range(ncam):
        out[:, c, :] = grid_sample_nearest_ids(mask_data[c], pts_2d[c]).T
    return out, pts_2d


def cam_select_score(obj_id, mask_anno, score_col=4):
    """FSF.img_cross_attn cam-select (FSF.py:716-718) + get_all_cls_preds_2d (:506-535) + the nuScenes
    `en_feat = en_score` branch of encode_preds_2d (:472-473), one batch sample.
    obj_id i64 [n,ncam,ncls], mask_anno f32 [A,9] -> (ids i64 [n,ncls], score f32 [n,ncls])."""
    obj_id = np.asarray(obj_id)
    anno = np.asarray(mask_anno, dtype=np.float32)
    cam = obj_id.sum(-1).argmax(-1)  # first maximum, like torch.max(dim)[1] on CPU
    ids = obj_id[np.arange(obj_id.shape[0]), cam, :]
    score = np.zeros(ids.shape, dtype=np.float32)
    valid = ids > 0
    score[valid] = anno[ids[valid] - 1, score_col]
    return ids, score


def gather_bilinear(points, lidar2img, feat, img_h, img_w):
    """North-star "per-point bilinear image-feature gather": FSF.prj_points_2d (FSF.py:169-200, restated above) followed by
    what FSF.points_in_mask does with the id planes (:216-225) but on a float feature map and in bilinear mode:
    torch.nn.functional.grid_sample(feat[cam][None], grid[None, None], mode='bilinear', align_corners=False,
    padding_mode='zeros') — torch's own fp32 kernel is the reference.  feat f32 [ncam, C, Hf, Wf] -> [n, ncam, C]; invalid
    projections carry the grid value -2, which samples zero padding."""
    import torch
    import torch.nn.functional as F

    pts_2d = prj_points_2d(points, lidar2img, img_h, img_w)  # [ncam, n, 2]
    feat = torch.as_tensor(np.asarray(feat), dtype=torch.float32)
    out = []
    for c in range(feat.shape[0]):
        grid = torch.from_numpy(pts_2d[c])[None, None]       # [1, 1, n, 2]
        out.append(F.grid_sample(feat[c][None], grid, mode="bilinear", align_corners=False, padding_mode="zeros")[0, :, 0].T)
    valid = (pts_2d[:, :, 0] > -1.5).T                        # [n, ncam]
    return torch.stack(out, 1).numpy(), valid
