"""Seeded synthetic nuScenes-shape inputs (SURVEY.md §8 d): there is no dataset or checkpoint on the box, so the
bench and the full-size parity tests run on these.  Pure numpy, CPU side; the arrays have exactly the layout the
reference's pipeline hands to the model:

  points     f32 [N, 8]   (x, y, z, intensity, dt, x0, y0, z0)  — `SaveNoAugPoints` appends the un-augmented xyz
                          (projects/mmdet3d_plugin/datasets/pipelines/loading.py:347-354)
  mask_data  u8  [6, 10, 900, 1600] per-class instance-id planes (`LoadMaskFromFiles`, loading.py:213-234)
  mask_anno  f32 [250, 9] (x1, y1, x2, y2, score, category, cam_id, obj_id, valid)  (loading.py:301-339)
  lidar2img  f32 [6, 4, 4]
"""
import math

import numpy as np

POINT_RANGE = [-50.0, -50.0, -4.99, 50.0, 50.0, 2.99]  # PointsRangeFilter, _base_/datasets/nuscenes_dataloader.py:15
PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]        # FSF_nuScenes_config.py:11
SEG_VOXEL = (0.2, 0.2, 0.2)                             # :10
SPARSE_SHAPE = [40, 512, 512]                           # :12


def lidar_sweep(rng, beams=32, azimuth_steps=1084, sensor_height=1.84, sectors=64, noise=0.02):
    """One ray-cast sweep: ground plane + one wall per azimuth sector, sigma = 2 cm range noise."""
    elev = np.deg2rad(np.linspace(-30.67, 10.67, beams))
    az = np.linspace(-math.pi, math.pi, azimuth_steps, endpoint=False)
    wall_r = rng.uniform(8.0, 48.0, sectors)
    e, a = np.meshgrid(elev, az, indexing="ij")
    sector = ((a + math.pi) / (2 * math.pi) * sectors).astype(np.int64) % sectors
    r_wall = wall_r[sector] / np.maximum(np.cos(e), 1e-3)
    with np.errstate(divide="ignore"):
        r_ground = np.where(e < 0, sensor_height / np.maximum(-np.sin(e), 1e-6), np.inf)
    r = np.minimum(r_wall, r_ground) + rng.normal(0.0, noise, e.shape)
    x = r * np.cos(e) * np.cos(a)
    y = r * np.cos(e) * np.sin(a)
    z = r * np.sin(e)
    intensity = rng.uniform(0.0, 1.0, e.shape)
    return np.stack([x.ravel(), y.ravel(), z.ravel(), intensity.ravel()], 1).astype(np.float32)


def range_filter(p, pr=POINT_RANGE):
    keep = (p[:, 0] > pr[0]) & (p[:, 1] > pr[1]) & (p[:, 2] > pr[2]) & (p[:, 0] < pr[3]) & (p[:, 1] < pr[4]) & (p[:, 2] < pr[5])
    return p[keep]


def remove_close(p, radius=1.0):
    """`_remove_close` (loading.py:803-823): drop points with |x| < r and |y| < r."""
    keep = ~((np.abs(p[:, 0]) < radius) & (np.abs(p[:, 1]) < radius))
    return p[keep]


def make_points(num_sweeps=10, seed=0):
    """points f32 [N, 8]; N ~ 3.1e4 for one sweep, ~3.07e5 for ten."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(num_sweeps):
        s = remove_close(lidar_sweep(rng))
        s[:, 0] -= 0.5 * k  # ego motion between sweeps
        dt = np.full((s.shape[0], 1), 0.05 * k, dtype=np.float32)
        out.append(np.concatenate([s, dt], 1))
    p = range_filter(np.concatenate(out, 0)).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([p, p[:, :3]], 1))


def make_lidar2img(ncam=6, fx=1266.0, cx=800.0, cy=450.0):
    """Pinhole cameras at 360/ncam degree yaw steps, nuScenes-like mounting offsets."""
    mats = []
    for c in range(ncam):
        yaw = c * 2 * math.pi / ncam
        R = np.array([[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]])
        cam_from_lidar = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], dtype=np.float64) @ R.T
        T = np.eye(4)
        T[:3, :3] = cam_from_lidar
        T[:3, 3] = [0.02 * c, -0.35, -0.6]
        K = np.eye(4)
        K[0, 0] = K[1, 1] = fx
        K[0, 2], K[1, 2] = cx, cy
        mats.append((K @ T).astype(np.float32))
    return np.ascontiguousarray(np.stack(mats))


def make_mask_data(rng, ncam=6, ncls=10, H=900, W=1600, num_inst=250, dtype=np.uint8, painted=None, max_area=0.05):
    """Instance-id planes painted as axis-aligned rectangles (0.1-5 % of the image) + the matching anno rows
    sorted by obj_id.  `painted` < num_inst: only that many instances exist, the remaining anno rows are the zero padding
    (valid = 0) the reference's loader appends (datasets/pipelines/loading.py:301-339)."""
    mask = np.zeros((ncam, ncls, H, W), dtype=dtype)
    anno = np.zeros((num_inst, 9), dtype=np.float32)
    for inst in range(1, (num_inst if painted is None else min(painted, num_inst)) + 1):
        cam = int(rng.integers(ncam))
        cls = int(rng.integers(ncls))
        area = rng.uniform(0.001, max_area) * H * W
        aspect = rng.uniform(0.4, 2.5)
        w = int(min(W - 1, max(2, math.sqrt(area * aspect))))
        h = int(min(H - 1, max(2, area / max(w, 1))))
        x1 = int(rng.integers(0, W - w))
        y1 = int(rng.integers(0, H - h))
        mask[cam, cls, y1:y1 + h, x1:x1 + w] = inst
        anno[inst - 1] = [x1, y1, x1 + w, y1 + h, rng.uniform(0.1, 1.0), cls, cam, inst, 1.0]
    return mask, anno


def make_frame(num_sweeps=10, seed=0, mask_instances=None, mask_max_area=0.05):
    """One synthetic frame of BASELINE.json config 3 (10 sweeps) or config 2 (1 sweep).  `mask_instances` / `mask_max_area`: fewer,
    smaller 2-D instances (bench.py's trained-like variant: a real frame has a few dozen masks covering a few % of the pixels, not
    250 covering most of them)."""
    rng = np.random.default_rng(seed + 1000)
    pts = make_points(num_sweeps, seed)
    mask, anno = make_mask_data(rng, painted=mask_instances, max_area=mask_max_area)
    return dict(points=pts, mask_data=mask, mask_anno=anno, lidar2img=make_lidar2img())


def make_frame_av2(seed=0, n_points=150000):
    """One synthetic frame of BASELINE.json config 5 (Argoverse 2 shape): ~1.5e5 4-d points out to 200 m (two 32-beam
    rings, dense near the sensor like a real long-range sweep), 7 ring cameras with ONE int32 instance-id plane each
    (1550 x 2048, ids beyond 255), 26 classes; points f32 [N, 7] = x y z intensity | no-aug xyz."""
    rng = np.random.default_rng(seed + 5000)
    r = np.minimum(rng.exponential(35.0, n_points) + 2.0, 200.0)
    a = rng.uniform(-math.pi, math.pi, n_points)
    z = np.clip(rng.normal(-1.4, 0.35, n_points) + 0.004 * r * rng.normal(0, 1, n_points), -3.1, 3.1)
    wall = rng.random(n_points) < 0.25                      # a quarter of the returns come from vertical structure
    z = np.where(wall, rng.uniform(-1.5, 3.0, n_points), z)
    xyz = np.stack([r * np.cos(a), r * np.sin(a), z], 1)
    xyz = xyz[(np.abs(xyz[:, 0]) < 204.7) & (np.abs(xyz[:, 1]) < 204.7)]
    pts = np.concatenate([xyz, rng.random((xyz.shape[0], 1)), xyz], 1).astype(np.float32)
    mask, anno = make_mask_data(rng, 7, 1, 1550, 2048, 400, dtype=np.int32)
    anno[:, 5] = rng.integers(0, 26, anno.shape[0])
    return dict(points=np.ascontiguousarray(pts), mask_data=mask, mask_anno=anno,
                lidar2img=make_lidar2img(7, fx=1780.0, cx=1024.0, cy=775.0))
