"""Input side of the hot path: the TEST-TIME data pipeline of the reference, from the on-disk formats to
the tensors `FSF.simple_test` takes —

    .bin float32 sweeps  -> LoadPointsFromFile / LoadPointsFromMultiSweeps   (mmdet3d stock, mirrored as
                            MyLoadPointsFromFile / MyLoadPointsFromMultiSweeps in
                            projects/mmdet3d_plugin/datasets/pipelines/loading.py:560-700, :702-877)
    + no-aug xyz copy    -> SaveNoAugPoints                                   (loading.py:341-354)
    PNG id planes + anno.json -> LoadMaskFromFiles                             (loading.py:22-339; written by
                            tools/mask_tools/save_mask_nusc.py:138-171)
    range filter, intensity / 255 -> PointsRangeFilter, NormalizePoints       (loading.py:537-563)
    bundle / collect     -> DefaultFormatBundle3D, Collect3D, MultiScaleFlipAug3D (single scale, no flip)

as configured by `test_pipeline` of projects/configs/_base_/datasets/nuscenes_dataloader.py:97-136.  Host side (numpy /
torch CPU, as upstream: these run in DataLoader workers); `frame_to_device` does the one host->device copy per tensor
and keeps the id planes in their stored integer type (u8 nuScenes / i32 Argoverse) — the reference's `.float()`
conversions of the 86 MB mask never happen.  Train-time augmentations (GT sampling, random flip / rotation / scaling)
stay placeholders: the dataset / training control plane is out of scope.
"""
import json
import os

import numpy as np
import torch

from ..registry import PIPELINES


class LiDARPoints:
    """The part of mmdet3d.core.points.LiDARPoints the FSF pipelines touch: a [N, C] float tensor with a few helpers."""

    def __init__(self, tensor, points_dim=None, attribute_dims=None):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(np.ascontiguousarray(tensor), dtype=torch.float32)
        self.tensor = tensor.reshape(-1, points_dim or tensor.shape[-1]).clone()
        self.points_dim = self.tensor.shape[1]
        self.attribute_dims = attribute_dims

    def new_point(self, data):
        return LiDARPoints(torch.as_tensor(np.ascontiguousarray(data), dtype=self.tensor.dtype), self.points_dim)

    @classmethod
    def cat(cls, points_list):
        return cls(torch.cat([p.tensor for p in points_list], 0), points_list[0].points_dim)

    def __getitem__(self, item):
        if isinstance(item, tuple):
            return LiDARPoints(self.tensor[item])
        return LiDARPoints(self.tensor[item], self.points_dim)

    def __len__(self):
        return self.tensor.shape[0]

    def in_range_3d(self, rng):
        t = self.tensor
        return ((t[:, 0] > rng[0]) & (t[:, 1] > rng[1]) & (t[:, 2] > rng[2]) & (t[:, 0] < rng[3]) & (t[:, 1] < rng[4]) &
                (t[:, 2] < rng[5]))


def _load_float32(path):
    if path.endswith(".npy"):
        return np.load(path)
    return np.fromfile(path, dtype=np.float32)


@PIPELINES.register_module(force=True)
class LoadPointsFromFile:
    def __init__(self, coord_type="LIDAR", load_dim=6, use_dim=[0, 1, 2], shift_height=False, use_color=False,
                 file_client_args=dict(backend="disk"), virtual_path=None):
        if isinstance(use_dim, int):
            use_dim = list(range(use_dim))
        assert max(use_dim) < load_dim, f"Expect all used dimensions < {load_dim}, got {use_dim}"
        assert coord_type in ["CAMERA", "LIDAR", "DEPTH"] and not shift_height and not use_color and virtual_path is None
        self.coord_type, self.load_dim, self.use_dim = coord_type, load_dim, use_dim

    def __call__(self, results):
        points = _load_float32(results["pts_filename"]).reshape(-1, self.load_dim)[:, self.use_dim]
        results["points"] = LiDARPoints(points, points_dim=points.shape[-1])
        return results


@PIPELINES.register_module(force=True)
class LoadPointsFromMultiSweeps:
    def __init__(self, sweeps_num=10, load_dim=5, use_dim=[0, 1, 2, 4], file_client_args=dict(backend="disk"),
                 pad_empty_sweeps=False, remove_close=False, test_mode=False, virtual_path=None):
        assert virtual_path is None
        self.load_dim, self.sweeps_num, self.use_dim = load_dim, sweeps_num, use_dim
        self.pad_empty_sweeps, self.remove_close, self.test_mode = pad_empty_sweeps, remove_close, test_mode

    @staticmethod
    def _remove_close(points, radius=1.0):
        arr = points if isinstance(points, np.ndarray) else points.tensor.numpy()
        not_close = np.logical_not(np.logical_and(np.abs(arr[:, 0]) < radius, np.abs(arr[:, 1]) < radius))
        return points[not_close]

    def __call__(self, results):
        points = results["points"]
        points.tensor[:, 4] = 0
        sweep_points_list = [points]
        ts = results["timestamp"]
        if self.pad_empty_sweeps and len(results["sweeps"]) == 0:
            for _ in range(self.sweeps_num):
                sweep_points_list.append(self._remove_close(points) if self.remove_close else points)
        else:
            if len(results["sweeps"]) <= self.sweeps_num:
                choices = np.arange(len(results["sweeps"]))
            elif self.test_mode:
                choices = np.arange(self.sweeps_num)
            else:
                choices = np.random.choice(len(results["sweeps"]), self.sweeps_num, replace=False)
            for idx in choices:
                sweep = results["sweeps"][idx]
                points_sweep = np.copy(_load_float32(sweep["data_path"])).reshape(-1, self.load_dim)
                if self.remove_close:
                    points_sweep = self._remove_close(points_sweep)
                sweep_ts = sweep["timestamp"] / 1e6
                points_sweep[:, :3] = points_sweep[:, :3] @ np.asarray(sweep["sensor2lidar_rotation"]).T
                points_sweep[:, :3] += np.asarray(sweep["sensor2lidar_translation"])
                points_sweep[:, 4] = ts - sweep_ts
                sweep_points_list.append(points.new_point(points_sweep))
        points = LiDARPoints.cat(sweep_points_list)
        results["points"] = points[:, self.use_dim]
        return results


PIPELINES.register_module("MyLoadPointsFromFile", force=True, module=LoadPointsFromFile)
PIPELINES.register_module("MyLoadPointsFromMultiSweeps", force=True, module=LoadPointsFromMultiSweeps)
MyLoadPointsFromFile, MyLoadPointsFromMultiSweeps = LoadPointsFromFile, LoadPointsFromMultiSweeps


@PIPELINES.register_module(force=True)
class SaveNoAugPoints:
    def __call__(self, results):
        points = results["points"].tensor.clone()
        results["points"].tensor = torch.cat([results["points"].tensor, points[:, :3]], -1)
        results["points"].points_dim = results["points"].tensor.shape[1]
        if "gt_bboxes_3d" in results:
            results["no_aug_gt_bboxes_3d"] = results["gt_bboxes_3d"].clone()
            results["no_aug_gt_labels_3d"] = torch.from_numpy(results["gt_labels_3d"])
        return results


def _read_id_plane(path):
    """cv2.imread(path, -1) of the reference: the stored integer plane, unchanged (8- or 16-bit PNG)."""
    from PIL import Image

    with Image.open(path) as im:
        arr = np.array(im)
    assert arr.ndim == 2, f"{path}: expected a single-channel id plane"
    return arr


def _resize_nearest(plane, shape):
    """torchvision resize(InterpolationMode.NEAREST) of an integer plane: src index = floor(dst index * in / out)."""
    h, w = plane.shape[-2:]
    oh, ow = shape
    ys = torch.floor(torch.arange(oh, dtype=torch.float32) * (h / oh)).long().clamp_(max=h - 1)
    xs = torch.floor(torch.arange(ow, dtype=torch.float32) * (w / ow)).long().clamp_(max=w - 1)
    return plane[..., ys[:, None], xs[None, :]]


@PIPELINES.register_module(force=True)
class LoadMaskFromFiles:
    """Per-camera instance-id planes + the 2-D detections they index (`anno.json`) -> `mask_data`, `mask_anno`."""

    def __init__(self, data_path, class_names=["car", "truck", "trailer", "bus", "construction_vehicle", "bicycle", "motorcycle",
                                               "pedestrian", "traffic_cone", "barrier"],
                 obj_max_num=250, is_argo=False, is_waymo=False):
        self.data_path, self.obj_max_num, self.class_names = data_path, obj_max_num, class_names
        self.is_argo, self.is_waymo = is_argo, is_waymo

    # -- annotations ---------------------------------------------------------------------------------
    def pad_tensor(self, data_list):
        data = torch.tensor(data_list)
        pad_shape = (self.obj_max_num - len(data_list),) + tuple(data.shape[1:])
        return torch.cat((data, data.new_zeros(pad_shape)), dim=0)

    def _finish(self, rows):
        anno = self.pad_tensor(rows) if len(rows) else torch.zeros((self.obj_max_num, 8))
        valid = torch.zeros((self.obj_max_num, 1), dtype=torch.bool)
        valid[:len(rows)] = True
        return torch.cat([anno, valid], dim=-1)  # bbox(4), score, category, cam_id, obj_id, valid

    @staticmethod
    def _row(obj):
        return list(obj["bbox"]) + [obj["score"], obj["category"], obj["cam_id"], obj["obj_id"]]

    def reorg_anno_single_cls(self, annos):
        return self._finish([self._row(obj) for cam in annos for obj in cam])

    def reorg_anno_multi_cls(self, annos):
        rows, ids = [], []
        for cam in annos:
            for _, objs in cam.items():
                for obj in objs:
                    ids.append(obj["obj_id"])
                    rows.append(self._row(obj))
        order = torch.sort(torch.tensor(ids))[1] if ids else []
        return self._finish([rows[int(i)] for i in order])

    # -- resizing (the ring-front Argoverse camera, the two rear Waymo cameras) -------------------------
    @staticmethod
    def _scale_cam(results, cam_id, wf, hf):
        lidar2img = results["lidar2img"][cam_id]
        lidar2img[0] *= wf
        lidar2img[1] *= hf
        results["lidar2img"][cam_id] = lidar2img

    @staticmethod
    def _scale_boxes(objs, wf, hf):
        for obj in objs:
            b = obj["bbox"]
            obj["bbox"] = [b[0] * wf, b[1] * hf, b[2] * wf, b[3] * hf]

    # -- datasets --------------------------------------------------------------------------------------
    def load_nusc(self, results):
        sample_dir = os.path.join(self.data_path, results["sample_idx"])
        planes = [torch.from_numpy(_read_id_plane(os.path.join(sample_dir, f"{cam}_{name}.png")))
                  for cam in range(6) for name in self.class_names]
        anno = json.load(open(os.path.join(sample_dir, "anno.json"), "r"))
        results["mask_anno"] = self.reorg_anno_multi_cls(anno)
        results["mask_data"] = torch.stack(planes, 0).reshape(6, len(self.class_names), *planes[0].shape)
        return results

    def load_argo(self, results):
        sample_dir = os.path.join(self.data_path, results["img_info"]["uuid"])
        planes = [torch.from_numpy(_read_id_plane(os.path.join(sample_dir, f"{i}.png")).astype(np.int32)) for i in range(7)]
        anno = json.load(open(os.path.join(sample_dir, "anno.json"), "r"))
        oh, ow = planes[0].shape
        hf, wf = 1550 / oh, 2048 / ow  # ring_front_center is stored portrait-size; bring it to the common plane size
        self._scale_cam(results, 0, wf, hf)
        planes[0] = _resize_nearest(planes[0], (1550, 2048))
        self._scale_boxes(anno[0], wf, hf)
        results["mask_anno"] = self.reorg_anno_single_cls(anno)
        results["mask_data"] = torch.stack(planes, 0).unsqueeze(1)
        return results

    def load_waymo(self, results):
        sample_dir = os.path.join(self.data_path, results["pts_filename"].split("/")[-1].replace(".bin", ""))
        names = ["vehicle", "pedestrian", "cyclist"]
        planes = [torch.from_numpy(_read_id_plane(os.path.join(sample_dir, f"{cam}_{n}.png"))) for cam in range(5) for n in names]
        anno = json.load(open(os.path.join(sample_dir, "anno.json"), "r"))
        oh, ow = planes[3 * len(names)].shape
        hf, wf = 1280 / oh, 1920 / ow
        for cam in (3, 4):
            self._scale_cam(results, cam, wf, hf)
            for _, objs in anno[cam].items():
                self._scale_boxes(objs, wf, hf)
        for i in range(3 * len(names), 5 * len(names)):
            planes[i] = _resize_nearest(planes[i], (1280, 1920))
        results["mask_anno"] = self.reorg_anno_multi_cls(anno)
        results["mask_data"] = torch.stack(planes, 0).reshape(5, len(names), *planes[0].shape)
        return results

    def __call__(self, results):
        if self.is_argo:
            return self.load_argo(results)
        if self.is_waymo:
            return self.load_waymo(results)
        return self.load_nusc(results)


@PIPELINES.register_module(force=True)
class PointsRangeFilter:
    def __init__(self, point_cloud_range):
        self.pcd_range = np.array(point_cloud_range, dtype=np.float32)

    def __call__(self, input_dict):
        points = input_dict["points"]
        input_dict["points"] = points[points.in_range_3d(self.pcd_range)]
        return input_dict


@PIPELINES.register_module(force=True)
class NormalizePoints:
    def __init__(self, std=[255], mean=[0], dims=[3]):
        self.dims, self.std, self.mean = dims, std, mean

    def __call__(self, input_dict):
        points = input_dict["points"]
        mean, std = torch.tensor(self.mean), torch.tensor(self.std)
        points.tensor[:, self.dims] = (points.tensor[:, self.dims] - mean[None, :]) / std[None, :]
        return input_dict


@PIPELINES.register_module(force=True)
class GlobalRotScaleTrans:
    """Test-time identity only (rot_range [0, 0], scale [1, 1], translation_std 0 — the values the test pipelines pass)."""

    def __init__(self, rot_range=[-0.78539816, 0.78539816], scale_ratio_range=[0.95, 1.05], translation_std=[0, 0, 0],
                 shift_height=False):
        ts = translation_std if isinstance(translation_std, (list, tuple)) else [translation_std] * 3
        self.identity = list(rot_range) == [0, 0] and list(scale_ratio_range) == [1.0, 1.0] and all(t == 0 for t in ts)

    def __call__(self, input_dict):
        if not self.identity:
            raise NotImplementedError("train-time GlobalRotScaleTrans augmentation is outside the built path")
        input_dict.update(pcd_rotation=torch.eye(3), pcd_scale_factor=1.0, pcd_trans=np.zeros(3, dtype=np.float32))
        return input_dict


@PIPELINES.register_module(force=True)
class RandomFlip3D:
    """Inside MultiScaleFlipAug3D(flip=False) the flip flags arrive preset to False: nothing to flip."""

    def __init__(self, sync_2d=True, flip_ratio_bev_horizontal=0.0, flip_ratio_bev_vertical=0.0, **kwargs):
        self.ratios = (flip_ratio_bev_horizontal, flip_ratio_bev_vertical)

    def __call__(self, input_dict):
        if input_dict.get("pcd_horizontal_flip", False) or input_dict.get("pcd_vertical_flip", False) or any(self.ratios):
            raise NotImplementedError("train-time RandomFlip3D augmentation is outside the built path")
        input_dict.setdefault("pcd_horizontal_flip", False)
        input_dict.setdefault("pcd_vertical_flip", False)
        return input_dict


@PIPELINES.register_module(force=True)
class DefaultFormatBundle3D:
    def __init__(self, class_names, with_gt=True, with_label=True):
        self.class_names, self.with_gt, self.with_label = class_names, with_gt, with_label

    def __call__(self, results):
        if "points" in results and isinstance(results["points"], LiDARPoints):
            results["points"] = results["points"].tensor
        return results


@PIPELINES.register_module(force=True)
class Collect3D:
    META = ("filename", "ori_shape", "img_shape", "lidar2img", "depth2img", "cam2img", "pad_shape", "scale_factor", "flip",
            "pcd_horizontal_flip", "pcd_vertical_flip", "box_mode_3d", "box_type_3d", "img_norm_cfg", "pcd_trans", "sample_idx",
            "pcd_scale_factor", "pcd_rotation", "pts_filename", "transformation_3d_flow")

    def __init__(self, keys, meta_keys=META):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = {"img_metas": {k: results[k] for k in self.meta_keys if k in results}}
        for key in self.keys:
            data[key] = results[key]
        return data


class Compose:
    def __init__(self, transforms):
        self.transforms = [PIPELINES.build(t) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


@PIPELINES.register_module(force=True)
class MultiScaleFlipAug3D:
    """mmdet3d test-time wrapper; the FSF configs use one scale and no flip, so it wraps every collected value in a
    one-element list (the `num_augs == 1` form `FSF.forward_test` expects, FSF.py:1096-1112)."""

    def __init__(self, transforms, img_scale, pts_scale_ratio, flip=False, flip_direction="horizontal",
                 pcd_horizontal_flip=False, pcd_vertical_flip=False):
        scales = pts_scale_ratio if isinstance(pts_scale_ratio, list) else [pts_scale_ratio]
        if flip or pcd_horizontal_flip or pcd_vertical_flip or len(scales) != 1 or float(scales[0]) != 1.0:
            raise NotImplementedError("test-time augmentation (multi-scale / flip) is outside the built path")
        self.transforms = Compose(transforms)

    def __call__(self, results):
        res = dict(results)
        res.update(scale=None, flip=False, pcd_scale_factor=1.0, pcd_horizontal_flip=False, pcd_vertical_flip=False)
        data = self.transforms(res)
        return {k: [v] for k, v in data.items()}


PIPELINES.register_module("MyMultiScaleFlipAug3D", force=True, module=MultiScaleFlipAug3D)
PIPELINES.register_module("MyPointsRangeFilter", force=True, module=PointsRangeFilter)


class DevicePointAssembler:
    """The point side of the test pipeline (configs/_base_/datasets/nuscenes_dataloader.py:96-137) with its per-point passes
    on the device: the raw key-frame and sweep files are read on the host (disk I/O is host work), go to the GPU in ONE
    host->device copy, and `fsf_assemble_sweeps` (K0) does what LoadPointsFromMultiSweeps
    (datasets/pipelines/loading.py:825-877), SaveNoAugPoints (:341-354), PointsRangeFilter and NormalizePoints (:537-563) do
    in numpy / torch on the host — same rows, same order, bit-identical values (tests pin it to input_pipeline.npz through
    the host classes above).  Returns f32 [N, load_dim + 3] on `device`: what `FSF.simple_test` takes as `points[0]`."""

    def __init__(self, load_dim=5, sweeps_num=9, pad_empty_sweeps=True, remove_close=True, test_mode=False,
                 point_cloud_range=None, norm_dims=(3,), norm_mean=(0,), norm_std=(255,), close_radius=1.0):
        assert len(norm_dims) <= 1, "the kernel normalises one column (the reference configs use dims=[3])"
        self.load_dim, self.sweeps_num = load_dim, sweeps_num
        self.pad_empty_sweeps, self.remove_close, self.test_mode = pad_empty_sweeps, remove_close, test_mode
        self.point_cloud_range = None if point_cloud_range is None else [float(v) for v in point_cloud_range]
        self.norm = (int(norm_dims[0]), float(norm_mean[0]), float(norm_std[0])) if len(norm_dims) else (-1, 0.0, 1.0)
        self.close_radius = close_radius

    def __call__(self, results, device):
        from ... import hip_ops

        key = np.copy(_load_float32(results["pts_filename"])).reshape(-1, self.load_dim)
        ts = results["timestamp"]
        identity = [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0, 0, 0, 0, 0.0]
        chunks, params, transform, close = [key], [identity], [False], [False]
        sweeps = results.get("sweeps", [])
        if self.pad_empty_sweeps and len(sweeps) == 0:  # the key frame repeated, close points removed from the copies
            for _ in range(self.sweeps_num):
                chunks.append(key)
                params.append(identity)
                transform.append(False)
                close.append(self.remove_close)
        else:
            if len(sweeps) <= self.sweeps_num:
                choices = np.arange(len(sweeps))
            elif self.test_mode:
                choices = np.arange(self.sweeps_num)
            else:
                choices = np.random.choice(len(sweeps), self.sweeps_num, replace=False)
            for idx in choices:
                sw = sweeps[idx]
                chunks.append(np.copy(_load_float32(sw["data_path"])).reshape(-1, self.load_dim))
                rot = np.asarray(sw["sensor2lidar_rotation"], dtype=np.float64).reshape(9)
                tr = np.asarray(sw["sensor2lidar_translation"], dtype=np.float64).reshape(3)
                params.append([*rot, *tr, float(np.float32(ts - sw["timestamp"] / 1e6))])  # (the column it lands in is fp32)
                transform.append(True)
                close.append(self.remove_close)
        offsets = np.concatenate([[0], np.cumsum([c.shape[0] for c in chunks])]).tolist()
        raw = torch.from_numpy(np.concatenate(chunks, 0))
        if torch.cuda.is_available():
            raw = raw.pin_memory()
        col, mean, std = self.norm
        return hip_ops.assemble_sweeps(raw.to(device, non_blocking=True), offsets, params, transform, close, self.close_radius,
                                       self.point_cloud_range, col, mean, std)


def frame_to_device(data, device):
    """Pipeline output (one sample, `num_augs == 1`) -> the argument tuple of `FSF.simple_test`: ONE host->device copy per
    tensor; the id planes keep their stored integer type."""
    unwrap = lambda v: v[0] if isinstance(v, list) else v  # noqa: E731
    points = unwrap(data["points"])
    mask = unwrap(data["mask_data"])
    anno = unwrap(data["mask_anno"])
    meta = dict(unwrap(data["img_metas"]))
    if mask.dtype not in (torch.uint8, torch.int32):
        mask = mask.to(torch.int32)
    meta["lidar2img"] = torch.as_tensor(np.asarray(meta["lidar2img"]), dtype=torch.float32).to(device)
    return ([points.float().contiguous().to(device)], [meta], mask.to(device)[None], anno.float().to(device)[None])
