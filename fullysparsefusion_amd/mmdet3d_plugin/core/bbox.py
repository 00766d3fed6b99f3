"""Box-side pieces the heads need at inference:

* `BasePointBBoxCoder` — projects/mmdet3d_plugin/core/bbox/coders/base_point_bbox_coder.py:8-82.
* `LiDARInstance3DBoxes`, `xywhr2xyxyr`, `box3d_multiclass_nms`, `nms_gpu`, `nms_normal_gpu`, `bbox3d2result` — the
  handful of mmdet3d 0.x [UNVENDORED] symbols imported at
  projects/mmdet3d_plugin/models/dense_heads/frustum_cluster_head.py:9 and detectors/FSF.py (bbox3d2result), restated
  from their published behaviour.  The NMS itself is the HIP kernel pair behind `fsf_nms_bev` (K20): the greedy scan
  stays on the device instead of mmdet3d's bitmask-to-host round trip.
"""
import torch

from ... import hip_ops
from ..registry import BBOX_CODERS


@BBOX_CODERS.register_module(force=True)
class BasePointBBoxCoder:
    """reg = (center - base_point, log(dims), sin(yaw), cos(yaw)[, vx, vy])."""

    def __init__(self, post_center_range=None, score_thresh=0.1, num_classes=10, max_num=500, code_size=10):
        self.post_center_range = post_center_range
        self.code_size = code_size
        self.EPS = 1e-6
        self.score_thresh = score_thresh
        self.num_classes = num_classes
        self.max_num = max_num

    def encode(self, bboxes, base_points):
        assert bboxes.size(1) in (7, 9, 10), f"bboxes shape: {bboxes.shape}"
        assert bboxes.size(0) == base_points.size(0)
        yaw = bboxes[:, 6:7]
        target = torch.cat([bboxes[:, :3] - base_points, (bboxes[:, 3:6] + self.EPS).log(), yaw.sin(), yaw.cos()], dim=1)
        if bboxes.size(1) in (9, 10):  # velocity (or copy-paste flag) rides along
            assert self.code_size == 10
            target = torch.cat([target, bboxes[:, [7, 8]]], dim=1)
        return target

    def decode(self, reg_preds, base_points, detach_yaw=False):
        assert reg_preds.size(1) in (8, 10) and reg_preds.size(1) == self.code_size
        velo = reg_preds[:, -2:] if self.code_size == 10 else None
        reg = reg_preds[:, :8]
        dims = reg[:, 3:6].exp() - self.EPS
        xyz = reg[:, :3] + base_points
        yaw = torch.atan2(reg[:, 6:7], reg[:, 7:8])
        if detach_yaw:
            yaw = yaw.clone().detach()
        parts = [xyz, dims, yaw] + ([velo] if velo is not None else [])
        return torch.cat(parts, dim=1)


class LiDARInstance3DBoxes:
    """mmdet3d 0.x LiDAR boxes, as far as the FSF heads touch them: rows (x, y, z_bottom, w, l, h, yaw[, extras])."""

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, box_dim)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == box_dim, tensor.size()
        if tensor.shape[-1] == 6:
            assert box_dim == 6
            tensor = torch.cat((tensor, tensor.new_zeros(tensor.shape[0], 1)), dim=-1)
            box_dim, with_yaw = 7, False
        self.box_dim, self.with_yaw = box_dim, with_yaw
        self.tensor = tensor.clone()
        if tuple(origin) != (0.5, 0.5, 0):
            self.tensor[:, :3] += self.tensor[:, 3:6] * (self.tensor.new_tensor((0.5, 0.5, 0)) - self.tensor.new_tensor(origin))

    @classmethod
    def _wrap(cls, tensor, box_dim, with_yaw=True):
        """The boxes AS `tensor` (no copy, no origin shift): for rows that were just produced for this object alone."""
        self = cls.__new__(cls)
        self.box_dim, self.with_yaw, self.tensor = box_dim, with_yaw, tensor
        return self

    @property
    def bev(self):
        return self.tensor[:, [0, 1, 3, 4, 6]]

    @property
    def gravity_center(self):
        c = self.tensor[:, :3].clone()
        c[:, 2] = c[:, 2] + self.tensor[:, 5] * 0.5
        return c

    @property
    def dims(self):
        return self.tensor[:, 3:6]

    @property
    def yaw(self):
        return self.tensor[:, 6]

    @property
    def device(self):
        return self.tensor.device

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        if isinstance(item, int):
            return type(self)(self.tensor[item].view(1, -1), box_dim=self.box_dim, with_yaw=self.with_yaw)
        return type(self)(self.tensor[item], box_dim=self.box_dim, with_yaw=self.with_yaw)

    def to(self, device):
        return type(self)(self.tensor.to(device), box_dim=self.box_dim, with_yaw=self.with_yaw)

    def clone(self):
        return type(self)(self.tensor.clone(), box_dim=self.box_dim, with_yaw=self.with_yaw)

    @classmethod
    def cat(cls, boxes_list):
        assert isinstance(boxes_list, (list, tuple))
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        assert all(isinstance(b, cls) for b in boxes_list)
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0), box_dim=boxes_list[0].tensor.shape[1],
                   with_yaw=boxes_list[0].with_yaw)

    def __repr__(self):
        return self.__class__.__name__ + "(\n    " + str(self.tensor) + ")"


def xywhr2xyxyr(boxes_xywhr):
    """(x, y, w, h, r) -> (x - w/2, y - h/2, x + w/2, y + h/2, r)."""
    boxes = torch.zeros_like(boxes_xywhr)
    half_w, half_h = boxes_xywhr[:, 2] / 2, boxes_xywhr[:, 3] / 2
    boxes[:, 0] = boxes_xywhr[:, 0] - half_w
    boxes[:, 1] = boxes_xywhr[:, 1] - half_h
    boxes[:, 2] = boxes_xywhr[:, 0] + half_w
    boxes[:, 3] = boxes_xywhr[:, 1] + half_h
    boxes[:, 4] = boxes_xywhr[:, 4]
    return boxes


def _nms(boxes, scores, thresh, rotated, pre_maxsize=None, post_max_size=None):
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep = hip_ops.nms_bev(boxes[order].float().contiguous(), thresh, rotated=rotated)
    keep = order[keep].contiguous()
    if post_max_size is not None:
        keep = keep[:post_max_size]
    return keep


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """mmdet3d.ops.iou3d.nms_gpu: rotated BEV NMS; boxes (x1, y1, x2, y2, ry); returns kept indices into `boxes`."""
    return _nms(boxes, scores, thresh, True, pre_maxsize, post_max_size)


def nms_normal_gpu(boxes, scores, thresh):
    """mmdet3d.ops.iou3d.nms_normal_gpu: the same with the yaw ignored."""
    return _nms(boxes, scores, thresh, False)


def box3d_multiclass_nms(mlvl_bboxes, mlvl_bboxes_for_nms, mlvl_scores, score_thr, max_num, cfg):
    """mmdet3d.core.post_processing.box3d_multiclass_nms (the arguments the FSF heads pass): per class, threshold ->
    BEV NMS -> concatenate class by class; over max_num keep the best scores.  The last score column is the padded
    background.  Upstream runs one nms_gpu per class (each with its own host round trip); here every class goes through
    ONE `fsf_nms_bev_multiclass` call and there is a single device->host read (how many boxes survived)."""
    num_classes = mlvl_scores.shape[1] - 1
    n = mlvl_scores.shape[0]
    if n == 0 or num_classes == 0:
        return (mlvl_scores.new_zeros((0, mlvl_bboxes.size(-1))), mlvl_scores.new_zeros((0,)),
                mlvl_scores.new_zeros((0,), dtype=torch.long))
    st = mlvl_scores[:, :num_classes].t().contiguous()                      # [C, n]
    valid = st > score_thr
    order = torch.where(valid, st, st.new_full((), float("-inf"))).sort(dim=1, descending=True, stable=True)[1]
    count = valid.sum(1, dtype=torch.int32)
    pos = torch.arange(n, device=st.device, dtype=torch.int32).expand(num_classes, n)
    rank = torch.empty_like(pos).scatter_(1, order, pos)
    rank = torch.where(valid, rank, rank.new_full((), -1))
    cap = int(max_num) if max_num is not None and max_num > 0 else 0  # (only the best max_num survive below)
    boxes_nms = mlvl_bboxes_for_nms.float()
    rotated = bool(cfg.get("use_rotate_nms", False))
    if cap > 0:
        # per-class masks over each class's best-scoring window only; a class that runs out of window before `cap` keeps
        # (checked after the host sync the compaction below needs anyway) repeats the call on full masks
        keep, num, incomplete = hip_ops.nms_bev_multiclass(boxes_nms, rank, count, cfg["nms_thr"], rotated=rotated, max_keep=cap,
                                                           windowed=True)
    else:
        keep, num = hip_ops.nms_bev_multiclass(boxes_nms, rank, count, cfg["nms_thr"], rotated=rotated)
        incomplete = None
    kept = (torch.arange(n, device=st.device)[None, :] < num[:, None]).nonzero(as_tuple=False)  # class-major, score order
    if incomplete is not None and bool(incomplete.item()):
        keep, num = hip_ops.nms_bev_multiclass(boxes_nms, rank, count, cfg["nms_thr"], rotated=rotated, max_keep=cap)
        kept = (torch.arange(n, device=st.device)[None, :] < num[:, None]).nonzero(as_tuple=False)
    labels = kept[:, 0]
    box_idx = order[labels, keep[labels, kept[:, 1]]]
    bboxes, scores = mlvl_bboxes[box_idx], st[labels, box_idx]
    if bboxes.shape[0] > max_num:
        inds = scores.sort(descending=True)[1][:max_num]
        bboxes, labels, scores = bboxes[inds, :], labels[inds], scores[inds]
    return bboxes, scores, labels


def bbox3d2result(bboxes, scores, labels, attrs=None):
    """mmdet3d.core.bbox3d2result: results on the host, the form the dataset evaluators take."""
    t = getattr(bboxes, "tensor", None)
    hr = getattr(bboxes, "_host_rows", None)  # (boxes | score | label) rows already on the host (the fused box tail's one read-back)
    if (hr is not None and t is not None and attrs is None and hr[1] is t and t._version == hr[2] and hr[3] is scores
            and scores._version == hr[4] and hr[5] is labels and labels._version == hr[6]):
        # ... and still what the device tensors hold: the very tensors the box tail returned, un-edited since
        host = hr[0]
        c = t.size(1)
        return dict(boxes_3d=type(bboxes)._wrap(host[:, :c].contiguous(), c, getattr(bboxes, "with_yaw", True)),
                    scores_3d=host[:, c].contiguous(), labels_3d=host[:, c + 1].to(labels.dtype))
    if t is not None and t.is_cuda and t.dtype == torch.float32 and scores.dtype == torch.float32 and labels.numel() == len(t):
        # one device -> host transfer instead of three (class indices are exact in fp32)
        packed = torch.cat([t, scores[:, None], labels[:, None].to(torch.float32)], dim=1).cpu()
        c = t.size(1)
        host_boxes = bboxes.to("cpu") if len(t) == 0 else type(bboxes)(packed[:, :c].contiguous(), box_dim=c, with_yaw=getattr(bboxes, "with_yaw", True))
        result = dict(boxes_3d=host_boxes, scores_3d=packed[:, c].contiguous(), labels_3d=packed[:, c + 1].to(labels.dtype))
    else:
        result = dict(boxes_3d=bboxes.to("cpu"), scores_3d=scores.cpu(), labels_3d=labels.cpu())
    if attrs is not None:
        result["attrs_3d"] = attrs.cpu()
    return result
