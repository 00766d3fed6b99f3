"""Query heads, inference side: `SparseClusterHead` (projects/mmdet3d_plugin/models/dense_heads/sparse_cluster_head.py
:18-115, split_by_batch :269-278), `FSDSeparateHead` / `SparseClusterHeadV2` (sparse_cluster_head_v2.py:18-167,
get_bboxes :447-608) and `FrustumClusterHead` (frustum_cluster_head.py:19-95, get_bboxes :500-697).

Same constructor arguments, sub-module names (state-dict keys `shared_mlp.*`, `task_heads.N.<attr>.*`) and return
conventions.  The MLPs are the fused Linear -> LayerNorm+GELU blocks of ops/sst_ops.py; box decoding is the coder of
core/bbox.py; NMS is the HIP kernel pair K20.  Losses / target assignment (train time) are not built: `loss` raises.
"""
from .... import switches
import copy
import os

import torch
import torch.nn as nn

from .... import hip_ops
from ...core.bbox import BasePointBBoxCoder, LiDARInstance3DBoxes, box3d_multiclass_nms, xywhr2xyxyr
from ...ops.sst_ops import build_mlp
from ...registry import BBOX_ASSIGNERS, BBOX_CODERS, HEADS, build_head, build_loss


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    return cfg.get(key, default)


@HEADS.register_module()
class FSDSeparateHead(nn.Module):
    def __init__(self, in_channels, attrs, norm_cfg=dict(type="LN"), act="relu", init_cfg=None):
        super().__init__()
        self.attrs = attrs
        for attr_name in self.attrs:
            out_dim, num_layer, hidden_dim = self.attrs[attr_name]
            self.add_module(attr_name, build_mlp(in_channels, [hidden_dim] * num_layer + [out_dim], norm_cfg, is_head=True, act=act))

    def forward(self, x):
        fused = self._forward_sliced(x)
        if fused is not None:
            return fused
        return {attr_name: getattr(self, attr_name)(x) for attr_name in self.attrs}

    def accepts_planes(self, n_rows):
        """Will `forward` take the query features in plane form (RowPlanes) for `n_rows` rows?  (Its first layer then runs on K22h.)"""
        if not switches.K22H or self.training or n_rows < switches.K22H_MIN_ROWS or len(self.attrs) <= 1:
            return False
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return False
        plan = self._sliced_plan()
        return plan is not None and "first_f16" in plan

    # ---- inference on the GPU: the attribute branches are independent MLPs of one shape on the same input, so layer i of
    # all of them is ONE K22 launch (fsf_linear_norm_act_sliced) instead of one per attribute — per query head 15 launches
    # (10 fused blocks + 5 library GEMMs for the 2..10-wide outputs) become 3, and a 10 k-row x 1024 -> 128 layer, which
    # fills a third of the chip on its own, runs five abreast.  Per branch the arithmetic is unchanged.
    _OUT_PAD = 16  # output channels per branch in the last (plain Linear) layer's launch: padded with zero weight rows

    def _sliced_plan(self):
        from ...ops.sst_ops import MLPBlock

        names = list(self.attrs)
        mlps = [getattr(self, a) for a in names]
        depth = len(mlps[0])
        if any(len(m) != depth for m in mlps) or depth < 2:
            return None
        params = self.__dict__.get("_fsf_param_list")  # (the Parameter objects: walking the module tree per frame and head costs ~50 us)
        if params is None:
            params = self.__dict__["_fsf_param_list"] = [p for m in mlps for p in m.parameters()]
        key = tuple((p.data_ptr(), p._version) for p in params)
        cache = self.__dict__.get("_fsf_sliced")
        if cache is not None and cache[0] == key:
            return cache[1]
        plan = None
        blocks_ok = True
        for li in range(depth - 1):
            layer = [m[li] for m in mlps]
            if not all(isinstance(b, MLPBlock) and len(b) == 3 and isinstance(b[0], nn.Linear) and isinstance(b[1], nn.LayerNorm)
                       and b[1].elementwise_affine and len(b[1].normalized_shape) == 1 for b in layer):
                blocks_ok = False
                break
            ref = layer[0]
            act_code = "relu" if isinstance(ref[2], nn.ReLU) else (
                "gelu" if isinstance(ref[2], nn.GELU) and getattr(ref[2], "approximate", "none") == "none" else None)
            if act_code is None or any(type(b[2]) is not type(ref[2]) or b[0].weight.shape != ref[0].weight.shape
                                       or b[1].eps != ref[1].eps or (b[0].bias is None) != (ref[0].bias is None) for b in layer):
                blocks_ok = False
                break
            if ref[0].out_features % 4 or ref[0].out_features > 128:
                blocks_ok = False
                break
        last = [m[depth - 1] for m in mlps]
        if blocks_ok and all(isinstance(l, nn.Linear) and l.out_features <= self._OUT_PAD for l in last) \
                and len({l.in_features for l in last}) == 1:
            with torch.no_grad():
                layers = []
                for li in range(depth - 1):
                    layer = [m[li] for m in mlps]
                    h = layer[0][0].out_features
                    w = torch.cat([b[0].weight for b in layer], 0)
                    layers.append(dict(
                        k=layer[0][0].in_features, h=h, planes=hip_ops.linear_prepare_weight_sliced(w, len(names), h),
                        bias=torch.cat([b[0].bias for b in layer]) if layer[0][0].bias is not None else None,
                        gamma=torch.cat([b[1].weight for b in layer]), beta=torch.cat([b[1].bias for b in layer]),
                        eps=layer[0][1].eps,
                        act="relu" if isinstance(layer[0][2], nn.ReLU) else "gelu"))
                pad = self._OUT_PAD
                kin = last[0].in_features
                w = torch.zeros((len(names) * pad, kin), dtype=torch.float32, device=last[0].weight.device)
                b = torch.zeros((len(names) * pad,), dtype=torch.float32, device=w.device)
                for i, l in enumerate(last):
                    w[i * pad:i * pad + l.out_features] = l.weight
                    if l.bias is not None:
                        b[i * pad:i * pad + l.out_features] = l.bias
                plan = dict(names=names, layers=layers, out_k=kin, out_planes=hip_ops.linear_prepare_weight_sliced(w, len(names), pad),
                            out_bias=b, out_dims=[l.out_features for l in last])
                # the branches' FIRST layer on K22h (f16 x 3 planes; all branches read the same >= 256-wide query features)
                first = [m[0][0] for m in mlps]
                k0, h0 = first[0].in_features, first[0].out_features
                if switches.K22H and k0 % 32 == 0 and k0 >= 256 and 64 < h0 <= 128 and h0 % 4 == 0:
                    plan["first_f16"] = hip_ops.linear_prepare_weight_f16(torch.cat([l.weight for l in first], 0), h0)
        self.__dict__["_fsf_sliced"] = (key, plan)
        return plan

    def _forward_sliced(self, x):
        from ...ops.sst_ops import as_row_planes

        is_planes = isinstance(x, hip_ops.RowPlanes)
        if self.training or (torch.is_grad_enabled() and ((not is_planes and x.requires_grad) or any(p.requires_grad for p in self.parameters()))):
            return None
        if not is_planes and not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.size(0) >= 1 and x.stride(1) == 1
                                  and (x.size(0) == 1 or x.stride(0) % 4 == 0) and x.data_ptr() % 16 == 0 and len(self.attrs) > 1):
            return None
        plan = self._sliced_plan()
        k_in, n_rows = (x.c, x.n) if is_planes else (x.size(1), x.size(0))
        if plan is None or plan["layers"][0]["k"] != k_in:
            return None
        ns = len(plan["names"])
        h_prev = 0
        for li, lay in enumerate(plan["layers"]):
            if li == 0 and "first_f16" in plan and (is_planes or (n_rows >= switches.K22H_MIN_ROWS and hip_ops.rows_to_planes_supported(x))):
                x = hip_ops.linear_planes_norm_act(as_row_planes(x), plan["first_f16"], ns * lay["h"], lay["h"], bias=lay["bias"], norm="ln",
                                                   gamma=lay["gamma"], beta=lay["beta"], eps=lay["eps"], act=lay["act"])
                h_prev = lay["h"]
                continue
            if isinstance(x, hip_ops.RowPlanes):
                return None
            x = hip_ops.linear_norm_act_sliced(x, lay["k"], h_prev, lay["planes"], ns, lay["h"], bias=lay["bias"], norm="ln",
                                               gamma=lay["gamma"], beta=lay["beta"], eps=lay["eps"], act=lay["act"])
            h_prev = lay["h"]
        y = hip_ops.linear_norm_act_sliced(x, plan["out_k"], h_prev, plan["out_planes"], ns, self._OUT_PAD, bias=plan["out_bias"])
        pad = self._OUT_PAD
        return {a: y[:, i * pad:i * pad + d] for i, (a, d) in enumerate(zip(plan["names"], plan["out_dims"]))}


@HEADS.register_module()
class SparseClusterHead(nn.Module):
    def __init__(self, num_classes, bbox_coder, loss_cls, loss_center, loss_size, loss_rot, in_channel, shared_mlp_dims,
                 shared_dropout=0, cls_mlp=None, reg_mlp=None, iou_mlp=None, train_cfg=None, test_cfg=None,
                 norm_cfg=dict(type="LN"), loss_iou=None, act="relu", corner_loss_cfg=None, enlarge_width=None,
                 as_rpn=False, init_cfg=None):
        super().__init__()
        self.print_info = {}
        self.loss_center, self.loss_size = build_loss(loss_center), build_loss(loss_size)
        self.loss_rot, self.loss_cls = build_loss(loss_rot), build_loss(loss_cls)
        self.bbox_coder = BBOX_CODERS.build(bbox_coder)
        self.box_code_size = self.bbox_coder.code_size
        self.corner_loss_cfg = corner_loss_cfg
        self.num_classes = num_classes
        self.enlarge_width = enlarge_width
        self.sync_reg_avg_factor = False if train_cfg is None else train_cfg.get("sync_reg_avg_factor", True)
        self.sync_cls_avg_factor = False if train_cfg is None else train_cfg.get("sync_cls_avg_factor", True)
        self.as_rpn = as_rpn
        self.train_cfg = self.test_cfg = None
        if train_cfg is not None:
            self.cfg = self.train_cfg = train_cfg
        if test_cfg is not None:
            self.cfg = self.test_cfg = test_cfg
        self.num_anchors = num_anchors = 1
        self.loss_iou = build_loss(loss_iou) if loss_iou is not None else None
        self.fp16_enabled = False
        self.shared_mlp = None
        if len(shared_mlp_dims) > 0:
            self.shared_mlp = build_mlp(in_channel, shared_mlp_dims, norm_cfg, act=act, dropout=shared_dropout)
        end_channel = shared_mlp_dims[-1] if len(shared_mlp_dims) > 0 else in_channel
        if cls_mlp is not None:
            self.conv_cls = build_mlp(end_channel, cls_mlp + [num_classes * num_anchors], norm_cfg, True, act=act)
        else:
            self.conv_cls = nn.Linear(end_channel, num_classes * num_anchors)
        if reg_mlp is not None:
            self.conv_reg = build_mlp(end_channel, reg_mlp + [self.box_code_size * num_anchors], norm_cfg, True, act=act)
        else:
            self.conv_reg = nn.Linear(end_channel, self.box_code_size * num_anchors)
        self.save_list = []

    def forward(self, feats, pts_xyz=None, pts_inds=None):
        if self.shared_mlp is not None:
            feats = self.shared_mlp(feats)
        return dict(cls_logits=self.conv_cls(feats), reg_preds=self.conv_reg(feats))

    def split_by_batch(self, data, batch_idx, batch_size):
        if batch_size == 1:
            return [data]
        return [data[batch_idx == i] for i in range(batch_size)]

    def combine_by_batch(self, data_list, batch_idx, batch_size):
        assert len(data_list) == batch_size
        if data_list[0] is None:
            return None
        full = data_list[0].new_zeros((len(batch_idx),) + data_list[0].shape[1:])
        for i, data in enumerate(data_list):
            full[batch_idx == i] = data
        return full

    def loss(self, *args, **kwargs):
        raise NotImplementedError("head losses / target assignment are train-time code outside the built path")


@HEADS.register_module()
class SparseClusterHeadV2(SparseClusterHead):
    BATCH_COL = 1  # cluster_inds rows are (class, batch, cluster id)  (sparse_cluster_head_v2.py:509-512)
    EMPTY_BOX_DIM = 7

    def __init__(self, num_classes, bbox_coder, loss_cls, loss_center, loss_size, loss_rot, in_channel, shared_mlp_dims,
                 tasks, class_names, common_attrs, num_cls_layer, cls_hidden_dim, separate_head, cls_mlp=None,
                 reg_mlp=None, iou_mlp=None, train_cfg=None, test_cfg=None, norm_cfg=dict(type="LN"), loss_iou=None,
                 act="relu", corner_loss_cfg=None, enlarge_width=None, as_rpn=False, init_cfg=None, shared_dropout=0,
                 loss_vel=None):
        super().__init__(num_classes, bbox_coder, loss_cls, loss_center, loss_size, loss_rot, in_channel, shared_mlp_dims,
                         shared_dropout, cls_mlp, reg_mlp, iou_mlp, train_cfg, test_cfg, norm_cfg, loss_iou, act,
                         corner_loss_cfg, enlarge_width, as_rpn, init_cfg)
        self.conv_cls = None  # overridden by the per-task separate heads
        self.conv_reg = None
        sep_head_in_channels = shared_mlp_dims[-1] if self.shared_mlp is not None else in_channel
        self.tasks = tasks
        self.task_heads = nn.ModuleList()
        for t in tasks:
            attrs = copy.deepcopy(dict(common_attrs))
            attrs.update(dict(score=(len(t["class_names"]), num_cls_layer, cls_hidden_dim)))
            head_cfg = dict(separate_head)
            head_cfg.update(in_channels=sep_head_in_channels, attrs=attrs)
            self.task_heads.append(build_head(head_cfg))
        self.class_names = class_names
        self.loss_vel = build_loss(loss_vel) if loss_vel is not None else None

    def forward(self, feats, pts_xyz=None, pts_inds=None):
        if self.shared_mlp is not None:
            # (the shared MLP's output only feeds the task heads' branches: in plane form when every one of them takes it)
            as_planes = (torch.is_tensor(feats) and feats.dim() == 2 and feats.is_cuda
                         and all(getattr(h, "accepts_planes", lambda n: False)(feats.size(0)) for h in self.task_heads)
                         and self.task_heads[0]._sliced_plan()["layers"][0]["k"] == self.shared_mlp[-1][0].out_features)
            feats = self.shared_mlp(feats, planes_out=True) if as_planes else self.shared_mlp(feats)
        cls_logit_list, reg_pred_list, iou_logits_list = [], [], []
        for h in self.task_heads:
            ret = h(feats)
            parts = [ret["center"], ret["dim"], ret["rot"]] + ([ret["vel"]] if "vel" in ret else [])
            reg_pred_list.append(torch.cat(parts, dim=-1))  # same column order as v1's single regression branch
            cls_logit_list.append(ret["score"])
            if "iou" in ret:
                iou_logits_list.append(ret["iou"])
        outs = dict(cls_logits=cls_logit_list, reg_preds=reg_pred_list)
        if len(iou_logits_list) > 0:
            outs.update(iou_logits=iou_logits_list)
        return outs

    # ------------------------------------------------------------------------------------------- boxes
    @torch.no_grad()
    def get_bboxes(self, cls_logits, reg_preds, cluster_xyz, cluster_inds, input_metas, iou_logits=None, rescale=False):
        return self._get_bboxes_all_tasks(cls_logits, reg_preds, None, cluster_xyz, cluster_inds, input_metas, iou_logits)

    def _get_bboxes_all_tasks(self, cls_logits, reg_preds, preds_2d, cluster_xyz, cluster_inds, input_metas, iou_logits):
        assert isinstance(cls_logits, list) and isinstance(reg_preds, list)
        assert len(cls_logits) == len(reg_preds) == len(self.tasks)
        per_task = [self.get_bboxes_single_task(i, cls_logits[i], reg_preds[i], preds_2d, cluster_xyz, cluster_inds,
                                                input_metas, iou_logits[i] if iou_logits is not None else None)
                    for i in range(len(self.tasks))]
        batch_size = len(input_metas)
        assert len(per_task[0]) <= batch_size
        if len(per_task) == 1:  # one task (the nuScenes / AV2 configs): nothing to concatenate
            return [tuple(per_task[0][b_idx]) for b_idx in range(batch_size)]
        out = []
        for b_idx in range(batch_size):
            out.append((LiDARInstance3DBoxes.cat([t[b_idx][0] for t in per_task]),
                        torch.cat([t[b_idx][1] for t in per_task], dim=0),
                        torch.cat([t[b_idx][2] for t in per_task], dim=0)))
        return out

    @torch.no_grad()
    def get_bboxes_single_task(self, task_id, cls_logits, reg_preds, preds_2d, cluster_xyz, cluster_inds, input_metas,
                               iou_logits=None, rescale=False):
        batch_inds = cluster_inds if cluster_inds.ndim == 1 else cluster_inds[:, self.BATCH_COL]
        batch_size = len(input_metas)
        split = lambda t: self.split_by_batch(t, batch_inds, batch_size)  # noqa: E731
        cls_l, reg_l, xyz_l = split(cls_logits), split(reg_preds), split(cluster_xyz)
        p2d_l = split(preds_2d) if preds_2d is not None else [None] * len(cls_l)
        iou_l = split(iou_logits) if iou_logits is not None else [None] * len(cls_l)
        return [self._get_bboxes_single(task_id, cls_l[b], iou_l[b], reg_l[b], p2d_l[b], xyz_l[b], input_metas[b])
                for b in range(len(cls_l))]

    def _box_type(self, input_meta):
        return input_meta.get("box_type_3d", LiDARInstance3DBoxes) if isinstance(input_meta, dict) else LiDARInstance3DBoxes

    def _get_bboxes_single(self, task_id, cls_logits, iou_logits, reg_preds, preds_2d, cluster_xyz, input_meta):
        """One sample, one task: sigmoid scores -> (optional top-k) -> decode -> per-class rotated BEV NMS."""
        if self.as_rpn:
            cfg = self.train_cfg["rpn"] if self.training else self.test_cfg["rpn"]
        else:
            cfg = self.test_cfg
        box_type = self._box_type(input_meta)
        assert cls_logits.size(0) == reg_preds.size(0) == cluster_xyz.size(0)
        assert cls_logits.size(1) == len(self.tasks[task_id]["class_names"])
        assert reg_preds.size(1) == self.box_code_size
        if len(cls_logits) == 0:
            empty = reg_preds.new_zeros((0, self.EMPTY_BOX_DIM))
            return box_type(empty, box_dim=self.EMPTY_BOX_DIM), reg_preds.new_zeros(0), reg_preds.new_zeros(0)
        fused = self._box_tail_fused(task_id, cfg, box_type, cls_logits, iou_logits, reg_preds, cluster_xyz)
        if fused is not None:
            return fused
        scores = cls_logits.sigmoid()
        if iou_logits is not None:
            a = cfg.get("iou_score_weight", 0.5)
            scores = (scores ** (1 - a)) * (iou_logits.sigmoid() ** a)
        nms_pre = cfg.get("nms_pre", -1)
        if nms_pre > 0 and scores.shape[0] > nms_pre:
            topk_inds = scores.max(dim=1)[0].topk(nms_pre)[1]
            reg_preds, scores, cluster_xyz = reg_preds[topk_inds, :], scores[topk_inds, :], cluster_xyz[topk_inds, :]
        bboxes = self.bbox_coder.decode(reg_preds, cluster_xyz)
        bboxes = self._append_debug_columns(bboxes, preds_2d)
        bboxes_for_nms = xywhr2xyxyr(box_type(bboxes, box_dim=bboxes.size(1)).bev)
        scores = torch.cat([scores, scores.new_zeros(scores.shape[0], 1)], dim=1)  # dummy background column
        out_bboxes, out_scores, out_labels = box3d_multiclass_nms(bboxes, bboxes_for_nms, scores, cfg.get("score_thr", 0),
                                                                  cfg["max_num"], cfg)
        out_bboxes, out_scores = self._strip_debug_columns(out_bboxes, out_scores)
        out_bboxes = box_type(out_bboxes, out_bboxes.size(1))
        # task-local label -> global class index: one table gather (the reference loops over the class names with a
        # masked assignment each — a hidden device sync per class — and asserts on the host that every label was mapped;
        # with a table every label is mapped by construction)
        lut = self._label_lut(task_id, out_labels.device)
        new_labels = lut[out_labels] if len(out_labels) > 0 else torch.zeros_like(out_labels) - 1
        return out_bboxes, out_scores, new_labels

    def _label_lut(self, task_id, device):
        luts = self.__dict__.setdefault("_label_luts", {})
        lut = luts.get((task_id, device))
        if lut is None:
            lut = torch.tensor([self.class_names.index(name) for name in self.tasks[task_id]["class_names"]], dtype=torch.long,
                               device=device)
            luts[(task_id, device)] = lut
        return lut

    def _box_tail_fused(self, task_id, cfg, box_type, cls_logits, iou_logits, reg_preds, cluster_xyz):
        """Inference on the GPU: everything between the head's outputs and the host-side result as four C-ABI calls (K24 decode,
        K24 class ranks, K20 capped multi-class NMS, K24 selection) and ONE read-back, instead of ~95 ATen launches and three host
        round trips.  Returns None when the configuration is outside what that path takes (the generic path below runs)."""
        max_num = cfg.get("max_num", 0)
        nms_pre = cfg.get("nms_pre", -1)
        n, c = cls_logits.shape
        if not (switches.BOX_TAIL_FUSED and cls_logits.is_cuda and cls_logits.dtype == torch.float32 and reg_preds.dtype == torch.float32
                and cluster_xyz.dtype == torch.float32 and iou_logits is None and not torch.is_grad_enabled()
                and box_type is LiDARInstance3DBoxes and getattr(self, "vis_dir", None) is None
                and isinstance(max_num, int) and max_num > 0 and c * max_num <= hip_ops.nms_select_capacity()
                and c <= hip_ops.box_tail_max_classes()
                and not (nms_pre > 0 and n > nms_pre) and type(self.bbox_coder) is BasePointBBoxCoder):
            return None
        boxes, boxes_nms, scores_t = hip_ops.decode_cluster_boxes(cls_logits, reg_preds, cluster_xyz, self.bbox_coder.EPS)
        order, rank, count = hip_ops.class_rank_desc(scores_t, cfg.get("score_thr", 0))
        keep, num, incomplete = hip_ops.nms_bev_multiclass(boxes_nms, rank, count, cfg["nms_thr"],
                                                           rotated=bool(cfg.get("use_rotate_nms", False)), max_keep=max_num, windowed=True)
        buf = hip_ops.nms_select(boxes, scores_t, order, keep, num, max_num, max_num, self._label_lut(task_id, cls_logits.device),
                                 incomplete)
        d = boxes.size(1)
        hook = self.__dict__.pop("_before_readback", None)
        if hook is not None:
            hook()  # (FSF: the next frame's front, issued while the device works through this frame's tail)
        host = buf.cpu()  # the frame's one read-back for the box tail: rows + (rows written, boxes kept, incomplete)
        meta = host[max_num * (d + 2):].view(torch.int32)
        if int(meta[2]) != 0:  # a class ran out of its mask window before max_num keeps (never seen): the generic path repeats it in full
            return None
        k = int(meta[0])
        rows = buf[:max_num * (d + 2)].view(max_num, d + 2)[:k]
        # (views of `buf`: row stride d + 2 — consumers index / reshape them; `.view()` needs `.contiguous()` first)
        out_bboxes, out_scores, out_labels = LiDARInstance3DBoxes._wrap(rows[:, :d], d), rows[:, d], rows[:, d + 1].long()
        # the same rows already on the host, for bbox3d2result — valid only while nobody has edited the device tensors since
        # (rescale / flip / score re-weighting between get_bboxes and bbox3d2result): identity + version counters travel along
        out_bboxes._host_rows = (host[:max_num * (d + 2)].view(max_num, d + 2)[:k], out_bboxes.tensor, out_bboxes.tensor._version,
                                 out_scores, out_scores._version, out_labels, out_labels._version)
        return out_bboxes, out_scores, out_labels

    def _append_debug_columns(self, bboxes, preds_2d):
        return bboxes

    def _strip_debug_columns(self, out_bboxes, out_scores):
        return out_bboxes, out_scores


@HEADS.register_module()
class FrustumClusterHead(SparseClusterHeadV2):
    BATCH_COL = 0  # frustum query coors are (batch, 0, obj id)  (frustum_cluster_head.py:562-565)
    EMPTY_BOX_DIM = 9

    def __init__(self, num_classes, bbox_coder, loss_cls, loss_center, loss_size, loss_rot, in_channel, shared_mlp_dims,
                 tasks, class_names, common_attrs, num_cls_layer, cls_hidden_dim, separate_head, cls_mlp=None,
                 reg_mlp=None, iou_mlp=None, train_cfg=dict(), test_cfg=dict(), norm_cfg=dict(type="LN"), loss_iou=None,
                 act="relu", corner_loss_cfg=None, enlarge_width=None, as_rpn=False, init_cfg=None, shared_dropout=0,
                 loss_vel=None, assigner=None, num_objs=250, vis_dir=None, use_one_to_one=False):
        super().__init__(num_classes, bbox_coder, loss_cls, loss_center, loss_size, loss_rot, in_channel, shared_mlp_dims,
                         tasks, class_names, common_attrs, num_cls_layer, cls_hidden_dim, separate_head, cls_mlp, reg_mlp,
                         iou_mlp, train_cfg, test_cfg, norm_cfg, loss_iou, act, corner_loss_cfg, enlarge_width, as_rpn,
                         init_cfg, shared_dropout, loss_vel)
        self.assigner = BBOX_ASSIGNERS.build(assigner) if assigner is not None else None
        self.num_objs = num_objs
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.task_info = {}
        self.vis_dir = vis_dir
        self.use_one_to_one = use_one_to_one

    @torch.no_grad()
    def get_bboxes(self, cls_logits, reg_preds, preds_2d, cluster_xyz, cluster_inds, input_metas, iou_logits=None,
                   rescale=False):
        return self._get_bboxes_all_tasks(cls_logits, reg_preds, preds_2d, cluster_xyz, cluster_inds, input_metas, iou_logits)

    def _append_debug_columns(self, bboxes, preds_2d):
        if self.vis_dir is not None:  # visualisation runs carry the 2-D object id through NMS (:629-631)
            bboxes = torch.cat([bboxes, preds_2d[:, 7:8]], dim=-1)
        return bboxes

    def _strip_debug_columns(self, out_bboxes, out_scores):
        if self.vis_dir is not None:
            out_bboxes, out_obj_id = out_bboxes[:, :-1], out_bboxes[:, -1]
            out_scores = out_scores + out_obj_id
        return out_bboxes, out_scores
