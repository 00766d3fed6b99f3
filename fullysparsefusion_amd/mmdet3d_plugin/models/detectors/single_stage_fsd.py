"""`VoteSegmentor`, `SingleStageFSD`, `ClusterAssigner`: inference path of
projects/mmdet3d_plugin/models/detectors/single_stage_fsd.py (voxelize :206-226, extract_feat :228-245,
voxel_downsample :263-273, simple_test :342-378; SingleStageFSD.extract_feat :458-474, pre_voxelize :585-605,
group_sample :802-865, get_fg_mask :742-784, ClusterAssigner :903-982).

Same class / method names and argument meaning; torch.unique / torch_scatter / Voxelization / scipy CCL calls go
to the HIP library.  Training-only branches (targets, losses, SSG/Hybrid assigners) are outside this round.
"""
from .... import switches
import os

import math

import torch
from torch import nn

from .... import hip_ops
from ...ops.sst_ops import (GatheredRows, RowsMinusGroup, gather_by_inverse, get_inner_win_inds, scatter_mean_multi, scatter_v2, seed_unique_cache,
                            seed_unique_result, unique_with_plan, with_key_bounds)
from ...ops.voxel import Voxelization
from ...registry import (DETECTORS, SEGMENTORS, build_backbone, build_detector, build_head, build_middle_encoder,
                         build_neck, build_voxel_encoder)


def filter_almost_empty(coors, min_points):
    """single_stage_fsd.py:31-35."""
    new_coors, unq_inv, unq_cnt = unique_with_plan(coors)
    return unq_cnt[unq_inv] >= min_points


def find_connected_componets(points, batch_idx, dist):
    """single_stage_fsd.py:45-67 (per-sample components, ids offset sample by sample) without the host
    round trip: one device union-find over all samples with the batch index as an extra adjacency condition,
    then relabel in (sample, first-member) order."""
    assert len(points) > 0
    labels = hip_ops.connected_components(points, dist, batch_idx=batch_idx)
    # device labels are numbered by first member overall; the reference numbers sample 0's components first
    key = torch.stack([batch_idx.long(), labels.long()], 1)
    _, inv, _ = unique_with_plan(key)
    return inv.int()


def find_connected_componets_single_batch(points, batch_idx, dist):
    """single_stage_fsd.py:69-82: the batch index is ignored (test-time, one sample)."""
    return hip_ops.connected_components(points, dist)


def modify_cluster_by_class(cluster_inds_list):
    """single_stage_fsd.py:139-152: prepend the class id to every (batch, cluster) pair."""
    new_list = []
    for i, inds in enumerate(cluster_inds_list):
        cls_pad = inds.new_ones((len(inds),)) * i
        new_list.append(torch.cat([cls_pad[:, None], inds], 1))
    return new_list


@SEGMENTORS.register_module()
@DETECTORS.register_module()
class VoteSegmentor(nn.Module):
    def __init__(self, voxel_layer, voxel_encoder, middle_encoder, backbone, segmentation_head, decode_neck=None,
                 auxiliary_head=None, voxel_downsampling_size=None, train_cfg=None, test_cfg=None, init_cfg=None,
                 pretrained=None, tanh_dims=[], **extra_kwargs):
        super().__init__()
        self.voxel_layer = Voxelization(**voxel_layer)
        self.voxel_encoder = build_voxel_encoder(voxel_encoder)
        self.middle_encoder = build_middle_encoder(middle_encoder)
        self.backbone = build_backbone(backbone)
        self.segmentation_head = build_head(segmentation_head)
        self.segmentation_head.train_cfg = train_cfg
        self.segmentation_head.test_cfg = test_cfg
        self.decode_neck = build_neck(decode_neck)
        assert voxel_encoder["type"] == "DynamicScatterVFE"
        self.print_info = {}
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.cfg = train_cfg if train_cfg is not None else test_cfg
        self.num_classes = segmentation_head["num_classes"]
        self.save_list = []
        self.point_cloud_range = voxel_layer["point_cloud_range"]
        self.voxel_size = voxel_layer["voxel_size"]
        self.voxel_downsampling_size = voxel_downsampling_size
        self.tanh_dims = tanh_dims

    @torch.no_grad()
    def voxelize(self, points):
        """list of [N_b, C] -> (points [N, C], coors i64 [N, 4] (b,z,y,x)); one fused kernel per sample."""
        return self.voxel_layer.forward_batch(points)

    def extract_feat(self, points, img_metas):
        return self.extract_feat_finish(self.extract_feat_begin(points))

    def extract_feat_begin(self, points):
        """`extract_feat` up to the backbone's first convolution (:226-234): voxelization, DynamicScatterVFE, the voxel dict and — where
        the backbone splits its forward (SimpleSparseUNet.begin) — its row order, first index plan and input planes.  A chain of ~60
        small launches with one host wait (the voxel count): the part of a frame the HOST bounds.  `extract_feat_finish` runs the rest."""
        batch_points, coors = self.voxelize(points)
        self.voxel_encoder.max_batch = len(points)
        voxel_features, voxel_coors, voxel2point_inds = self.voxel_encoder(batch_points, coors, return_inv=True)
        voxel_info = self.middle_encoder(voxel_features, voxel_coors, batch_size=len(points))
        steps = self.backbone.begin(voxel_info) if hasattr(self.backbone, "begin") else None
        return dict(batch_points=batch_points, coors=coors, voxel2point_inds=voxel2point_inds, voxel_info=voxel_info, backbone_steps=steps)

    def extract_feat_finish(self, front):
        batch_points, coors, voxel2point_inds = front["batch_points"], front["coors"], front["voxel2point_inds"]
        voxel_info, steps = front["voxel_info"], front["backbone_steps"]
        x = (self.backbone.finish(steps) if steps is not None else self.backbone(voxel_info))[0]
        padding = -1
        assert "shuffle_inds" not in voxel_info  # SST-only branch (:238-241)
        permuted = getattr(x, "permuted", None)
        if permuted is not None and getattr(self.decode_neck, "takes_row_map", False) and not torch.is_grad_enabled():
            # (the U-Net's rows in ITS order + the map back: the neck gathers one row per point through the composed index)
            out = self.decode_neck(batch_points, coors, permuted[0], voxel2point_inds, padding, row_map=permuted[1])
        else:
            voxel_feats_reorder = x["voxel_feats"]
            out = self.decode_neck(batch_points, coors, voxel_feats_reorder, voxel2point_inds, padding)
        return out, coors, batch_points

    def voxel_downsample(self, points_list):
        out_points_list = []
        for points in points_list:
            coors = hip_ops.voxelize_divfloor(points, self.voxel_downsampling_size, self.point_cloud_range[:3], order="xyz")
            out_points, _ = scatter_v2(points, coors, mode="avg", return_inv=False)
            out_points_list.append(out_points)
        return out_points_list

    def _prep(self, points):
        if self.tanh_dims == []:
            return [p.contiguous() for p in points]
        if self.tanh_dims is not None:
            for p in points:
                p[:, self.tanh_dims] = torch.tanh(p[:, self.tanh_dims])
            return [p.contiguous() for p in points]
        if points[0].size(1) in (4, 5):
            return [torch.cat([p[:, :3], torch.tanh(p[:, 3:])], dim=1) for p in points]
        return points

    def simple_test(self, points, img_metas, gt_bboxes_3d=None, gt_labels_3d=None, extract_feat_only=False,
                    rescale=False, front=None):
        if front is not None:  # (FSF's frame front: `_prep` + `extract_feat_begin` already ran on these points)
            x, pts_coors, points = self.extract_feat_finish(front)
        else:
            points = self._prep(points)
            x, pts_coors, points = self.extract_feat(points, img_metas)
        if extract_feat_only:
            return x, pts_coors, points
        feats, valid_pts_mask = x[0], x[1]
        points = points[valid_pts_mask]
        pts_coors = pts_coors[valid_pts_mask]
        seg_logits, vote_preds = self.segmentation_head.forward_test(feats, img_metas, self.test_cfg)
        offsets = self.segmentation_head.decode_vote_targets(vote_preds)
        return dict(seg_points=points, seg_logits=seg_logits, seg_vote_preds=vote_preds, offsets=offsets,
                    seg_feats=feats, batch_idx=pts_coors[:, 0])

    def forward_train(self, *args, **kwargs):
        raise NotImplementedError("training path (targets + losses) is outside this round's hot path")


@DETECTORS.register_module()
class SingleStageFSD(nn.Module):
    def __init__(self, backbone, segmentor, voxel_layer=None, voxel_encoder=None, middle_encoder=None, neck=None,
                 bbox_head=None, train_cfg=None, test_cfg=None, cluster_assigner=None, pretrained=None, init_cfg=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        if bbox_head is not None:
            bbox_head = dict(bbox_head)
            # mmdet3d SingleStage3DDetector injects the model-level cfgs into the head
            bbox_head.update(train_cfg=train_cfg)
            bbox_head.update(test_cfg=test_cfg)
            self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if voxel_layer is not None:
            self.voxel_layer = Voxelization(**voxel_layer)
        if voxel_encoder is not None:
            self.voxel_encoder = build_voxel_encoder(voxel_encoder)
        if middle_encoder is not None:
            self.middle_encoder = build_middle_encoder(middle_encoder)
        self.segmentor = build_detector(segmentor)
        self.head_type = bbox_head["type"]
        self.num_classes = bbox_head["num_classes"]
        self.cfg = self.train_cfg if self.train_cfg else self.test_cfg
        cluster_assigner = dict(cluster_assigner)
        if "radius" in cluster_assigner or "hybrid" in cluster_assigner:
            raise NotImplementedError("SSGAssigner / HybridAssigner are not configured by the FSF configs")
        self.cluster_assigner = ClusterAssigner(**cluster_assigner)
        self.cluster_assigner.num_classes = self.num_classes
        self.print_info = {}
        self.as_rpn = bbox_head.get("as_rpn", False)
        if hasattr(self.backbone, "point_feats_needed"):
            self.backbone.point_feats_needed = bool(self.as_rpn)  # extract_feat only hands the per-point features on for an RPN
        self.runtime_info = dict() if self.cfg.get("disable_pretrain", False) else None

    def extract_feat(self, points, pts_feats, pts_cluster_inds, img_metas, center_preds):
        """:458-474 — cluster centroid of the vote centres, per-point offset to it, then the SIR backbone."""
        pre = self.__dict__.pop("_cluster_xyz_pre", None)
        if pre is not None and pre[0] is pts_cluster_inds and pre[1] is center_preds and not torch.is_grad_enabled():
            cluster_xyz = pre[2]  # (K30 formed the centroids behind its own unique)
            inv_inds = unique_with_plan(pts_cluster_inds)[1]
        else:
            cluster_xyz, _, inv_inds = scatter_v2(center_preds, pts_cluster_inds, mode="avg", return_inv=True)
        if (not torch.is_grad_enabled() and points.is_cuda and points.dtype == torch.float32 and cluster_xyz.dtype == torch.float32
                and hasattr(self.backbone, "_forward_sorted")):
            f_cluster = RowsMinusGroup(points, cluster_xyz, inv_inds)  # (formed by the SIR stack while it permutes its rows)
        else:
            f_cluster = points[:, :3] - gather_by_inverse(cluster_xyz, inv_inds)
        out_pts_feats, cluster_feats, out_coors = self.backbone(points, pts_feats, pts_cluster_inds, f_cluster)
        out_dict = dict(cluster_feats=cluster_feats, cluster_xyz=cluster_xyz, cluster_inds=out_coors)
        if self.as_rpn:
            out_dict["cluster_pts_feats"] = out_pts_feats
            out_dict["cluster_pts_xyz"] = points
        return out_dict

    def update_sample_results_by_mask(self, sampled_out, valid_mask_list):
        # one nonzero() per mask, reused by every field (the reference re-derives it per field via data[mask])
        valid_idx = [m.nonzero(as_tuple=False).squeeze(1) for m in valid_mask_list]
        for k in sampled_out:
            old_data = sampled_out[k]
            if len(old_data[0]) == len(valid_mask_list[0]) or "fg_mask" in k:
                if "fg_mask" in k:
                    new_data_list = []
                    for data, mask in zip(old_data, valid_mask_list):
                        new_data = data.clone()
                        new_data[data] = mask
                        new_data_list.append(new_data)
                    sampled_out[k] = new_data_list
                else:
                    sampled_out[k] = [data.index_select(0, idx) for data, idx in zip(old_data, valid_idx)]
        return sampled_out

    @torch.no_grad()
    def grouped_sample_and_cluster(self, d):
        """Inference: `group_sample` (:802-865) + `ClusterAssigner.forward` (:903-982) + `update_sample_results_by_mask`
        (:867-890) + `combine_classes` (:892-901) for ALL class groups at once.  Upstream (and `sample()` / the assigner
        above, kept for training) walks the six groups one by one — per group a threshold mask, a compaction, a unique, a
        second compaction, a scatter-mean and a connected-components call, each with its own host round trip; here the
        group id rides along as the leading key column, so the whole stage is one compaction, two uniques, one
        connected-components launch and one gather per field.  Rows come out group-major in ascending point order, i.e. in
        the order `combine_classes` concatenates them.
        Returns (points, seg_logits, seg_vote_preds, seg_feats, center_preds, pts_cluster_inds [N,3] = (group, batch, id))."""
        cfg = self.test_cfg
        ca = self.cluster_assigner  # (upstream pairs group i with class_names[i] to index its per-group tables, :912-916)
        batch_idx = d["batch_idx"]
        # (one sample per forward — the reference's test setting — is known on the host: no device round trip for it)
        bsz = 1 if getattr(self, "_batch_size_hint", None) == 1 else int(batch_idx.max().item()) + 1
        seg_logits = d["seg_logits"]
        nc = self.num_classes
        dev = seg_logits.device
        groups, class_names = cfg["group_names"], cfg["class_names"]
        ng = len(groups)
        group_cols = [sorted(class_names.index(n) for n in g) for g in groups]

        def const(key, make):  # per-model constants live on the device once (each upload was a blocking pageable copy per frame)
            cache = self.__dict__.setdefault("_dev_consts", {})
            t = cache.get((key, dev))
            if t is None:
                t = cache[(key, dev)] = make().to(dev)
            return t

        def _member():
            mb = torch.zeros((ng, nc), dtype=torch.bool)
            for gi, cols in enumerate(group_cols):
                mb[gi, cols] = True
            return mb

        member = const(("member", ng, nc), _member)
        scores = seg_logits.softmax(1)[:, :-1]
        thresh = const(("score_thresh", tuple(cfg["score_thresh"])), lambda: torch.tensor(cfg["score_thresh"], dtype=scores.dtype))
        small_groups = max(len(cols) for cols in group_cols) <= 2
        parts = [seg_logits, d["seg_vote_preds"], d["seg_feats"]]
        if (bsz == 1 and getattr(self, "native_cluster_frontend", True) and scores.is_cuda and scores.dtype == torch.float32 and ng <= 32
                and nc <= 32 and small_groups and scores.stride(1) == 1 and seg_logits.dtype == torch.float32
                and all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 for t in parts)
                and d["vote_offsets"].dtype == torch.float32 and d["seg_points"].dtype == torch.float32 and d["seg_points"].stride(1) == 1):
            # K30: everything from the (group, point) pairs to the SIR stack's unique and the cluster centroids — eleven C-ABI entry points,
            # four host waits — as ONE native call that sequences them from C++ (the waits stay, the interpreter between them does not)
            vs_rows = [ca._per_class(ca.cluster_voxel_size, n) for n in ca.class_names[:ng]]
            cells = [max(int(math.ceil((ca.point_cloud_range[3 + a] - ca.point_cloud_range[a]) / row[a])) for row in vs_rows) for a in range(3)]
            dist = const("connected_dist", lambda: torch.tensor([ca._per_class(ca.connected_dist, n) for n in ca.class_names[:ng]],
                                                                dtype=torch.float32))
            res = hip_ops.lidar_cluster_frontend(
                scores, thresh, group_cols, seg_logits, d["vote_offsets"], d["seg_points"], batch_idx, nc, vs_rows,
                ca.point_cloud_range[:3], [0] + [-(c // 2) - 8 for c in cells], [ng - 1] + [c + c // 2 + 8 for c in cells], ca.min_points, dist)
            pts_cluster_inds = res["cluster_inds"]
            with_key_bounds(pts_cluster_inds, [0, 0, 0], [ng - 1, 0, max(res["counts"]["kept_keys"] - 1, 0)])
            seed_unique_cache(pts_cluster_inds, res["new_coors"], res["plan"])  # (extract_feat's scatter_v2 and the SIR stack ask for it)
            self._cluster_xyz_pre = (pts_cluster_inds, res["centers"], res["cluster_xyz"])
            self._grouped_feats_concat = GatheredRows(parts, res["p_ids"])
            return res["points"], None, None, None, res["centers"], pts_cluster_inds
        if (bsz == 1 and scores.is_cuda and scores.dtype == torch.float32 and ng <= 32 and nc <= 32
                and small_groups and scores.stride(1) == 1):
            # K27: the group scores (every group has one or two classes — the nuScenes grouping: their sum has one value whatever
            # adds it), the threshold, "at least one point per group" (:832-834) and the group-major pair list in one C-ABI call
            g_ids, p_ids = hip_ops.group_pairs(scores, thresh, keep_one=True, group_cols=group_cols)
        else:
            if small_groups:
                # a 0/1 membership matmul adds the same one or two scores plus exact zeros — bit-identical to the reference's
                # per-group column sums, one launch instead of 18
                grouped_score = scores @ member.t().to(scores.dtype)
            else:
                # the reference's sums (same columns, same order as its boolean column mask; an index list does not sync the host)
                grouped_score = torch.stack([scores[:, cols].sum(1) for cols in group_cols], dim=1)
            fg = grouped_score > thresh[None, :]
            if bsz == 1:
                fg[0] |= ~fg.any(0)  # "at least one point per sample" (:832-834)
            else:
                for gi in range(ng):
                    if len(torch.unique(batch_idx[fg[:, gi]])) < bsz:
                        fg[self.get_sample_beg_position(batch_idx, fg[:, gi]), gi] = True
            gp = fg.t().nonzero(as_tuple=False)                      # (group, point), group-major
            g_ids, p_ids = gp[:, 0], gp[:, 1]
        # vote centre (the offsets of the group's classes weighted by "is the group's arg-max class", ties split evenly) and the
        # cluster-voxel key (torch.div(.., 'floor') with the group's voxel size, :948-950; group folded into the batch column)
        vs_rows = [ca._per_class(ca.cluster_voxel_size, n) for n in ca.class_names[:ng]]
        if (seg_logits.is_cuda and nc <= 32 and ng <= 16 and seg_logits.dtype == torch.float32
                ):
            # one pass (fsf_vote_centers_keys) instead of ~22 launches over [n_pairs, classes(, 3)] temporaries
            masks = [sum(1 << c for c in cols) for cols in group_cols]
            centers, keys, b_pts = hip_ops.vote_centers_keys(
                seg_logits, d["vote_offsets"], d["seg_points"], batch_idx, g_ids, p_ids, nc, masks, vs_rows,
                ca.point_cloud_range[:3], bsz)
        else:
            logit = seg_logits.index_select(0, p_ids)[:, :nc]
            mem = member.index_select(0, g_ids)
            masked = torch.where(mem, logit, logit.new_full((), float("-inf")))
            w = ((masked - masked.max(1)[0][:, None]).abs() < 1e-6) & mem
            w = w.float()
            w = w / w.sum(1)[:, None]
            offset = d["vote_offsets"].reshape(-1, nc + 1, 3).index_select(0, p_ids)[:, :nc, :]
            centers = d["seg_points"][:, :3].index_select(0, p_ids) + (offset * w[:, :, None]).sum(dim=1)
            vsize = const("cluster_vsize", lambda: torch.tensor(vs_rows, dtype=centers.dtype))
            rmin = const("cluster_rmin", lambda: torch.tensor(ca.point_cloud_range[:3], dtype=centers.dtype))
            vox = torch.div(centers - rmin[None, :], vsize.index_select(0, g_ids), rounding_mode="floor").long()
            b_pts = batch_idx.index_select(0, p_ids).long()
            keys = torch.cat([(g_ids * bsz + b_pts)[:, None], vox], dim=1)
        # (group * batch + sample, voxel of the voted centre): the centres are points of the range moved by a regressed offset, so the
        # cells get the range's extent plus half of it either side; a vote beyond that sends the unique through its range pass
        cells = [max(int(math.ceil((ca.point_cloud_range[3 + a] - ca.point_cloud_range[a]) / row[a])) for row in vs_rows) for a in range(3)]
        with_key_bounds(keys, [0] + [-(c // 2) - 8 for c in cells], [ng * bsz - 1] + [c + c // 2 + 8 for c in cells])
        new_keys, inv, cnt = unique_with_plan(keys)
        # Whether a pair survives is a property of its voxel key (dense enough, or its whole group has no dense voxel, :953-954),
        # so the voxels of the surviving pairs are a SUBSET of the unique just taken: their keys, the pair -> voxel map and the
        # voxel means (same rows in the same order per voxel) follow from it — upstream runs a second unique on the survivors.
        if new_keys.is_cuda and ng <= 64:
            # one C-ABI call and one read-back (K25) instead of ~22 small launches and two nonzero() round trips
            k_idx, vox_group_i32, v_idx, vox_inv = hip_ops.cluster_key_survival(new_keys, cnt, inv, bsz, ca.min_points, ng)
        else:
            vox_group_i32 = None
            key_ok = cnt >= ca.min_points
            key_group = torch.div(new_keys[:, 0], bsz, rounding_mode="floor")           # ascending: keys sort by (group, batch) first
            kcs = torch.cat([key_ok.new_zeros(1, dtype=torch.int64), key_ok.to(torch.int64).cumsum(0)])
            kb = torch.searchsorted(key_group, torch.arange(ng + 1, device=dev))
            has_valid = (kcs[kb[1:]] - kcs[kb[:-1]]) > 0
            key_keep = key_ok | ~has_valid.index_select(0, key_group)
            valid = key_keep.index_select(0, inv)
            v_idx = valid.nonzero(as_tuple=False).squeeze(1)
            k_idx = key_keep.nonzero(as_tuple=False).squeeze(1)
            remap = key_keep.to(torch.int64).cumsum(0) - 1
            vox_inv = remap.index_select(0, inv.index_select(0, v_idx))
        all_means, _, _ = scatter_v2(centers, keys, mode="avg", return_inv=True, unq_inv=inv, new_coors=new_keys,
                                     short_segments=True)
        if vox_group_i32 is not None and all_means.dtype == torch.float32 and centers.dtype == torch.float32:
            # the survivors' rows of all five tensors in one launch (K29b; was five index_selects)
            vox_centers, g_ids, p_ids, b_pts, centers = hip_ops.compact_pairs(all_means, k_idx, g_ids, p_ids, b_pts, centers, v_idx)
            vox_keys = None
        else:
            vox_centers = all_means.index_select(0, k_idx)
            vox_keys = new_keys.index_select(0, k_idx) if vox_group_i32 is None else None  # (only the generic tail below reads them)
            g_ids, p_ids, b_pts = g_ids.index_select(0, v_idx), p_ids.index_select(0, v_idx), b_pts.index_select(0, v_idx)
            centers = centers.index_select(0, v_idx)
        dist = const("connected_dist", lambda: torch.tensor([ca._per_class(ca.connected_dist, n) for n in ca.class_names[:ng]],
                                                            dtype=torch.float32))
        # test-time clustering ignores the sample index inside a group (:69-82); components never span groups
        if vox_group_i32 is not None:
            labels = hip_ops.connected_components_grouped(vox_centers, vox_group_i32, dist)
            # (a group's labels start at its first voxel's: renumbered from 0 per group and mapped to the pairs in two launches)
            pts_cluster_inds = hip_ops.cluster_point_ids(labels, vox_group_i32, vox_inv, g_ids, b_pts, ng)
            # (group, sample, cluster id within the group): a cluster holds at least one of the kept voxels
            with_key_bounds(pts_cluster_inds, [0, 0, 0], [ng - 1, bsz - 1, max(int(labels.numel()) - 1, 0)])
        else:
            vox_group = torch.div(vox_keys[:, 0], bsz, rounding_mode="floor")
            labels = hip_ops.connected_components_grouped(vox_centers, vox_group, dist).long()
            first = torch.searchsorted(vox_group, torch.arange(ng, device=dev))            # voxels are group-sorted
            base = labels[first.clamp(max=labels.numel() - 1)]                              # a group's labels start at its first voxel's
            cluster = (labels - base.index_select(0, vox_group)).index_select(0, vox_inv)
            pts_cluster_inds = torch.stack([g_ids, b_pts, cluster], 1)
        def take(t):  # rows p_ids of t (ATen's index_select is slow on narrow float rows: 118 us for [510 k, 4])
            if t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1:
                return hip_ops.gather_rows(t, p_ids)
            return t.index_select(0, p_ids)

        parts = [seg_logits, d["seg_vote_preds"], d["seg_feats"]]
        if all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 for t in parts):
            # the caller concatenates the three (FSF.py fsd_forward) and hands the result to the first SIR layer only: the rows stay
            # where they are, the layer's input kernel reads them through `p_ids` side by side (sst_ops.GatheredRows; materialised —
            # gathered straight into one [n, 11 + 33 + 131] buffer — by anything else that wants the matrix)
            lazy = GatheredRows(parts, p_ids)
            self._grouped_feats_concat = lazy
            return take(d["seg_points"]), None, None, None, centers, pts_cluster_inds
        self._grouped_feats_concat = None
        return (take(d["seg_points"]), take(seg_logits), take(d["seg_vote_preds"]), take(d["seg_feats"]), centers,
                pts_cluster_inds)

    def combine_classes(self, data_dict, name_list):
        return {name: torch.cat(data_dict[name], 0) for name in data_dict if name in name_list}

    def pre_voxel_keys(self, points, batch_idx, bsz):
        """:585-591 — the (batch, z, y, x) div-floor keys of `pre_voxelize`, with the bounds the unique packs its sort key from."""
        coors = hip_ops.voxelize_divfloor(points, self.cfg["pre_voxelization_size"],
                                          self.cluster_assigner.point_cloud_range[:3], order="zyx", batch_idx=batch_idx)
        if bsz is not None:  # (batch, z, y, x) cells of the cluster assigner's range, a few cells of slack either side
            rng, vs = self.cluster_assigner.point_cloud_range, self.cfg["pre_voxelization_size"]
            cells = [int(math.ceil((rng[3 + a] - rng[a]) / vs[a])) for a in (2, 1, 0)]
            with_key_bounds(coors, [0, -8, -8, -8], [bsz - 1] + [c + 8 for c in cells])
        return coors

    def pre_voxelize(self, data_dict):
        """:585-605 — torch.div-floor 0.1 m keys (zyx + batch), ONE unique, mean of every float field."""
        batch_idx = data_dict["batch_idx"]
        points = data_dict["seg_points"]
        pre = self.__dict__.pop("_pre_vox", None)
        if (pre is not None and pre["points"] is points and batch_idx.data_ptr() == pre["batch_ptr"] and batch_idx.dim() == 1
                and batch_idx.size(0) == points.size(0) and not torch.is_grad_enabled()):
            # (FSF._pre_voxel_keys_early: the same keys and their unique, formed on the front stream while the segmentor ran)
            torch.cuda.current_stream().wait_event(pre["event"])
            coors, (new_coors, unq_inv, _) = pre["keys"], pre["res"]
            seed_unique_result(coors, pre["res"])
        else:
            coors = self.pre_voxel_keys(points, batch_idx, getattr(self, "_batch_size_hint", None))
            new_coors, unq_inv, _ = unique_with_plan(coors)
        # (upstream: one scatter_v2(.., mode='avg') per float field over the shared unique; here the fields go through one launch)
        names = [name for name, data in data_dict.items() if data.dtype in (torch.float, torch.float16)]
        voxelized = dict(zip(names, scatter_mean_multi([data_dict[n] for n in names], new_coors, unq_inv)))
        voxel_coors = new_coors
        voxelized["batch_idx"] = voxel_coors[:, 0]
        return voxelized

    def get_sample_beg_position(self, batch_idx, fg_mask):
        assert batch_idx.shape == fg_mask.shape
        inner_inds = get_inner_win_inds(batch_idx.contiguous())
        return torch.where(inner_inds == 0)[0]

    def get_fg_mask(self, seg_scores, seg_points, cls_id, batch_inds, gt_bboxes_3d, gt_labels_3d):
        """(:740-782) score threshold (+ the train-time `threshold_buffer`) or, while detection is still disabled in
        pre-training, the top-k points; the ground-truth-box augmentation (`add_gt_fg_points`) needs box geometry ops
        that are outside the hot path."""
        train_cfg = getattr(self, "train_cfg", None) or {}
        runtime_info = getattr(self, "runtime_info", None) or {}
        seg_scores = seg_scores[:, cls_id]
        if self.training and train_cfg.get("disable_pretrain", False) and not runtime_info.get("enable_detection", False):
            k = min(train_cfg.get("disable_pretrain_topks", [100, 100, 100])[cls_id], len(seg_scores))
            fg_mask = torch.zeros_like(seg_scores, dtype=torch.bool)
            fg_mask[torch.topk(seg_scores, k)[1]] = True
        else:
            buffer_thr = runtime_info.get("threshold_buffer", 0) if self.training else 0
            fg_mask = seg_scores > self.cfg["score_thresh"][cls_id] + buffer_thr
        cfg = train_cfg if self.training else (getattr(self, "test_cfg", None) or {})
        if cfg.get("add_gt_fg_points", False):
            raise NotImplementedError("add_gt_fg_points (points_in_boxes on the GT boxes) is outside the hot path")
        return fg_mask

    def gather_group_by_names(self, scores):
        groups, class_names = self.cfg["group_names"], self.cfg["class_names"]
        assert (scores >= 0).all()
        return torch.stack([scores[:, [class_names.index(n) for n in g]].sum(1) for g in groups], dim=1)

    def get_offset_weight(self, seg_logit):
        if self.cfg["offset_weight"] != "max":
            raise NotImplementedError
        weight = ((seg_logit - seg_logit.max(1)[0][:, None]).abs() < 1e-6).float()
        return weight / weight.sum(1)[:, None]  # ties split evenly (the row maximum always has weight 1)

    def sample(self, dict_to_sample, offset, gt_bboxes_3d=None, gt_labels_3d=None):
        if self.cfg.get("group_sample", False):
            return self.group_sample(dict_to_sample, offset)
        raise NotImplementedError("per-class sampling (:682-730) is not used by the FSF configs")

    def group_sample(self, dict_to_sample, offset):
        """:802-865."""
        batch_idx = dict_to_sample["batch_idx"]
        bsz = int(batch_idx.max().item()) + 1
        cfg = self.train_cfg if self.training else self.test_cfg
        seg_logits = dict_to_sample["seg_logits"]
        assert seg_logits.size(1) == self.num_classes + 1  # (the reference also host-syncs on `(seg_logits < 0).any()`)
        seg_scores = seg_logits.softmax(1)
        offset = offset.reshape(-1, self.num_classes + 1, 3)
        seg_points = dict_to_sample["seg_points"][:, :3]
        fg_mask_list, fg_idx_list, center_preds_list = [], [], []
        cls_score_thrs, group_names, class_names = cfg["score_thresh"], cfg["group_names"], cfg["class_names"]
        assert len(group_names) == len(cls_score_thrs)
        grouped_score = self.gather_group_by_names(seg_scores[:, :-1])
        for i in range(len(group_names)):
            fg_mask = self.get_fg_mask(grouped_score, None, i, None, None, None)
            if bsz == 1:
                fg_mask[0] |= ~fg_mask.any()  # "at least one point per sample" (:832-834) without the unique() sync
            elif len(torch.unique(batch_idx[fg_mask])) < bsz:
                fg_mask[self.get_sample_beg_position(batch_idx, fg_mask)] = True
            fg_mask_list.append(fg_mask)
            fg_idx = fg_mask.nonzero(as_tuple=False).squeeze(1)  # one compaction per group, reused for every field
            fg_idx_list.append(fg_idx)
            tmp_idx = [class_names.index(name) for name in group_names[i]]
            this_offset = offset.index_select(0, fg_idx)[:, tmp_idx, :]
            this_logits = seg_logits.index_select(0, fg_idx)[:, tmp_idx]
            offset_weight = self.get_offset_weight(this_logits)
            this_offset = (this_offset * offset_weight[:, :, None]).sum(dim=1)
            center_preds_list.append(seg_points.index_select(0, fg_idx) + this_offset)
        output_dict = {name: [data.index_select(0, idx) for idx in fg_idx_list] for name, data in dict_to_sample.items()}
        output_dict["fg_mask_list"] = fg_mask_list
        output_dict["center_preds"] = center_preds_list
        return output_dict


class ClusterAssigner(nn.Module):
    """:903-982 — per class group: divfloor voxel keys, drop almost-empty voxels, voxel centroids, connected
    components of the centroids, map component ids back to the points."""

    def __init__(self, cluster_voxel_size, min_points, point_cloud_range, connected_dist,
                 class_names=["Car", "Cyclist", "Pedestrian"], gpu_clustering=(False, False)):
        super().__init__()
        self.cluster_voxel_size = cluster_voxel_size
        self.min_points = min_points
        self.connected_dist = connected_dist
        self.point_cloud_range = point_cloud_range
        self.class_names = class_names
        self.gpu_clustering = gpu_clustering  # kept for config parity; clustering always runs on the device here

    def _per_class(self, table, class_name):
        if isinstance(table, dict):
            return table[class_name]
        if isinstance(table, list):
            return table[self.class_names.index(class_name)]
        return table

    @torch.no_grad()
    def forward(self, points_list, batch_idx_list, gt_bboxes_3d=None, gt_labels_3d=None, origin_points=None):
        assert self.num_classes == len(self.class_names)
        outs = [self.forward_single_class(p, b, n, o)
                for p, b, n, o in zip(points_list, batch_idx_list, self.class_names, origin_points)]
        cluster_inds_list = modify_cluster_by_class([o[0] for o in outs])
        return cluster_inds_list, [o[1] for o in outs]

    def forward_single_class(self, points, batch_idx, class_name, origin_points):
        batch_idx = batch_idx.int()
        cluster_vsize = self._per_class(self.cluster_voxel_size, class_name)
        coors = hip_ops.voxelize_divfloor(points, cluster_vsize, self.point_cloud_range[:3], order="xyz",
                                          batch_idx=batch_idx.long())
        valid_mask = filter_almost_empty(coors, min_points=self.min_points)
        valid_idx = valid_mask.nonzero(as_tuple=False).squeeze(1)
        if valid_idx.numel() == 0:
            valid_mask = ~valid_mask
            valid_idx = valid_mask.nonzero(as_tuple=False).squeeze(1)
        points, batch_idx, coors = points.index_select(0, valid_idx), batch_idx.index_select(0, valid_idx), \
            coors.index_select(0, valid_idx)
        sampled_centers, voxel_coors, inv_inds = scatter_v2(points, coors, mode="avg", return_inv=True)
        dist = self._per_class(self.connected_dist, class_name)
        if self.training:
            cluster_inds = find_connected_componets(sampled_centers, voxel_coors[:, 0].int(), dist)
        else:
            cluster_inds = find_connected_componets_single_batch(sampled_centers, voxel_coors[:, 0], dist)
        assert len(cluster_inds) == len(sampled_centers)
        cluster_inds_per_point = cluster_inds[inv_inds]
        return torch.stack([batch_idx, cluster_inds_per_point], 1), valid_mask
