"""Refine stage: `DynamicPointROIExtractor`
(projects/mmdet3d_plugin/models/roi_heads/roi_extractors/dynamic_point_roi_extractor.py:9-100) and
`FullySparseBboxHead` (models/roi_heads/bbox_heads/fsd_bbox_head.py:22-197), inference path.

The extractor upstream loops over the samples of the batch, calls the TorchEx kernel per sample and offsets the
indices; here the batch index travels into ONE launch of K17 (`fsf_dynamic_point_pool`), whose rows come back already
in ascending (roi, point) order — i.e. sorted by the RoI index the SIR layers group on.
"""
import torch
import torch.nn as nn

from .... import hip_ops, switches
from ...ops.dynamic_point_pool_op import dynamic_point_pool
from ...ops.sst_ops import GatheredRows, sir_stack_descriptor, unique_with_plan, with_key_bounds
from ...registry import HEADS, ROI_EXTRACTORS, build_voxel_encoder


@ROI_EXTRACTORS.register_module()
class DynamicPointROIExtractor(nn.Module):
    def __init__(self, init_cfg=None, debug=True, extra_wlh=[0, 0, 0], max_inbox_point=512, max_all_pts=50000):
        super().__init__()
        self.debug = debug
        self.extra_wlh = extra_wlh
        self.max_inbox_point = max_inbox_point
        self.max_all_pts = max_all_pts  # upstream: the default of DynamicPointPoolFunction.forward, PER SAMPLE

    def forward(self, pts_xyz, batch_inds, rois):
        """pts_xyz [P,3], batch_inds [P] (sorted), rois [R,8] (batch, x, y, z_bottom, w, l, h, rz) ->
        (point indices [k], roi indices [k], dict(local_xyz [k,3], boundary_offset [k,6], is_in_margin [k]))."""
        assert len(pts_xyz) > 0 and len(batch_inds) > 0 and len(rois) > 0
        rois = rois.float()
        if rois.size(1) == 7:  # one sample: the reference's op itself (ops/dynamic_point_pool_op.py), fake row included
            inds, roi_inds, info = dynamic_point_pool(rois, pts_xyz.float(), self.extra_wlh, self.max_inbox_point,
                                                      self.max_all_pts)
        else:  # the per-sample loop of :44-73 as ONE launch: the batch index travels into the kernel
            inds, roi_inds, info = hip_ops.dynamic_point_pool(
                rois, pts_xyz.float(), self.extra_wlh, self.max_inbox_point, self.max_all_pts,
                roi_batch_col=0, box_col=1, pts_batch=batch_inds)
            real = inds.numel() > 0
            if not real:  # upstream fakes one (-1, -1, zeros) row so that downstream shapes stay non-empty
                inds = inds.new_full((1,), -1)
                roi_inds = roi_inds.new_full((1,), -1)
                info = info.new_zeros((1, 13))
            roi_inds._fsf_real_rows = real  # (known on the host: every roi index is then >= 0)
        if self.debug and inds[0] >= 0:
            roi_per_pts = rois[:, -7:][roi_inds]
            assert torch.isclose(pts_xyz[inds], info[:, :3]).all()
            assert torch.isclose(info[:, 6] + info[:, 9], roi_per_pts[:, 4], atol=1e-4).all()
            assert torch.isclose(info[:, 7] + info[:, 10], roi_per_pts[:, 3], atol=1e-4).all()
            assert torch.isclose(info[:, 8] + info[:, 11], roi_per_pts[:, 5], atol=1e-4).all()
        ext_pts_info = dict(local_xyz=info[:, 3:6], boundary_offset=info[:, 6:-1], is_in_margin=info[:, -1])
        if info.size(1) == 13:
            ext_pts_info["_fsf_info13"] = info  # (the three views' one tensor: FSF.query_feat_refine hands it to K29e)
        roi_inds._fsf_sorted = True  # K17's rows come in ascending (roi, point) order: FullySparseBboxHead's groups are contiguous runs
        return inds, roi_inds, ext_pts_info


@HEADS.register_module()
class FullySparseBboxHead(nn.Module):
    """Three `DynamicClusterVFE` (SIR-layer) blocks over the points pooled into each RoI; returns one feature row per
    RoI (zeros for RoIs without points) and the non-empty mask."""

    def __init__(self, num_classes, num_blocks, in_channels, feat_channels, with_distance, with_cluster_center,
                 with_rel_mlp, rel_mlp_hidden_dims, rel_mlp_in_channels, reg_mlp, cls_mlp, mode="max",
                 xyz_normalizer=[20, 20, 4], cat_voxel_feats=True, pos_fusion="mul", fusion="cat", act="gelu",
                 geo_input=True, use_middle_cluster_feature=True, norm_cfg=dict(type="LN", eps=1e-3, momentum=0.01),
                 dropout=0, unique_once=False, init_cfg=None, no_head=None):
        super().__init__()
        self.num_classes = num_classes
        self.geo_input = geo_input
        self.num_blocks = num_blocks
        self.use_middle_cluster_feature = use_middle_cluster_feature
        self.print_info = {}
        self.unique_once = unique_once
        blocks = []
        for i in range(num_blocks):
            blocks.append(build_voxel_encoder(dict(
                type="DynamicClusterVFE", in_channels=in_channels[i], feat_channels=feat_channels[i],
                with_distance=with_distance, with_cluster_center=with_cluster_center, with_rel_mlp=with_rel_mlp,
                rel_mlp_hidden_dims=rel_mlp_hidden_dims[i], rel_mlp_in_channel=rel_mlp_in_channels[i],
                with_voxel_center=False, voxel_size=[0.1, 0.1, 0.1], point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4],
                norm_cfg=norm_cfg, mode=mode, fusion_layer=None, return_point_feats=i != num_blocks - 1, return_inv=False,
                rel_dist_scaler=10.0, fusion=fusion, pos_fusion=pos_fusion, xyz_normalizer=xyz_normalizer,
                cat_voxel_feats=cat_voxel_feats, act=act, dropout=dropout)))
        self.block_list = nn.ModuleList(blocks)

    takes_gathered_rows = True  # (forward accepts sst_ops.GatheredRows as `pts_features`)

    def forward(self, pts_xyz, pts_features, pts_info, roi_inds, rois):
        assert pts_features.size(0) > 0
        rois = rois[:, 1:]
        lazy_feats = isinstance(pts_features, GatheredRows)
        sorted_ok = (switches.SIR_SORTED and self.unique_once and getattr(roi_inds, "_fsf_sorted", False)
                     and self.use_middle_cluster_feature and not torch.is_grad_enabled() and pts_xyz.is_cuda
                     and pts_xyz.dtype == torch.float32 and pts_features.dtype == torch.float32
                     and (lazy_feats or pts_features.stride(1) == 1)
                     and all(getattr(b, "sorted_supported", lambda: False)() for b in self.block_list))
        direct = sorted_ok and switches.REFINE_DIRECT and getattr(roi_inds, "_fsf_real_rows", False) and roi_inds.dtype == torch.int64
        if lazy_feats and not direct:
            pts_features = pts_features.materialize()  # (only the RoI-indexed path below reads the parts in place)
        if direct:
            # The pooled rows are sorted by RoI and every index is a real RoI: the RoI index IS the segment id and the
            # [rois, 768] result IS the group table — no unique (12 launches, a host wait), no scatter of the groups to their RoI
            # rows (11 ATen launches).  A RoI without points keeps the -inf the table starts with: that is the non-empty mask,
            # and its row becomes the zeros upstream returns (align_roi_feature_and_rois, :153-165).
            num_rois = len(rois)
            f_cluster = pts_info.get("_fsf_f_cluster")  # (K29e formed it with the pooled rows)
            if f_cluster is None:
                rel_xyz = pts_xyz[:, :3] - rois[:, :3][roi_inds]
                f_cluster = torch.cat([pts_info["local_xyz"], pts_info["boundary_offset"], pts_info["is_in_margin"][:, None], rel_xyz],
                                      dim=-1)
            widths = [b.group_width() for b in self.block_list]
            groups = torch.full((num_rois, sum(widths)), float("-inf"), dtype=torch.float32, device=pts_xyz.device)
            seg_ids = roi_inds.contiguous()
            srcs = pts_features.sources if lazy_feats else [pts_features]
            desc = (sir_stack_descriptor(self, self.block_list)
                    if getattr(self, "native_stack", True) and num_rois >= 16 and len(srcs) <= 3 and pts_xyz.size(0) > 0 else None)
            if desc is not None:  # K31: the three blocks as one native call (see SIR._forward_sorted)
                hip_ops.sir_stack_forward(desc, pts_xyz, srcs, f_cluster, seg_ids, groups, False, extra=f_cluster if self.geo_input else None,
                                          extra_div=10.0, feats_index=pts_features.index if lazy_feats else None,
                                          direct_parts=pts_features.direct if lazy_feats else ())
            else:
                out_feats, col = pts_features, 0
                for i, block in enumerate(self.block_list):
                    out_feats = block.forward_sorted(pts_xyz, out_feats, f_cluster, seg_ids, groups[:, col:col + widths[i]],
                                                     i < self.num_blocks - 1, extra=f_cluster if self.geo_input else None, extra_div=10.0)
                    col += widths[i]
            nonempty = groups[:, 0] > float("-inf")
            return torch.where(nonempty[:, None], groups, groups.new_zeros(())), nonempty
        rel_xyz = pts_xyz[:, :3] - rois[:, :3][roi_inds]
        coors = roi_inds.unsqueeze(1)  # the segment machinery takes key ROWS; upstream groups on the 1-D index
        with_key_bounds(coors, [0], [max(rois.size(0) - 1, 0)])
        if self.unique_once:  # torch.unique(roi_inds, return_inverse=True) upstream (:114-115), with the segment plan
            new_coors, unq_inv, _ = unique_with_plan(coors)
        else:
            new_coors = unq_inv = None
        out_feats = pts_features
        f_cluster = torch.cat([pts_info["local_xyz"], pts_info["boundary_offset"], pts_info["is_in_margin"][:, None], rel_xyz],
                              dim=-1)
        if sorted_ok and unq_inv is not None:
            # the pooled rows are sorted by RoI already: every Linear -> LN -> GELU -> max pair of the three blocks is one K22s launch,
            # the [rois, 768] group features are written in place (see SIR._forward_sorted)
            m = new_coors.size(0)
            widths = [b.group_width() for b in self.block_list]
            groups = torch.full((m, sum(widths)), float("-inf"), dtype=torch.float32, device=pts_xyz.device)
            col = 0
            for i, block in enumerate(self.block_list):
                out_feats = block.forward_sorted(pts_xyz, out_feats, f_cluster, unq_inv, groups[:, col:col + widths[i]],
                                                 i < self.num_blocks - 1, extra=f_cluster if self.geo_input else None, extra_div=10.0)
                col += widths[i]
            out_coors = new_coors.squeeze(1)
            return self.align_roi_feature_and_rois(groups, out_coors, len(rois)), self.get_nonempty_roi_mask(out_coors, len(rois))
        cluster_feat_list = []
        for i, block in enumerate(self.block_list):
            # in_feats = cat([pts_xyz, out_feats(, f_cluster / 10)], 1) (:127-132), folded into the block's input kernel
            geo = dict(extra=f_cluster, extra_div=10.0) if self.geo_input else {}
            if i < self.num_blocks - 1:
                out_feats, out_cluster_feats = block.forward_parts(pts_xyz, out_feats, coors, f_cluster, unq_inv_once=unq_inv,
                                                                   new_coors_once=new_coors, **geo)
                if self.use_middle_cluster_feature:
                    cluster_feat_list.append(out_cluster_feats)
            else:
                out_cluster_feats, out_coors = block.forward_parts(pts_xyz, out_feats, coors, f_cluster, unq_inv_once=unq_inv,
                                                                   new_coors_once=new_coors, **geo)
                cluster_feat_list.append(out_cluster_feats)
        final_cluster_feats = torch.cat(cluster_feat_list, dim=1)
        out_coors = out_coors.squeeze(1)
        nonempty_roi_mask = self.get_nonempty_roi_mask(out_coors, len(rois))
        return self.align_roi_feature_and_rois(final_cluster_feats, out_coors, len(rois)), nonempty_roi_mask

    def get_nonempty_roi_mask(self, out_coors, num_rois):
        # (the -1 group of an empty pooling result goes to a spare slot instead of being filtered out with a boolean index:
        # no device -> host round trip for the survivor count)
        mask = torch.zeros(num_rois + 1, dtype=torch.bool, device=out_coors.device)
        mask[torch.where(out_coors >= 0, out_coors, out_coors.new_full((), num_rois))] = True
        return mask[:num_rois]

    def align_roi_feature_and_rois(self, features, out_coors, num_rois):
        """Group features come out in ascending RoI index with a possible leading -1 group (the fake row of an empty
        pooling result); scatter them to one row per RoI."""
        coors_mask = out_coors >= 0
        if not (torch.is_grad_enabled() and features.requires_grad):
            # inference: rows of the -1 group land in a spare row that is cut off (same result, no host syncs)
            new_feature = features.new_zeros((num_rois + 1, features.size(1)))
            new_feature.index_copy_(0, torch.where(coors_mask, out_coors, out_coors.new_full((), num_rois)), features)
            return new_feature[:num_rois]
        new_feature = features.new_zeros((num_rois, features.size(1)))
        if not coors_mask.any():
            new_feature[:len(features), :] = features * 0  # pseudo gradient, as upstream
            return new_feature
        new_feature[out_coors[coors_mask]] = features[coors_mask]
        return new_feature
