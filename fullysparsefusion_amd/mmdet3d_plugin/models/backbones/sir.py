"""`SIR` (BACKBONES): a stack of SIRLayer blocks over (points, feats, group ids).
Mirror of projects/mmdet3d_plugin/models/backbones/sir.py:13-85 — same constructor, same block wiring, same
return triple; the single `torch.unique` of the stack (:68) is the HIP packed-key sort whose segment plan all
2 x num_blocks segmented max reductions and gathers of the stack reuse."""
import torch
import torch.nn as nn

from .... import hip_ops, switches
from ...ops.sst_ops import GatheredRows, RowsMinusGroup, plan_of, sir_stack_descriptor, unique_with_plan
from ...registry import BACKBONES, build_voxel_encoder


@BACKBONES.register_module()
class SIR(nn.Module):
    def __init__(self, num_blocks=5, in_channels=[], feat_channels=[], rel_mlp_hidden_dims=[], with_rel_mlp=True,
                 with_distance=False, with_cluster_center=False, norm_cfg=dict(type="LN", eps=1e-3), mode="max",
                 xyz_normalizer=[1.0, 1.0, 1.0], act="relu", dropout=0, unique_once=False):
        super().__init__()
        self.num_blocks = num_blocks
        self.unique_once = unique_once
        blocks = []
        for i in range(num_blocks):
            blocks.append(build_voxel_encoder(dict(
                type="SIRLayer",
                in_channels=in_channels[i],
                feat_channels=feat_channels[i],
                with_distance=with_distance,
                with_cluster_center=with_cluster_center,
                with_rel_mlp=with_rel_mlp,
                rel_mlp_hidden_dims=rel_mlp_hidden_dims[i],
                with_voxel_center=False,
                voxel_size=[0.1, 0.1, 0.1],  # unused by SIRLayer, kept for interface parity (sir.py:48-49)
                point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4],
                norm_cfg=norm_cfg,
                mode=mode,
                fusion_layer=None,
                return_point_feats=i != num_blocks - 1,
                return_inv=False,
                rel_dist_scaler=10.0,
                xyz_normalizer=xyz_normalizer,
                act=act,
                dropout=dropout,
            )))
        self.block_list = nn.ModuleList(blocks)
        # Does the caller read the per-point features this stack returns first?  FSF / SingleStageFSD only consume the group
        # features (FSF.py:436-447, single_stage_fsd.py:468-474) and say so; then the sorted path neither writes the last layer's
        # rows nor un-sorts them, and the first return value is None.
        self.point_feats_needed = True
        self.native_stack = True  # the sorted stack as one native call (K31); False: one C-ABI call per kernel (the tests' reference)

    def _forward_sorted(self, points, features, coors, f_cluster):
        """Inference: the whole stack on rows SORTED by group (one permutation in, through the indices K21 reads its sources by):
        a group's rows are then one contiguous run, and every `Linear -> LN -> GELU -> scatter_v2(max)` pair of sir.py:65-85 /
        SIRLayer is ONE kernel (K22s) — the [n, 128] activations are read back by nothing but the next layer, the twelve
        segmented-max launches and their gathers through `order` are gone, and the [g, 768] group features are written in place."""
        new_coors, unq_inv, _ = unique_with_plan(coors)
        m = new_coors.size(0)
        plan = plan_of(unq_inv, m)
        widths = [b.group_width() for b in self.block_list]
        groups = torch.empty((m, sum(widths)), dtype=torch.float32, device=points.device)
        lazy = isinstance(f_cluster, RowsMinusGroup)
        if lazy and not (f_cluster.inv is unq_inv and f_cluster.points is points and f_cluster.centers.size(0) == m):
            f_cluster, lazy = f_cluster.materialize(), False  # (an offset to some other grouping: the expression itself)
        if isinstance(features, GatheredRows) and features.direct:
            features = features.materialize()  # (a part that is not read through the index cannot follow the permutation)
        gathered = isinstance(features, GatheredRows)
        # ONE launch (K29a): the group id, the point rows, the centre offsets and the feature-row index of every sorted position,
        # and the -inf the group table starts with (was: .long(), two index_selects, two row gathers, a fill)
        seg_ids, pts_s, fcl_s, idx_s = hip_ops.sorted_rows(
            plan.order, plan.inv, points, f_cluster=None if lazy else f_cluster, centers=f_cluster.centers if lazy else None,
            index=features.index if gathered else None, fill=groups)
        feats = GatheredRows(features.sources if gathered else [features], idx_s)
        desc = sir_stack_descriptor(self, self.block_list) if self.native_stack and m >= 16 and len(feats.sources) <= 3 else None
        if desc is not None:
            # K31: the blocks below as ONE native call (fsf_sir_stack_forward: the same entry points with the same arguments, sequenced
            # from C++ — 13 C-ABI calls and their interpreter time per stack; bit-identical, tests/test_sir_stack_gpu.py)
            rows = hip_ops.sir_stack_forward(desc, pts_s, feats.sources, fcl_s, seg_ids, groups, self.point_feats_needed,
                                             feats_index=feats.index, direct_parts=feats.direct)
        else:
            col, rows = 0, None
            for i, block in enumerate(self.block_list):
                want = i < self.num_blocks - 1 or self.point_feats_needed
                rows = block.forward_sorted(pts_s, feats, fcl_s, seg_ids, groups[:, col:col + widths[i]], want)
                feats = rows
                col += widths[i]
        out_feats = None
        if self.point_feats_needed:
            out_feats = torch.empty_like(rows)
            out_feats.index_copy_(0, plan.order.long(), rows)
        return out_feats, groups, new_coors

    def forward(self, points, features, coors, f_cluster=None):
        if (switches.SIR_SORTED and self.unique_once and f_cluster is not None and not torch.is_grad_enabled() and points.is_cuda
                and points.dtype == torch.float32 and points.size(0) > 0 and f_cluster.dtype == torch.float32
                and (isinstance(features, GatheredRows) or (features.dtype == torch.float32 and features.stride(1) == 1))
                and all(getattr(b, "sorted_supported", lambda: False)() for b in self.block_list)):
            return self._forward_sorted(points, features, coors, f_cluster)
        if isinstance(f_cluster, RowsMinusGroup):
            f_cluster = f_cluster.materialize()
        if self.unique_once:
            new_coors, unq_inv, _ = unique_with_plan(coors)
        else:
            new_coors = unq_inv = None
        out_feats = features
        cluster_feat_list = []
        out_coors = None
        for i, block in enumerate(self.block_list):
            # block(torch.cat([points, out_feats], 1), coors, f_cluster, ...); SIRLayer folds the concat into its input kernel
            run = getattr(block, "forward_parts", None)
            if run is None:
                run = lambda p, f, *a, _b=block, **k: _b(torch.cat([p, f], 1), *a, **k)  # noqa: E731
            if i < self.num_blocks - 1:
                out_feats, out_cluster_feats = run(points, out_feats, coors, f_cluster, unq_inv_once=unq_inv,
                                                   new_coors_once=new_coors)
            else:
                out_feats, out_cluster_feats, out_coors = run(points, out_feats, coors, f_cluster, return_both=True,
                                                              unq_inv_once=unq_inv, new_coors_once=new_coors)
            cluster_feat_list.append(out_cluster_feats)
        final_cluster_feats = torch.cat(cluster_feat_list, dim=1)
        return out_feats, final_cluster_feats, out_coors
