"""`DynamicScatterVFE` and `SIRLayer` (VOXEL_ENCODERS) — defined in the authors' mmdet3d fork [UNVENDORED];
selected at projects/configs/nuScenes/FSF_nuScenes_config.py:42-52 and built by
projects/mmdet3d_plugin/models/backbones/sir.py:41-61.  Restated from the published SST/FSD modules
(SURVEY.md App. C); module / parameter names follow upstream so state dicts keep their keys
(`vfe_layers.N.linear.weight`, `vfe_layers.N.norm.*`, `rel_mlp.*`).

Per call: ONE packed-key radix sort (unique_once), then every segmented mean/max and every "map back to the
points" gather reuses that sort-once segment plan through the HIP library.
"""
from .... import switches
import os

import torch
import torch.nn as nn

from .... import hip_ops
from ...ops.sst_ops import (build_mlp, fused_norm_act, gather_by_inverse, get_activation_layer, linear_norm_act,
                            GatheredRows, GroupedConcat, point_group_concat, point_linear, scatter_v2, sorted_stack_forward,
                            sorted_stack_supported, unique_with_plan)
from ...registry import VOXEL_ENCODERS, build_norm_layer


class DynamicVFELayer(nn.Module):
    """Linear(bias=False) -> norm -> act (-> dropout); upstream `DynamicVFELayer`."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), act="relu",
                 dropout=0.0):
        super().__init__()
        self.fp16_enabled = False
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = nn.Linear(in_channels, out_channels, bias=False)
        self.act = get_activation_layer(act, out_channels)
        self.dropout = nn.Dropout(dropout) if dropout > 0 else None

    def forward(self, inputs):
        x = linear_norm_act(self.linear, self.norm, self.act, inputs)
        if self.dropout is not None:
            x = self.dropout(x)
        return x


@VOXEL_ENCODERS.register_module()
class DynamicScatterVFE(nn.Module):
    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), mode="max", fusion_layer=None,
                 return_point_feats=False, unique_once=False):
        super().__init__()
        assert mode in ("avg", "max") and len(feat_channels) > 0 and fusion_layer is None
        if with_cluster_center:
            in_channels += 3
        if with_voxel_center:
            in_channels += 3
        if with_distance:
            in_channels += 1
        self.in_channels = in_channels
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.return_point_feats = return_point_feats
        self.unique_once = unique_once
        self.fp16_enabled = False
        self.vx, self.vy, self.vz = voxel_size
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.point_cloud_range = list(point_cloud_range)
        self.voxel_size = list(voxel_size)
        self.mode = mode
        chans = [self.in_channels] + list(feat_channels)
        layers = []
        for i in range(len(chans) - 1):
            in_filters = chans[i] * 2 if i > 0 else chans[i]
            layers.append(DynamicVFELayer(in_filters, chans[i + 1], norm_cfg))
        self.vfe_layers = nn.ModuleList(layers)
        self.num_vfe = len(layers)
        gx = round((point_cloud_range[3] - point_cloud_range[0]) / voxel_size[0])
        gy = round((point_cloud_range[4] - point_cloud_range[1]) / voxel_size[1])
        gz = round((point_cloud_range[5] - point_cloud_range[2]) / voxel_size[2])
        self._grid_zyx = (gz, gy, gx)

    def _key_bounds(self, coors):
        """Known voxel-grid bounds let the unique kernel skip the device min/max pass + host sync."""
        if coors.size(1) != 4:
            return None, None
        gz, gy, gx = self._grid_zyx
        return [0, -1, -1, -1], [max(int(getattr(self, "max_batch", 64)) - 1, 0), gz - 1, gy - 1, gx - 1]

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None, return_inv=False):
        if self.unique_once:
            cmin, cmax = self._key_bounds(coors)
            new_coors, unq_inv, _ = unique_with_plan(coors, cmin, cmax)
            if new_coors.is_cuda and not torch.is_grad_enabled():
                # the voxel coordinates exist from here on: a consumer that only needs THEM (SimpleSparseUNet's index plan) may start
                # on another stream while this stream still runs the VFE layers
                ev = torch.cuda.Event()
                ev.record()
                new_coors._fsf_ready_event = ev
        else:
            new_coors = unq_inv = None
        if (features.is_cuda and features.dtype == torch.float32 and not self._with_distance and features.size(1) >= 3
                and (self._with_cluster_center or self._with_voxel_center)
                and not (torch.is_grad_enabled() and features.requires_grad)):
            # inference: the decorated input in one pass (fsf_vfe_decorate) instead of a gather, a dozen elementwise launches and
            # a cat; its rows are padded to 16 bytes, so the first layer's fused Linear reads them in place
            voxel_mean = unq_inv_c = None
            if self._with_cluster_center:
                voxel_mean, _, unq_inv_c = scatter_v2(features, coors, mode="avg", unq_inv=unq_inv, new_coors=new_coors,
                                                      short_segments=True)
                if unq_inv is None:
                    unq_inv, new_coors = unq_inv_c, _
            features = hip_ops.vfe_decorate(features, voxel_mean, unq_inv_c, coors, (self.vx, self.vy, self.vz),
                                            (self.x_offset, self.y_offset, self.z_offset), self._with_cluster_center,
                                            self._with_voxel_center)
            return self._vfe_stack(features, coors, unq_inv, new_coors, return_inv)
        features_ls = [features]
        if self._with_cluster_center:
            voxel_mean, mean_coors, unq_inv_c = scatter_v2(features, coors, mode="avg", unq_inv=unq_inv, new_coors=new_coors,
                                                           short_segments=True)  # (segments = voxels of the detection grid)
            points_mean = gather_by_inverse(voxel_mean, unq_inv_c)
            features_ls.append(features[:, :3] - points_mean[:, :3])
        if self._with_voxel_center:
            f_center = features.new_zeros(size=(features.size(0), 3))
            f_center[:, 0] = features[:, 0] - (coors[:, 3].type_as(features) * self.vx + self.x_offset)
            f_center[:, 1] = features[:, 1] - (coors[:, 2].type_as(features) * self.vy + self.y_offset)
            f_center[:, 2] = features[:, 2] - (coors[:, 1].type_as(features) * self.vz + self.z_offset)
            features_ls.append(f_center)
        if self._with_distance:
            features_ls.append(torch.norm(features[:, :3], 2, 1, keepdim=True))
        features = torch.cat(features_ls, dim=-1)
        return self._vfe_stack(features, coors, unq_inv, new_coors, return_inv)

    def _vfe_stack(self, features, coors, unq_inv, new_coors, return_inv):
        for i, vfe in enumerate(self.vfe_layers):
            last = i == len(self.vfe_layers) - 1
            point_feats, voxel_feats, voxel_coors, unq_inv_l, cat = point_group_concat(
                vfe, features, coors, self.mode, unq_inv, new_coors, want_concat=not last, short_segments=True)
            if not last:
                features = cat
        if self.return_point_feats:
            return point_feats
        if return_inv:
            return voxel_feats, voxel_coors, unq_inv_l
        return voxel_feats, voxel_coors


class _SirProductFn(torch.autograd.Function):
    """y = cat([points[:, :3] / normalizer, points[:, 3:], feats, extra / extra_div], 1) * h with the adjoints of feats, extra and h
    (hip_ops.concat_mul / concat_mul_backward, K28); nothing but the inputs is kept for the backward."""

    @staticmethod
    def forward(ctx, points, feats, extra, h, normalizer, extra_div):
        points, feats, h = points.contiguous(), feats.contiguous(), h.contiguous()
        extra = extra.contiguous() if extra is not None else None
        ctx.save_for_backward(points, feats, extra, h)
        ctx.normalizer, ctx.extra_div = normalizer, extra_div
        return hip_ops.concat_mul(points, feats, extra, h, normalizer, extra_div)

    @staticmethod
    def backward(ctx, grad):
        points, feats, extra, h = ctx.saved_tensors
        g_h, g_f, g_e = hip_ops.concat_mul_backward(points, feats, extra, h, grad.contiguous(), ctx.normalizer, ctx.extra_div,
                                                    want_feats=ctx.needs_input_grad[1],
                                                    want_extra=extra is not None and ctx.needs_input_grad[2])
        return None, g_f, g_e, g_h if ctx.needs_input_grad[3] else None, None, None


@VOXEL_ENCODERS.register_module()
class SIRLayer(nn.Module):
    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_rel_mlp=True, rel_mlp_hidden_dims=[16, ], rel_mlp_in_channel=3, with_voxel_center=False,
                 voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), mode="max", fusion_layer=None,
                 return_point_feats=False, return_inv=True, rel_dist_scaler=10.0, with_shortcut=True,
                 xyz_normalizer=[1.0, 1.0, 1.0], act="relu", dropout=0.0):
        super().__init__()
        assert mode in ("avg", "max") and len(feat_channels) > 0 and fusion_layer is None
        assert not with_cluster_center and not with_voxel_center and not with_distance, \
            "the FSF configs build SIRLayer without these decorations (sir.py:44-48)"
        self.in_channels = in_channels
        self.return_point_feats = return_point_feats
        self.return_inv = return_inv
        self.rel_dist_scaler = rel_dist_scaler
        self.mode = mode
        self.with_shortcut = with_shortcut
        self._with_rel_mlp = with_rel_mlp
        self.xyz_normalizer = list(xyz_normalizer)
        self.fp16_enabled = False
        chans = [self.in_channels] + list(feat_channels)
        layers = []
        for i in range(len(chans) - 1):
            in_filters = chans[i] * 2 if i > 0 else chans[i]
            layers.append(DynamicVFELayer(in_filters, chans[i + 1], norm_cfg, act=act, dropout=dropout))
        self.vfe_layers = nn.ModuleList(layers)
        self.num_vfe = len(layers)
        if with_rel_mlp:
            self.rel_mlp = build_mlp(rel_mlp_in_channel, list(rel_mlp_hidden_dims) + [in_channels], norm_cfg, act=act)

    def _fused_input_layers(self):
        """The position MLP as three (Linear weight, LN weight, LN bias) triples if it has the shape K21 fuses."""
        from ...ops.sst_ops import MLPBlock

        if not self._with_rel_mlp or len(self.rel_mlp) != 3:
            return None
        layers, eps, act = [], None, None
        for blk in self.rel_mlp:
            if not isinstance(blk, MLPBlock) or len(blk) != 3:
                return None
            lin, norm, a = blk[0], blk[1], blk[2]
            code = "relu" if isinstance(a, nn.ReLU) else "gelu" if isinstance(a, nn.GELU) and getattr(a, "approximate", "none") == "none" else None
            if lin.bias is not None or not isinstance(norm, nn.LayerNorm) or code is None or (act or code) != code or \
                    (eps or norm.eps) != norm.eps:
                return None
            eps, act = norm.eps, code
            layers.append((lin.weight, norm.weight, norm.bias))
        if layers[0][0].size(1) > 16 or layers[0][0].size(0) > 16 or layers[1][0].size(0) > 32 or layers[2][0].size(0) > 256:
            return None
        return layers, eps, act

    def forward_parts(self, points, feats, coors, f_cluster, extra=None, extra_div=1.0, **kwargs):
        """`forward(cat([points, feats(, extra / extra_div)], 1), coors, f_cluster, ...)` — what SIR.forward and the refine
        head feed a block.  Inference: concat, xyz normalisation, position MLP and the product are ONE kernel (K21)."""
        fused = None
        needs_grad = torch.is_grad_enabled() and (feats.requires_grad or points.requires_grad or
                                                  any(p.requires_grad for p in self.parameters()))
        if not needs_grad and feats.is_cuda and feats.dtype == torch.float32 and feats.size(0) > 0:
            fused = self._fused_input_layers()
        gathered = isinstance(feats, GatheredRows)
        if gathered and (fused is None or len(feats.sources) > 3):
            feats, gathered = feats.materialize(), False
        if fused is None:
            if (needs_grad and self._with_rel_mlp and not gathered and torch.is_tensor(feats)
                    and feats.is_cuda and feats.dtype == torch.float32 and points.dtype == torch.float32 and feats.dim() == 2
                    and not points.requires_grad and points.size(1) >= 3 and feats.size(0) > 0
                    and (extra is None or extra.dtype == torch.float32) and not torch.is_autocast_enabled()):
                # training: the position MLP stays in autograd; the two concatenations and the product around it are one kernel
                # each way (K28), bit-identical to the ATen chain of `forward`
                h = self.rel_mlp(f_cluster / self.rel_dist_scaler)
                if h.dtype == torch.float32:  # (anything else — a mixed-precision wrapper around the MLP — takes the ATen chain below)
                    features = _SirProductFn.apply(points, feats, extra, h, tuple(float(v) for v in self.xyz_normalizer),
                                                   float(extra_div))
                    return self._run_vfe(features, coors, **kwargs)
                parts = [points, feats] + ([extra / extra_div] if extra is not None else [])
                x = torch.cat(parts, 1)
                nrm = torch.tensor(self.xyz_normalizer, device=x.device, dtype=x.dtype)
                x = torch.cat([x[:, :3] / nrm[None, :], x[:, 3:]], dim=1) * h
                return self._run_vfe(x, coors, **kwargs)
            parts = [points, feats] + ([extra / extra_div] if extra is not None else [])
            return self.forward(torch.cat(parts, 1), coors, f_cluster, **kwargs)
        layers, eps, act = fused
        if gathered:  # rows of the source tensors through the sampling index, parts side by side: no gather, no concat
            features = hip_ops.sir_input(points, feats.sources, f_cluster, self.xyz_normalizer, (*layers, eps), act,
                                         self.rel_dist_scaler, extra=extra, extra_div=extra_div, feats_index=feats.index,
                                         direct_parts=feats.direct)
        else:
            features = hip_ops.sir_input(points, feats, f_cluster, self.xyz_normalizer, (*layers, eps), act, self.rel_dist_scaler,
                                         extra=extra, extra_div=extra_div)
        return self._run_vfe(features, coors, **kwargs)

    # ---- inference on rows SORTED by group (SIR.forward / FullySparseBboxHead permute once per stack) -------------------
    def sorted_supported(self):
        """K21 takes the position MLP and K22s (Linear + norm + act + segmented max in one pass) every layer of the stack."""
        # `_run_vfe` adds the residual on the LAST layer whenever its output is as wide as its input — `in_channels` for a one-layer
        # block, the `cat(point, group[inv])` of the previous layer (2 x prev) otherwise, e.g. feat_channels=[64, 128]; K22s has no
        # residual, so such a block stays on the unsorted path
        last = self.vfe_layers[-1].linear
        last_with_shortcut = self.with_shortcut and last.out_features == last.in_features
        return (not self.training and self._fused_input_layers() is not None and not last_with_shortcut
                and sorted_stack_supported(self.vfe_layers, self.mode))

    def group_width(self):
        return sum(v.linear.out_features for v in self.vfe_layers)

    def forward_sorted(self, points, feats, f_cluster, seg_ids, group_out, want_rows, extra=None, extra_div=1.0,
                       rows_index=None):
        """One block on rows sorted by group.  `points` / `f_cluster` / `extra` are in sorted order already; `feats` is either a
        tensor in sorted order or — with `rows_index` (sorted row -> source row) or as GatheredRows — read through an index by K21.
        group_out f32 [m, group_width()] (holding -inf) receives the block's group features; returns the point rows (sorted),
        or None unless `want_rows`."""
        layers, eps, act = self._fused_input_layers()
        direct = ()
        if isinstance(feats, GatheredRows):
            sources, index, direct = feats.sources, feats.index, feats.direct
        elif rows_index is not None:
            sources, index = [feats], rows_index
        else:
            sources, index = feats, None
        features = hip_ops.sir_input(points, sources, f_cluster, self.xyz_normalizer, (*layers, eps), act, self.rel_dist_scaler,
                                     extra=extra, extra_div=extra_div, feats_index=index, direct_parts=direct)
        return sorted_stack_forward(self.vfe_layers, features, seg_ids, group_out, want_rows)

    def forward(self, features, coors, f_cluster=None, points=None, img_feats=None, img_metas=None, return_both=False,
                unq_inv_once=None, new_coors_once=None):
        xyz_normalizer = torch.tensor(self.xyz_normalizer, device=features.device, dtype=features.dtype)
        features = torch.cat([features[:, :3] / xyz_normalizer[None, :], features[:, 3:]], dim=1)
        if self._with_rel_mlp:
            features = features * self.rel_mlp(f_cluster / self.rel_dist_scaler)
        return self._run_vfe(features, coors, return_both=return_both, unq_inv_once=unq_inv_once, new_coors_once=new_coors_once)

    def _run_vfe(self, features, coors, return_both=False, unq_inv_once=None, new_coors_once=None):
        voxel_feats_list = []
        for i, vfe in enumerate(self.vfe_layers):
            last = i == len(self.vfe_layers) - 1
            if last and self.with_shortcut and vfe.linear.out_features == features.shape[1]:
                # shortcut (never shape-compatible in the FSF configs): plain path
                if isinstance(features, GroupedConcat):
                    features = features.materialize()
                point_feats = vfe(features) + features
                voxel_feats, voxel_coors, unq_inv = scatter_v2(point_feats, coors, mode=self.mode, unq_inv=unq_inv_once,
                                                               new_coors=new_coors_once)
            else:
                point_feats, voxel_feats, voxel_coors, unq_inv, cat = point_group_concat(
                    vfe, features, coors, self.mode, unq_inv_once, new_coors_once, want_concat=not last)
                if not last:
                    features = cat
            voxel_feats_list.append(voxel_feats)
        voxel_feats = torch.cat(voxel_feats_list, dim=1)
        if return_both:
            if self.return_inv:
                return point_feats, voxel_feats, voxel_coors, unq_inv
            return point_feats, voxel_feats, voxel_coors
        if self.return_point_feats:
            return point_feats, voxel_feats
        if self.return_inv:
            return voxel_feats, voxel_coors, unq_inv
        return voxel_feats, voxel_coors


@VOXEL_ENCODERS.register_module()
class DynamicClusterVFE(SIRLayer):
    """The refine stage's SIR-layer variant [UNVENDORED in the reference; built by FullySparseBboxHead,
    projects/mmdet3d_plugin/models/roi_heads/bbox_heads/fsd_bbox_head.py:60-88].  With the only argument values that
    call site passes (`fusion='cat'`, `pos_fusion='mul'`, `cat_voxel_feats=True`, no cluster/voxel-centre/distance
    decorations, gelu => in-filters not doubled for block 0) the published module computes exactly what SIRLayer does;
    other values are refused instead of guessed."""

    def __init__(self, *args, fusion="cat", pos_fusion="mul", cat_voxel_feats=True, **kwargs):
        assert fusion == "cat" and pos_fusion == "mul" and cat_voxel_feats, \
            "DynamicClusterVFE: only the fusion='cat' / pos_fusion='mul' / cat_voxel_feats=True variant is built"
        kwargs.setdefault("with_shortcut", False)
        super().__init__(*args, **kwargs)
