"""Host-side mirror of projects/mmdet3d_plugin/ops/sst_ops.py for the functions the FSF configs use:
`scatter_v2` (:150-177), `get_inner_win_inds` (:239-259), `build_mlp` (:808-833),
`get_activation` / `get_activation_layer` (:835-864).  Same names, arguments and error behaviour; the work is
done by the HIP library (no torch_scatter, no TorchEx)."""
from ... import switches
import traceback

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import hip_ops
from ..registry import build_norm_layer

_PLAN_ATTR = "_fsf_segment_plan"


class _SegmentReduce(torch.autograd.Function):
    """Differentiable w.r.t. `feat` only, like torch_scatter's scatter / scatter_max (sst_ops.py:168-170)."""

    @staticmethod
    def forward(ctx, feat, plan, mode, short=False):
        ctx.plan, ctx.mode = plan, mode
        if mode == "max" and feat.requires_grad:
            if short:
                outs, arg = hip_ops.segment_reduce_short([feat], plan, "max", return_argmax=True)
                out = outs[0]
            else:
                out, arg = hip_ops.segment_reduce(feat, plan, "max", return_argmax=True)
            ctx.save_for_backward(arg)
            return out
        ctx.save_for_backward()
        if short:
            return hip_ops.segment_reduce_short([feat], plan, mode)[0]
        return hip_ops.segment_reduce(feat, plan, mode)

    @staticmethod
    def backward(ctx, grad_out):
        arg = ctx.saved_tensors[0] if ctx.mode == "max" else None
        return hip_ops.segment_reduce_backward(grad_out.contiguous(), ctx.plan, ctx.mode, argmax=arg), None, None, None


class _GatherRows(torch.autograd.Function):
    """rows[inv] with a deterministic segmented-sum adjoint (replaces ATen index + index_put atomics)."""

    @staticmethod
    def forward(ctx, src, plan):
        ctx.plan = plan
        return hip_ops.gather_rows(src.float(), plan.inv)

    @staticmethod
    def backward(ctx, grad_out):
        return hip_ops.segment_reduce(grad_out.contiguous(), ctx.plan, "sum"), None


import threading

_UNIQUE_TLS = threading.local()  # .entries: (coors tensor, _version, bounds, result) of this thread's last few calls, newest last
_UNIQUE_CACHE_SIZE = 2


def _unique_entries():
    e = getattr(_UNIQUE_TLS, "entries", None)
    if e is None:
        e = _UNIQUE_TLS.entries = []
    return e


_BOUNDS_ATTR = "_fsf_key_bounds"


def with_key_bounds(keys, col_min, col_max):
    """Attach per-column bounds the producer of `keys` knows from its configuration (grid sizes, class / batch / row counts) so that
    `unique_with_plan` packs the sort key from them instead of running the range kernel and WAITING for its result (two launches and
    one host round trip per unique).  Violations are detected on the device and fall back to the range pass."""
    setattr(keys, _BOUNDS_ATTR, ([int(v) for v in col_min], [int(v) for v in col_max]))
    return keys


def unique_with_plan(coors, col_min=None, col_max=None):
    """torch.unique(coors, return_inverse=True, return_counts=True, dim=0) + the sort-once segment plan.
    The plan rides on the returned inverse tensor so that the `unq_inv=`/`new_coors=` path of scatter_v2
    (used by unique_once VFE / SIR blocks) reuses it instead of sorting again.

    The reference calls scatter_v2 on the SAME key tensor from different places without passing the inverse along
    (SingleStageFSD.extract_feat takes the cluster means, then SIR.forward runs its own unique on `pts_cluster_inds`;
    get_cluster_delta_weighted, then the frustum SIR on `sir_coors`): the result of the last calls is kept, keyed on the tensor
    OBJECT and its version counter (the tensor is held, so its storage cannot be recycled under the key), at inference only."""
    cache_ok = _UNIQUE_CACHE_SIZE > 0 and coors.is_cuda and not torch.is_grad_enabled()
    hinted = False
    if col_min is None:
        b = getattr(coors, _BOUNDS_ATTR, None)
        if b is not None:
            (col_min, col_max), hinted = b, True
    key = (None if col_min is None else tuple(col_min), None if col_max is None else tuple(col_max))
    if cache_ok:  # (per thread: the two query branches run on two host threads and streams)
        for t, ver, k, res in reversed(_unique_entries()):
            if t is coors and ver == coors._version and k == key:
                return res
    try:
        new_coors, plan = hip_ops.unique_rows(coors, col_min=col_min, col_max=col_max)
    except hip_ops.FsfHipError as e:
        # bounds a producer ATTACHED to its keys (with_key_bounds) are a promise about typical data, not a contract: a key outside
        # them (or bounds too wide for a 64-bit packed key) sends this call through the data-dependent range pass instead
        if not hinted or getattr(e, "status", 0) != hip_ops.ERR_KEY_RANGE:
            raise
        try:  # the promise did not hold for THIS tensor: later uniques on it go straight to the range pass
            delattr(coors, _BOUNDS_ATTR)
        except AttributeError:
            pass
        key = (None, None)
        new_coors, plan = hip_ops.unique_rows(coors)
    inv = plan.inv.detach()  # a second tensor object on the same storage: tensor -> plan -> tensor would be a cycle
    setattr(inv, _PLAN_ATTR, plan)
    res = (new_coors, inv, plan.cnt)
    if cache_ok:
        entries = _unique_entries()
        entries.append((coors, coors._version, key, res))
        del entries[:-_UNIQUE_CACHE_SIZE]
    return res


def seed_unique_cache(coors, new_coors, plan):
    """Enter a unique that was computed elsewhere (the native LiDAR front end, K30: fsf_unique_rows inside the stage driver) as if
    `unique_with_plan(coors)` had just returned it — keyed like that call would key it (the bounds attached to `coors`)."""
    if not (_UNIQUE_CACHE_SIZE > 0 and coors.is_cuda and not torch.is_grad_enabled()):
        return
    b = getattr(coors, _BOUNDS_ATTR, None)
    key = (None, None) if b is None else (tuple(b[0]), tuple(b[1]))
    inv = plan.inv.detach()
    setattr(inv, _PLAN_ATTR, plan)
    entries = _unique_entries()
    entries.append((coors, coors._version, key, (new_coors, inv, plan.cnt)))
    del entries[:-_UNIQUE_CACHE_SIZE]


def clear_unique_cache():
    """Drop the cached uniques (and the key / result tensors they hold): the detector calls this at the end of a frame."""
    del _unique_entries()[:]


def seed_unique_result(coors, res):
    """Enter `res` = what `unique_with_plan(coors)` returned on another stream / at another time as this thread's newest entry."""
    if not (_UNIQUE_CACHE_SIZE > 0 and coors.is_cuda and not torch.is_grad_enabled()):
        return
    b = getattr(coors, _BOUNDS_ATTR, None)
    key = (None, None) if b is None else (tuple(b[0]), tuple(b[1]))
    entries = _unique_entries()
    entries.append((coors, coors._version, key, res))
    del entries[:-_UNIQUE_CACHE_SIZE]


def swap_unique_cache(entries):
    """Replace this thread's cached uniques by `entries` and return the ones that were there (FSF's frame front runs inside the frame
    before its own and must neither see nor evict that frame's entries; what it caches travels with its state)."""
    old = list(_unique_entries())
    _unique_entries()[:] = entries
    return old


def plan_of(unq_inv, num_segments):
    plan = getattr(unq_inv, _PLAN_ATTR, None)
    if plan is None or plan.m != num_segments:
        plan = hip_ops.segment_plan_from_inverse(unq_inv, int(num_segments))
        setattr(unq_inv, _PLAN_ATTR, plan)
    return plan


def gather_by_inverse(rows, unq_inv, out=None):
    """`rows[unq_inv]` (the "map back to points" gather of DynamicScatterVFE / SIRLayer / the neck).  With `out` (a
    column slice of a wider buffer; inference only) the rows land in place and no concat copy is needed."""
    plan = plan_of(unq_inv, rows.size(0))
    if out is not None:
        return hip_ops.gather_rows(rows, plan.inv, out=out)
    return _GatherRows.apply(rows, plan)


_SMALL_N_MIN = 16  # rows from which a shallow product runs on K22 rather than on the library (host time of the GEMM selection)


class GroupedConcat:
    """`cat([point_feats, group_feats[inv]], 1)` kept as its three parts.  The only consumer is the next
    layer's Linear, and `cat(p, g[inv]) W^T = p W_left^T + (g W_right^T)[inv]`: the right half is applied once per group
    ([g, C] instead of [n, C] rows) and added per row inside K22's epilogue (fsf_linear_norm_act_grouped) — the [n, 2C]
    tensor is never written and the per-point product is half as deep.  In training the same identity is taken through autograd
    (`_grouped_linear_training`): the adjoint of the per-row add is a deterministic segmented sum."""

    def __init__(self, point_feats, group_feats, inv):
        self.point_feats, self.group_feats, self.inv = point_feats, group_feats, inv

    @property
    def shape(self):
        return (self.point_feats.size(0), self.point_feats.size(1) + self.group_feats.size(1))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def materialize(self):
        n, c = self.point_feats.shape
        if torch.is_grad_enabled() and (self.point_feats.requires_grad or self.group_feats.requires_grad):
            return torch.cat([self.point_feats, gather_by_inverse(self.group_feats, self.inv)], dim=1)
        buf = torch.empty((n, c + self.group_feats.size(1)), dtype=self.point_feats.dtype, device=self.point_feats.device)
        buf[:, :c] = self.point_feats
        gather_by_inverse(self.group_feats, self.inv, out=buf[:, c:])
        return buf


class GatheredRows:
    """`torch.cat([t.index_select(0, index) for t in sources], 1)` kept as its parts (inference).  The only consumer of the
    group-sampled points' features is the first SIR layer's input kernel, which reads rows through an index and several
    tensors side by side (fsf_sir_input_gather): the [n_pairs, 11 + 33 + 131] matrix is never written."""

    def __init__(self, sources, index, direct=()):
        # `direct`: positions in `sources` of tensors that hold the gathered rows ALREADY (n rows, read as they stand)
        self.sources, self.index, self.direct = list(sources), index, tuple(direct)

    @property
    def shape(self):
        return (self.index.numel(), sum(t.size(1) for t in self.sources))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def __len__(self):
        return self.index.numel()

    is_cuda = property(lambda self: self.sources[0].is_cuda)
    dtype = property(lambda self: self.sources[0].dtype)
    requires_grad = property(lambda self: any(t.requires_grad for t in self.sources))

    def materialize(self):
        n, widths = self.index.numel(), [t.size(1) for t in self.sources]
        buf = torch.empty((n, sum(widths)), dtype=self.sources[0].dtype, device=self.sources[0].device)
        c0 = 0
        for i, (t, w) in enumerate(zip(self.sources, widths)):
            if i in self.direct:
                buf[:, c0:c0 + w] = t
            else:
                hip_ops.gather_rows(t, self.index, out=buf[:, c0:c0 + w])
            c0 += w
        return buf


class RowsMinusGroup:
    """`points[:, :3] - centers[inv]` kept as its parts (inference): the offset of every point to its group's centre
    (SingleStageFSD.extract_feat, single_stage_fsd.py:458-474; FSF.get_cluster_delta_weighted, FSF.py:313-329).  Its only consumer is
    the SIR stack that follows, which on its sorted path forms the rows while it permutes its operands (fsf_sorted_rows): the gather
    and the subtraction are not launches of their own.  `materialize()` is the expression itself, for anything else."""

    def __init__(self, points, centers, inv):
        self.points, self.centers, self.inv = points, centers, inv

    dtype = property(lambda self: self.points.dtype)
    is_cuda = property(lambda self: self.points.is_cuda)

    @property
    def shape(self):
        return (self.points.size(0), 3)

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def materialize(self):
        return self.points[:, :3] - gather_by_inverse(self.centers, self.inv)


def _grouped_linear_norm_act(linear, norm, act, gc):
    """act(norm(linear(cat))) for a GroupedConcat through fsf_linear_norm_act_grouped; None when the layer is not covered."""
    p, g, inv = gc.point_feats, gc.group_feats, gc.inv
    c_left = p.size(1)
    act_code = "relu" if isinstance(act, nn.ReLU) else (
        "gelu" if isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none" else None)
    if (act_code is None or not isinstance(linear, nn.Linear) or linear.in_features != gc.shape[1] or c_left % 4
            or linear.out_features % 4 or p.size(0) < 1024 or not hip_ops.linear_norm_act_supported(p, linear.out_features)
            or inv.dtype != torch.int64):
        return None
    if isinstance(norm, nn.LayerNorm) and len(norm.normalized_shape) == 1 and norm.elementwise_affine and linear.out_features <= 128:
        kind, gamma, beta, eps = "ln", norm.weight, norm.bias, norm.eps
    elif isinstance(norm, nn.BatchNorm1d) and not norm.training and norm.track_running_stats:
        from .spconv import _bn_affine

        kind, eps = "affine", 0.0
        gamma, beta = _bn_affine(norm)
    else:
        return None
    key = (linear.weight.data_ptr(), linear.weight._version, linear.weight.device, c_left)
    cache = linear.__dict__.get("_fsf_planes_grouped")
    if cache is None or cache[0] != key:
        w = linear.weight.detach()
        w_right = w[:, c_left:].contiguous()
        cache = (key, hip_ops.linear_prepare_weight(w[:, :c_left].contiguous()), w_right, hip_ops.linear_prepare_weight(w_right))
        linear.__dict__["_fsf_planes_grouped"] = cache
    # [groups, C_out]: the right half, once per group (a few hundred to 1e4 rows: on K22 as well — the library spends
    # 60 us of host time per call choosing a GEMM for a 244-row input)
    if g.size(0) >= _SMALL_N_MIN and g.size(1) <= 256 and hip_ops.linear_norm_act_supported(g, linear.out_features):
        table = hip_ops.linear_norm_act(g, cache[3], linear.out_features)
    else:
        table = F.linear(g, cache[2])
    return hip_ops.linear_norm_act(p, cache[1], linear.out_features, bias=linear.bias, norm=kind, gamma=gamma, beta=beta,
                                   eps=eps, act=act_code, row_add=table, row_add_index=inv.contiguous())


def _k22_norm_act(norm, act, out_features):
    """(norm kind, gamma, beta, eps, act code) of a [norm, act] pair K22's epilogue covers, else None."""
    act_code = "relu" if isinstance(act, nn.ReLU) else (
        "gelu" if isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none" else None)
    if act_code is None:
        return None
    if isinstance(norm, nn.LayerNorm) and len(norm.normalized_shape) == 1 and norm.elementwise_affine and out_features <= 128:
        return "ln", norm.weight, norm.bias, norm.eps, act_code
    if isinstance(norm, nn.BatchNorm1d) and not norm.training and norm.track_running_stats:
        from .spconv import _bn_affine

        gamma, beta = _bn_affine(norm)
        return "affine", gamma, beta, 0.0, act_code
    return None


def sorted_stack_supported(vfe_layers, mode):
    """Can the `DynamicVFELayer`s of a SIRLayer run as K22s launches (Linear + norm + act + segmented max in one pass over rows
    sorted by segment)?  Every layer: plain nn.Linear into 36..128 channels (a multiple of 4), LayerNorm / eval BatchNorm, ReLU /
    exact GELU, no dropout; the reduction is a max; every layer after the first takes `cat(point, group[inv])` of the previous."""
    if mode != "max" or len(vfe_layers) == 0:
        return False
    prev = None
    for i, vfe in enumerate(vfe_layers):
        lin = vfe.linear
        c = lin.out_features
        if (type(lin) is not nn.Linear and not isinstance(lin, nn.Linear)) or vfe.dropout is not None or c % 4 or not 32 < c <= 128:
            return False
        na = _k22_norm_act(vfe.norm, vfe.act, c)
        if na is None or na[0] != "ln":  # (K22s is built for LayerNorm + ReLU / GELU: what SIRLayer uses)
            return False
        if i > 0 and (lin.in_features != 2 * prev or prev % 4):
            return False
        prev = c
    return True


def _grouped_planes(lin, c_left):
    """(key, prepared W[:, :c_left], W[:, c_left:], prepared W[:, c_left:]) of a layer that takes `cat([point, group[inv]], 1)`:
    prepared once per weight version."""
    key = (lin.weight.data_ptr(), lin.weight._version, lin.weight.device, c_left)
    cache = lin.__dict__.get("_fsf_planes_grouped")
    if cache is None or cache[0] != key:
        w = lin.weight.detach()
        w_right = w[:, c_left:].contiguous()
        cache = (key, hip_ops.linear_prepare_weight(w[:, :c_left].contiguous()), w_right, hip_ops.linear_prepare_weight(w_right))
        lin.__dict__["_fsf_planes_grouped"] = cache
    return cache


def sir_stack_descriptor(owner, blocks):
    """hip_ops.SirStackDescriptor of a stack of SIRLayer / DynamicClusterVFE blocks (fsf_sir_stack_forward, K31), cached on `owner`
    until a parameter of the stack changes; None when a block is outside what the native stack takes (the per-block path then runs)."""
    params = [p for b in blocks for p in b.parameters()]
    key = tuple((p.data_ptr(), p._version) for p in params) + (switches.K22F,)
    cached = owner.__dict__.get("_fsf_sir_stack_desc")
    if cached is not None and cached[0] == key:
        return cached[1]
    specs = []
    for b in blocks:
        fused = b._fused_input_layers()
        if fused is None or not (1 <= len(b.vfe_layers) <= 4):
            specs = None
            break
        layers, eps, act = fused
        ls, prev = [], None
        for i, vfe in enumerate(b.vfe_layers):
            lin = vfe.linear
            c = lin.out_features
            na = _k22_norm_act(vfe.norm, vfe.act, c)
            if na is None:
                specs = None
                break
            kind, gamma, beta, e, act_code = na
            if i == 0:
                left, right = _prepared_planes(lin), None
            else:
                cache = _grouped_planes(lin, prev)
                left, right = cache[1], cache[3]
            ls.append(dict(planes_left=left, planes_right=right, bias=lin.bias, gamma=gamma, beta=beta, eps=e, norm=kind, act=act_code, c=c))
            prev = c
        if specs is None:
            break
        specs.append(dict(mlp=layers, mlp_eps=eps, mlp_act=act, xyz_normalizer=b.xyz_normalizer, rel_div=b.rel_dist_scaler,
                          in_cols=layers[2][0].size(0), layers=ls))
    desc = hip_ops.SirStackDescriptor(specs) if specs is not None else None
    owner.__dict__["_fsf_sir_stack_desc"] = (key, desc)
    return desc


def sorted_stack_forward(vfe_layers, x, seg_ids, group_out, want_last_rows):
    """The layer stack of one SIRLayer on rows SORTED by group: per layer ONE K22s launch (`fsf_linear_norm_act_segmax`) computes
    point_feats = act(norm(linear(.))) and the group maxima; from the second layer on the input `cat([point, group[inv]], 1)` is
    taken as `point W_left^T + (group W_right^T)[seg_ids]` (the right half once per group, added per row in the epilogue).
    `group_out` f32 [m, sum of the layers' widths] (pre-filled with -inf) receives the layers' group features side by side —
    the `cat` of SIRLayer's return value is never formed.  Returns the last layer's point rows (None unless `want_last_rows`)."""
    col = 0
    point = x
    for i, vfe in enumerate(vfe_layers):
        lin = vfe.linear
        c = lin.out_features
        kind, gamma, beta, eps, act_code = _k22_norm_act(vfe.norm, vfe.act, c)
        last = i == len(vfe_layers) - 1
        seg_out = group_out[:, col:col + c]
        if i == 0:
            point = hip_ops.linear_norm_act_segmax(point, _prepared_planes(lin), c, seg_ids, seg_out, bias=lin.bias,
                                                   norm=kind, gamma=gamma, beta=beta, eps=eps, act=act_code,
                                                   want_rows=(not last) or want_last_rows)
        else:
            c_left = point.size(1)
            cache = _grouped_planes(lin, c_left)
            g = group_out[:, col - c_left:col]  # the previous layer's group maxima (a column slice: rows 16-byte aligned)
            if g.size(0) >= _SMALL_N_MIN and hip_ops.linear_norm_act_supported(g, c):
                table = hip_ops.linear_norm_act(g, cache[3], c)
            else:
                table = F.linear(g, cache[2])
            point = hip_ops.linear_norm_act_segmax(point, cache[1], c, seg_ids, seg_out, bias=lin.bias, norm=kind,
                                                   gamma=gamma, beta=beta, eps=eps, act=act_code, row_add=table,
                                                   row_add_index=seg_ids, want_rows=(not last) or want_last_rows)
        col += c
    return point




def _grouped_linear_training(linear, gc):
    """`linear(cat([p, g[inv]], 1))` with gradients, without the concat: p W_left^T (per point; weight gradient on K10) +
    (g W_right^T (+ b))[inv] (per group, then one gather-add pass).  None when the input is not covered."""
    p, g = gc.point_feats, gc.group_feats
    if not (isinstance(linear, nn.Linear) and p.is_cuda and p.dtype == torch.float32 and p.dim() == 2
            and g.dtype == torch.float32 and linear.in_features == p.size(1) + g.size(1) and p.size(0) >= 16384 and g.size(0) > 0):
        return None
    c = p.size(1)
    w = linear.weight
    t = F.linear(g, w[:, c:], linear.bias)
    return _PointLinearFn.apply(p, w[:, :c], None, t, plan_of(gc.inv, g.size(0)))


def point_group_concat(vfe_layer, features, coors, mode, unq_inv, new_coors, want_concat, short_segments=False):
    """One `DynamicVFELayer` step of DynamicScatterVFE / SIRLayer: point_feats = act(norm(linear(x))), group feats =
    segmented reduce, and (unless it is the last layer) `cat([point_feats, group_feats[inv]], 1)`.
    Inference: the fused norm+act writes the left half of the concat buffer and the row gather the right half — the
    [n, 2C] tensor is written exactly once; the segmented reduce reads the left half through its row stride."""
    grouped_in = isinstance(features, GroupedConcat)
    no_grad = not (torch.is_grad_enabled() and ((features.point_feats.requires_grad or features.group_feats.requires_grad
                                                 if grouped_in else features.requires_grad)
                                                or any(p.requires_grad for p in vfe_layer.parameters())))
    point_feats = None
    if grouped_in:
        if no_grad and vfe_layer.dropout is None:
            point_feats = _grouped_linear_norm_act(vfe_layer.linear, vfe_layer.norm, vfe_layer.act, features)
        elif not no_grad:
            pre = _grouped_linear_training(vfe_layer.linear, features)
            if pre is not None:
                point_feats = fused_norm_act(pre, vfe_layer.norm, vfe_layer.act)
                if vfe_layer.dropout is not None:
                    point_feats = vfe_layer.dropout(point_feats)
        if point_feats is None:
            features = features.materialize()
    if no_grad and vfe_layer.dropout is None and (want_concat or point_feats is not None):
        if point_feats is None:
            point_feats = linear_norm_act(vfe_layer.linear, vfe_layer.norm, vfe_layer.act, features)
        group_feats, group_coors, inv = scatter_v2(point_feats, coors, mode=mode, unq_inv=unq_inv, new_coors=new_coors,
                                                   short_segments=short_segments)
        cat = None
        if want_concat:
            cat = GroupedConcat(point_feats, group_feats, inv)
        return point_feats, group_feats, group_coors, inv, cat
    if point_feats is None:
        point_feats = vfe_layer(features)
    group_feats, group_coors, inv = scatter_v2(point_feats, coors, mode=mode, unq_inv=unq_inv, new_coors=new_coors,
                                               short_segments=short_segments)
    cat = None
    if want_concat:
        if (not no_grad and point_feats.is_cuda and point_feats.dtype == torch.float32
                and point_feats.size(0) >= 16384):
            cat = GroupedConcat(point_feats, group_feats, inv)  # consumed by the next layer's _grouped_linear_training
        else:
            cat = torch.cat([point_feats, gather_by_inverse(group_feats, inv)], dim=1)
    return point_feats, group_feats, group_coors, inv, cat




def scatter_v2(feat, coors, mode, return_inv=True, min_points=0, unq_inv=None, new_coors=None, short_segments=False):
    """`short_segments` (not upstream): the caller knows the segments are voxels — a few rows each — and the reduction takes
    the thread-per-(segment, channel) kernel (one launch instead of three; results equal up to the fp32 summation order)."""
    assert feat.size(0) == coors.size(0)
    if mode == "avg":
        mode = "mean"
    if mode not in ("max", "mean", "sum"):
        raise NotImplementedError
    unq_cnt = None
    if unq_inv is None:
        new_coors, unq_inv, unq_cnt = unique_with_plan(coors)
    else:
        assert new_coors is not None, "please pass new_coors for interface consistency, caller: {}".format(
            traceback.extract_stack()[-2][2])
    if min_points > 0:
        cnt_per_point = unq_cnt[unq_inv]  # NameError-equivalent upstream when unq_inv was supplied (:161)
        valid_mask = cnt_per_point >= min_points
        feat = feat[valid_mask]
        coors = coors[valid_mask]
        new_coors, unq_inv, unq_cnt = unique_with_plan(coors)
    plan = plan_of(unq_inv, new_coors.size(0))
    new_feat = _SegmentReduce.apply(feat.float(), plan, mode, bool(short_segments))
    if not return_inv:
        return new_feat, new_coors
    return new_feat, new_coors, unq_inv


def scatter_mean_multi(feats, new_coors, unq_inv):
    """The mean of several per-point tensors over ONE short-segment plan (pre_voxelize: every float field of the point dict
    over the same 0.1 m voxels) — at inference two launches at most; otherwise one scatter_v2 per tensor."""
    plan = plan_of(unq_inv, new_coors.size(0))
    feats = [f.float() for f in feats]
    if (1 <= len(feats) <= 8 and all(f.is_cuda and f.dim() == 2 for f in feats)
            and not (torch.is_grad_enabled() and any(f.requires_grad for f in feats))):
        # Everything that can be read with float4 lanes goes into one launch: tensors whose rows sit in 16-byte-aligned storage padded
        # to a multiple of four floats (the 131-column point features in their 132-float rows, the 33 vote offsets in 36: the pad
        # column's mean is computed and dropped), and column blocks that stand side by side in one buffer (the segmentation head's
        # logits | vote predictions = the [n, 11 + 33] result of its one stacked launch) as that buffer.  The odd-width rest (the 5-column
        # points) takes the four-byte lanes.  Four-byte lanes over all 213 columns took 324 us on the 0.1 m voxels of the 10-sweep frame;
        # round 5's split (only >= 64-wide tensors on float4 lanes) left 82 columns = 139 us on them, this one 5.
        def aligned(f, c):
            return (f.stride(1) == 1 and f.size(0) > 1 and f.stride(0) % 4 == 0 and f.stride(0) >= c and f.data_ptr() % 16 == 0)

        groups = []  # (view to reduce, [(index into feats, first column, width)])
        used = [False] * len(feats)
        for i, f in enumerate(feats):
            if used[i]:
                continue
            members, end = [(i, 0, f.size(1))], f.size(1)
            if f.stride(1) == 1 and f.size(0) > 1:
                grown = True
                while grown:  # column blocks of the same rows that follow this one directly
                    grown = False
                    for j, g in enumerate(feats):
                        if (not used[j] and j != i and all(j != m[0] for m in members) and g.stride(1) == 1 and g.stride(0) == f.stride(0)
                                and g.size(0) == f.size(0) and g.data_ptr() == f.data_ptr() + 4 * end and end + g.size(1) <= f.stride(0)):
                            members.append((j, end, g.size(1)))
                            end += g.size(1)
                            grown = True
            c4 = (end + 3) // 4 * 4
            if aligned(f, c4):
                for m in members:
                    used[m[0]] = True
                groups.append((f.as_strided((f.size(0), c4), (f.stride(0), 1)), members))
        out = [None] * len(feats)
        if groups:
            for (view, members), o in zip(groups, hip_ops.segment_reduce_short([v for v, _ in groups], plan, "mean")):
                for idx, c0, w in members:
                    out[idx] = o[:, c0:c0 + w]
        rest = [i for i in range(len(feats)) if out[i] is None]
        if rest:
            for i, o in zip(rest, hip_ops.segment_reduce_short([feats[i] for i in rest], plan, "mean")):
                out[i] = o
        return out
    return [_SegmentReduce.apply(f, plan, "mean", True) for f in feats]


@torch.no_grad()
def get_inner_win_inds(group_inds):
    """IngroupIndicesFunction.apply (sst_ops.py:239-259): per element its rank inside its group
    (non-differentiable).  TorchEx hands out ranks in atomicAdd order; this is the stable rank."""
    return hip_ops.ingroup_rank(group_inds)


def get_activation(activation):
    if activation == "relu":
        return torch.nn.functional.relu
    if activation == "gelu":
        return torch.nn.functional.gelu
    if activation == "glu":
        return torch.nn.functional.glu
    raise RuntimeError(f"activation should be relu/gelu, not {activation}.")


def get_activation_layer(act, dim=None):
    act = act.lower()
    table = {
        "relu": lambda: nn.ReLU(inplace=True),
        "gelu": lambda: nn.GELU(),
        "leakyrelu": lambda: nn.LeakyReLU(inplace=True),
        "prelu": lambda: nn.PReLU(num_parameters=dim),
        "swish": lambda: nn.SiLU(inplace=True),
        "silu": lambda: nn.SiLU(inplace=True),
        "glu": lambda: nn.GLU(),
        "elu": lambda: nn.ELU(inplace=True),
    }
    if act not in table:
        raise NotImplementedError
    return table[act]()


class _NormActFn(torch.autograd.Function):
    """act(LayerNorm(x)) with both directions fused: the forward keeps only x, the backward recomputes the row statistics
    and the pre-activation in registers and emits grad_x, grad_gamma, grad_beta in one pass (fsf_norm_act_backward)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act_code):
        x = x.contiguous()
        ctx.save_for_backward(x, gamma, beta)
        ctx.eps, ctx.act_code = eps, act_code
        return hip_ops.norm_act(x, gamma, beta, eps, "ln", act_code, inplace=False)

    @staticmethod
    def backward(ctx, grad):
        x, gamma, beta = ctx.saved_tensors
        gx, dg, db = hip_ops.norm_act_backward(x, grad, gamma, beta, ctx.eps, ctx.act_code)
        return gx, dg, db, None, None


def fused_norm_act(x, norm, act, out=None):
    """`act(norm(x))` for x [n, C].  When no gradient is needed and the pair is LayerNorm / eval BatchNorm1d followed by
    ReLU / GELU, this is ONE fused HIP pass over the activations (fsf_norm_act) instead of two or three ATen kernels;
    otherwise the torch modules run (training, exotic activations)."""
    act_code = None
    if isinstance(act, nn.ReLU):
        act_code = "relu"
    elif isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none":
        act_code = "gelu"
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in norm.parameters()))
    fusable = act_code is not None and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.size(1) <= 1024
    if (fusable and needs_grad and out is None and isinstance(norm, nn.LayerNorm) and len(norm.normalized_shape) == 1
            and norm.elementwise_affine and x.size(0) > 0
            and (x.size(1) <= 512 or (x.size(1) <= 1024 and x.size(1) % 4 == 0))):
        return _NormActFn.apply(x, norm.weight, norm.bias, norm.eps, act_code)  # training: fused forward AND backward (K12)
    if fusable and not needs_grad:
        x = x.contiguous()
        if isinstance(norm, nn.LayerNorm) and len(norm.normalized_shape) == 1:
            return hip_ops.norm_act(x, norm.weight, norm.bias, norm.eps, "ln", act_code, out=out)
        if isinstance(norm, nn.BatchNorm1d) and not norm.training and norm.track_running_stats:
            from .spconv import _bn_affine

            scale, shift = _bn_affine(norm)
            return hip_ops.norm_act(x, scale, shift, 0.0, "affine", act_code, out=out)
    y = batch_norm_act_training(norm, x, True) if isinstance(act, nn.ReLU) else None
    if y is None:
        y = act(norm(x))
    if out is not None:
        out.copy_(y)
        return out
    return y




class _BatchNormActFn(torch.autograd.Function):
    """Training-mode BatchNorm1d over the rows of [n, C] (+ ReLU) on K23: two-pass batch statistics, one fused
    normalise + activate pass, and a backward of two reads of (x, grad) + one write."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, relu):
        x = x.contiguous()
        n = x.size(0)
        track = bn.track_running_stats and bn.running_mean is not None
        if track:
            with torch.no_grad():
                bn.num_batches_tracked += 1
        if not track or bn.momentum is not None:
            # statistics, invstd / scale / shift and the running-statistics update in four launches (fsf_batch_norm_train_stats)
            mean, invstd, scale, shift = hip_ops.batch_norm_train_stats(
                x, weight.detach() if weight is not None else None, bias.detach() if bias is not None else None, bn.eps,
                bn.momentum if track else 0.0, bn.running_mean if track else None, bn.running_var if track else None)
        else:
            mean, var = hip_ops.column_mean_var(x)
            invstd = torch.rsqrt(var + bn.eps)
            scale = weight.detach() * invstd if weight is not None else invstd
            shift = (bias.detach() if bias is not None else 0.0) - mean * scale
            if track:
                with torch.no_grad():
                    momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                    bn.running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
                    bn.running_var.mul_(1 - momentum).add_(var, alpha=momentum * n / max(n - 1, 1))
        ctx.save_for_backward(x, mean, invstd, scale, shift)
        ctx.relu, ctx.affine = relu, weight is not None
        return hip_ops.batch_norm_act_forward(x, scale, shift, relu)

    @staticmethod
    def backward(ctx, grad):
        x, mean, invstd, scale, shift = ctx.saved_tensors
        gx, dg, db = hip_ops.batch_norm_act_backward(x, grad, mean, invstd, scale, shift, ctx.relu)
        return gx, (dg if ctx.affine else None), (db if ctx.affine else None), None, None


def batch_norm_act_training(bn, x, relu):
    """`relu(bn(x))` / `bn(x)` for a training-mode BatchNorm1d (or naiveSyncBN1d on one rank) on the GPU; None when the
    module / input is not covered (the caller then runs the stock modules)."""
    import torch.distributed as dist

    if not (isinstance(bn, nn.BatchNorm1d) and bn.training and torch.is_grad_enabled() and x.is_cuda and x.dim() == 2
            and x.dtype == torch.float32 and x.size(0) > 1 and (bn.weight is None) == (bn.bias is None)
            ):
        return None
    if type(bn).__name__ != "BatchNorm1d" and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # naiveSyncBN1d across ranks: K23 still does the row passes, the statistics travel as one packed [2C] all-reduce per
        # direction (ops/norm.py::_SyncBatchNormAct)
        return bn.forward_act(x, relu) if hasattr(bn, "forward_act") else None
    return _BatchNormActFn.apply(x, bn.weight, bn.bias, bn, relu)


class _PointLinearFn(torch.autograd.Function):
    """y = x W^T (+ b) over n ~ 1e5..1e6 points.  Forward and the input gradient are ordinary GEMMs (rocBLAS through torch);
    the WEIGHT gradient X^T dY is a [<=256 x <=256] result reduced over all n rows, which the GEMM library runs at a few
    TFLOP/s (0.8 ms per layer, 27 layers per training step) — it goes through the pair-list kernel K10 with the identity
    pairing instead."""

    @staticmethod
    def forward(ctx, x, weight, bias, row_add=None, plan=None):
        """`row_add` f32 [g, c_out] with `plan` (segment plan of the row -> group index): y += row_add[plan.inv] — the per-group
        half of a Linear over cat([point, group[inv]], 1); its adjoint is the segmented sum of grad over the plan."""
        ctx.save_for_backward(x, weight)
        ctx.has_bias, ctx.plan = bias is not None, plan
        if _train_wide(x, weight.size(1), weight.size(0)):  # the heads' wide layers: K22h (both operands as f16 hi | lo planes, 3 passes)
            y = hip_ops.linear_planes_norm_act(hip_ops.rows_to_planes(x), hip_ops.linear_prepare_weight_f16(weight), weight.size(0), 128,
                                               bias=bias)
            return hip_ops.gather_rows_add(row_add, plan.inv, y) if row_add is not None else y
        if _train_k22(x, weight.size(0)):  # the product on K22 (split-bf16 matrix cores, fp32-accurate; no norm, no activation)
            return hip_ops.linear_norm_act(x, hip_ops.linear_prepare_weight(weight), weight.size(0), bias=bias,
                                           row_add=row_add, row_add_index=plan.inv if row_add is not None else None)
        y = F.linear(x, weight, bias)
        return hip_ops.gather_rows_add(row_add, plan.inv, y) if row_add is not None else y

    @staticmethod
    def backward(ctx, grad):
        x, weight = ctx.saved_tensors
        grad = grad.contiguous()
        g_x = None
        if ctx.needs_input_grad[0]:
            if _train_wide(grad, weight.size(0), weight.size(1)):
                g_x = hip_ops.linear_planes_norm_act(hip_ops.rows_to_planes(grad), hip_ops.linear_prepare_weight_f16(weight.t().contiguous()),
                                                     weight.size(1), 128)
            elif _train_k22(grad, weight.size(1)):
                g_x = hip_ops.linear_norm_act(grad, hip_ops.linear_prepare_weight(weight.t()), weight.size(1))
            else:
                g_x = grad @ weight
        g_w = None
        if ctx.needs_input_grad[1]:
            cin, cout = x.size(1), grad.size(1)
            # the kernel moves 16-byte chunks: pad the channel counts (one extra pass, still several times cheaper than the
            # library's [c, n] x [n, k] product: 1.2 ms for the 3 -> 16 layer over 4.9e5 points)
            xp = F.pad(x, (0, 4 - cin % 4)) if cin % 4 else x
            gp = F.pad(grad, (0, 4 - cout % 4)) if cout % 4 else grad
            g_w = hip_ops.linear_backward_weight(xp, gp)[:cin, :cout].t()
        g_b = hip_ops.column_sum(grad) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        g_t = hip_ops.segment_reduce(grad, ctx.plan, "sum") if (len(ctx.needs_input_grad) > 3 and ctx.needs_input_grad[3]) else None
        return g_x, g_w, g_b, g_t, None


_TRAIN_WIDE_MIN_ROWS = 4096  # training: rows from which the heads' wide products run on K22h rather than on the library


def _train_wide(x, k, c):
    """Training: a [>= 4096 rows, k] x [k, c] product of the query / refine heads (k, c >= 256: `shared_mlp_dims`, `embed_dims`) on K22h —
    the library's fp32 GEMMs run these at ~100 TFLOP/s, 36 of them per step."""
    return (switches.K22H and x.dim() == 2 and x.size(0) >= _TRAIN_WIDE_MIN_ROWS and k % 32 == 0 and k >= 256 and c >= 256 and c % 4 == 0
            and x.size(1) == k and hip_ops.rows_to_planes_supported(x))


def _train_k22(x, out_features):
    return (x.size(0) >= 16384 and x.size(1) % 4 == 0 and x.size(1) >= 32
            and hip_ops.linear_norm_act_supported(x, out_features))



# ---- K22h: the wide Linears of the query / refine heads on f16 x 3 planes ------------------------------------------------
def _rows_of(x):
    return x.n if isinstance(x, hip_ops.RowPlanes) else x.size(0)


def wide_linear_supported(linear, x):
    """Does `linear(x)` take K22h (fsf_linear_planes_norm_act)?  Inference, >= K22H_MIN_ROWS rows, whole 32-wide k chunks, at least 256
    channels either side (the `shared_mlp_dims=[1024, 1024]` / `embed_dims=1024` layers: bound by K22's six matrix passes)."""
    if not (switches.K22H and isinstance(linear, nn.Linear) and linear.in_features % 32 == 0 and linear.in_features >= 256
            and linear.out_features >= 256 and linear.out_features % 4 == 0 and linear.weight.is_cuda
            and linear.weight.dtype == torch.float32):
        return False
    if torch.is_grad_enabled() and linear.weight.requires_grad:
        return False
    if isinstance(x, hip_ops.RowPlanes):
        return x.c == linear.in_features
    return (torch.is_tensor(x) and x.dim() == 2 and x.size(0) >= switches.K22H_MIN_ROWS and x.size(1) == linear.in_features
            and not (torch.is_grad_enabled() and x.requires_grad) and hip_ops.rows_to_planes_supported(x))


def as_row_planes(x):
    """x (f32 rows or RowPlanes) -> RowPlanes; the conversion of a tensor is kept on it (two consumers of the LiDAR query features)."""
    if isinstance(x, hip_ops.RowPlanes):
        return x
    c = x.__dict__.get("_fsf_row_planes") if hasattr(x, "__dict__") else None
    if c is not None and c[1] == x._version and c[2] == x.data_ptr():
        return c[0]
    rp = hip_ops.rows_to_planes(x)
    try:
        x._fsf_row_planes = (rp, x._version, x.data_ptr())
    except AttributeError:
        pass
    return rp


def _prepared_planes_f16(linear, slice_c=128):
    key = (linear.weight.data_ptr(), linear.weight._version, linear.weight.device, slice_c)
    cache = linear.__dict__.get("_fsf_planes_f16")
    if cache is None or cache[0] != key:
        cache = (key, hip_ops.linear_prepare_weight_f16(linear.weight, slice_c))
        linear.__dict__["_fsf_planes_f16"] = cache
    return cache[1]


def _wide_linear_norm_act(linear, norm, act, x, out=None, planes_out=False):
    """act(norm(linear(x))) on K22h; returns f32 rows, or RowPlanes with `planes_out` (the next wide layer's operand: the LayerNorm
    + activation pass writes the planes instead of fp32 rows).  None when the norm / activation pair is not covered."""
    act_code = "relu" if isinstance(act, nn.ReLU) else (
        "gelu" if isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none" else None)
    ln = isinstance(norm, nn.LayerNorm) and len(norm.normalized_shape) == 1 and norm.elementwise_affine
    if act_code is None or not ln:
        return None
    c = linear.out_features
    y = hip_ops.linear_planes_norm_act(as_row_planes(x), _prepared_planes_f16(linear), c, 128, bias=linear.bias)
    if planes_out and out is None and hip_ops.rows_to_planes_supported(y):
        return hip_ops.rows_to_planes(y, "ln", norm.weight, norm.bias, norm.eps, act_code)
    return hip_ops.norm_act(y, norm.weight, norm.bias, norm.eps, "ln", act_code, out=out)


def materialize_rows(x):
    """RowPlanes reached a consumer that wants fp32 rows (not on the built paths: producers only emit planes for a wide consumer —
    but a head whose first layer has another shape, or a switch flipped between producer and consumer, lands here): the rows back
    from the planes, `RowPlanes.rows()`."""
    if isinstance(x, hip_ops.RowPlanes):
        return x.rows()
    return x


def linear_norm_act(linear, norm, act, x, out=None, planes_out=False):
    """`act(norm(linear(x)))` for the [Linear, norm, act] blocks applied to every point / cluster row.  At inference, for
    up to 128 output channels with LayerNorm or eval-mode BatchNorm1d and ReLU / exact GELU, this is ONE HIP kernel
    (fsf_linear_norm_act, K22: the product from an exact 3-way bf16 split on the bf16 matrix cores — fp32-accurate — with
    the norm and the activation as its epilogue); otherwise the GEMM runs on the library and `fused_norm_act` follows."""
    if wide_linear_supported(linear, x):
        y = _wide_linear_norm_act(linear, norm, act, x, out=out, planes_out=planes_out)
        if y is not None:
            return y
    x = materialize_rows(x)
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or linear.weight.requires_grad)
    act_code = "relu" if isinstance(act, nn.ReLU) else (
        "gelu" if isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none" else None)
    if (not needs_grad and act_code is not None and isinstance(linear, nn.Linear) and x.dim() == 2 and x.size(0) >= 1024
            and x.is_cuda and x.dtype == torch.float32 and linear.in_features <= 64
            and not hip_ops.linear_norm_act_supported(x, linear.out_features) and linear.out_features % 4 == 0):
        k = x.size(1)  # thin inputs ([n, 10] image features, [n, 11] VFE decorations): one small copy buys aligned rows
        x = F.pad(x, (0, (-k) % 4))[:, :k]
    # (a few hundred rows — the camera queries — with a shallow product: the kernel is latency-bound at ~10 us, the
    # library spends 60-150 us of host time per call choosing a GEMM; deep products on few rows stay on the library)
    if (not needs_grad and act_code is not None and isinstance(linear, nn.Linear) and x.dim() == 2
            and (x.size(0) >= 1024 or (x.size(0) >= _SMALL_N_MIN and linear.in_features <= 256))
            and hip_ops.linear_norm_act_supported(x, linear.out_features)
            and (out is None or (out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0))):
        kind = None
        if isinstance(norm, nn.LayerNorm) and len(norm.normalized_shape) == 1 and norm.elementwise_affine:
            kind, gamma, beta, eps = "ln", norm.weight, norm.bias, norm.eps
        elif isinstance(norm, nn.BatchNorm1d) and not norm.training and norm.track_running_stats:
            from .spconv import _bn_affine

            kind, eps = "affine", 0.0
            gamma, beta = _bn_affine(norm)
        if kind is not None:
            planes = _prepared_planes(linear)
            if kind == "ln" and linear.out_features > 128:  # LayerNorm statistics span the kernel's 128-channel slices
                y = hip_ops.linear_norm_act(x, planes, linear.out_features, bias=linear.bias)
                return fused_norm_act(y, norm, act, out=out)
            return hip_ops.linear_norm_act(x, planes, linear.out_features, bias=linear.bias, norm=kind, gamma=gamma,
                                           beta=beta, eps=eps, act=act_code, out=out)
    return fused_norm_act(point_linear(linear, x), norm, act, out=out)


def _prepared_planes(linear):
    """The layer's weight as K22's fragment-ordered bf16 planes, prepared once per weight version."""
    key = (linear.weight.data_ptr(), linear.weight._version, linear.weight.device)
    cache = linear.__dict__.get("_fsf_planes")
    if cache is None or cache[0] != key:
        cache = (key, hip_ops.linear_prepare_weight(linear.weight))
        linear.__dict__["_fsf_planes"] = cache
    return cache[1]


def point_linear(linear, x):
    """`linear(x)` for a per-point nn.Linear; training on the GPU routes the weight gradient through K10, inference on
    >= 1024 rows runs the product on K22 (no norm, no activation: 10 641 x 1024 -> 1024 158 us vs 225 us on the library)."""
    if wide_linear_supported(linear, x):  # (K22h: 10 641 x 1024 -> 1024 on f16 x 3 planes)
        return hip_ops.linear_planes_norm_act(as_row_planes(x), _prepared_planes_f16(linear), linear.out_features, 128, bias=linear.bias)
    x = materialize_rows(x)
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or linear.weight.requires_grad)
    if (needs_grad and linear.weight.requires_grad and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
            and (x.size(0) >= 16384 or _train_wide(x, linear.in_features, linear.out_features))):
        return _PointLinearFn.apply(x, linear.weight, linear.bias)
    if (not needs_grad and x.dim() == 2 and linear.out_features % 4 == 0
            and (x.size(0) >= 1024 or (x.size(0) >= _SMALL_N_MIN and linear.in_features <= 256))
            and hip_ops.linear_norm_act_supported(x, linear.out_features)):
        return hip_ops.linear_norm_act(x, _prepared_planes(linear), linear.out_features, bias=linear.bias)
    return F.linear(x, linear.weight, linear.bias)


_IDENTITY_ROWS = {}


def _identity_rows(n, device):
    """arange(n) i64 on `device` (grow-only, per device and stream owner thread): the row index of `point_linear_add`'s addend."""
    key = (device, threading.get_ident())
    t = _IDENTITY_ROWS.get(key)
    if t is None or t.numel() < n:
        t = _IDENTITY_ROWS[key] = torch.arange(max(n, 1 << 16), dtype=torch.int64, device=device)
    return t[:n]


def point_linear_add(linear, x, addend):
    """`linear(x) + addend` for a per-point nn.Linear whose width is NOT a multiple of 4 (the 131-wide image feature update of
    FSF.segmentor_feat_inhance_test, FSF.py:789-792) in ONE K22 launch: the weight padded with zero rows to the next multiple of 4,
    the addend — rows of a buffer padded the same way, as the neck emits them — added in the kernel's epilogue, the result a
    `[:, :c]` view of a padded buffer (a legal K22 operand for the next layer).  Returns None when the shapes are not covered
    (the caller then runs the library GEMM and the add)."""
    c = linear.out_features
    cpad = (c + 3) // 4 * 4
    n = x.size(0)
    if (torch.is_grad_enabled() and (x.requires_grad or linear.weight.requires_grad or addend.requires_grad)) or n < 1024:
        return None
    if not (x.is_cuda and x.dim() == 2 and addend.dim() == 2 and addend.shape == (n, c) and addend.dtype == torch.float32
            and addend.stride(1) == 1 and addend.stride(0) == cpad and addend.data_ptr() % 16 == 0
            and hip_ops.linear_norm_act_supported(x, cpad)):
        return None
    key = (linear.weight.data_ptr(), linear.weight._version, None if linear.bias is None else (linear.bias.data_ptr(), linear.bias._version),
           linear.weight.device)
    cache = linear.__dict__.get("_fsf_planes_padded")
    if cache is None or cache[0] != key:
        with torch.no_grad():
            w = linear.weight.new_zeros((cpad, linear.in_features))
            w[:c] = linear.weight
            b = linear.weight.new_zeros((cpad,))
            if linear.bias is not None:
                b[:c] = linear.bias
            cache = (key, hip_ops.linear_prepare_weight(w), b)
        linear.__dict__["_fsf_planes_padded"] = cache
    rows = torch.as_strided(addend, (n, cpad), (cpad, 1))  # the padded buffer behind the view (its last columns are never used)
    out = torch.empty((n, cpad), dtype=torch.float32, device=x.device)
    hip_ops.linear_norm_act(x, cache[1], cpad, bias=cache[2], out=out, row_add=rows, row_add_index=_identity_rows(n, x.device))
    return out[:, :c]


class PointLinear(nn.Linear):
    """nn.Linear (same parameters, same state-dict keys) whose training-time weight gradient takes the K10 route."""

    def forward(self, x):
        return point_linear(self, x)


class MLPBlock(nn.Sequential):
    """[Linear, norm, act(, Dropout)] with the same child names ('0', '1', '2'[, '3']) as the reference's
    nn.Sequential, so state-dict keys are unchanged; forward fuses norm + act."""

    def forward(self, x, planes_out=False):
        mods = list(self._modules.values())
        x = linear_norm_act(mods[0], mods[1], mods[2], x, planes_out=planes_out and len(mods) == 3)
        for m in mods[3:]:
            x = m(x)
        return x


class MLPSequential(nn.Sequential):
    """The nn.Sequential `build_mlp` returns (same child names, same state-dict keys).  Inference: a block whose consumer is a wide
    Linear (K22h) hands its output over in plane form — written by its own LayerNorm + activation pass instead of fp32 rows;
    `planes_out=True` asks the same of the LAST block (a caller whose next step is a wide layer: the cluster heads' branches)."""

    def forward(self, x, planes_out=False):
        mods = list(self._modules.values())
        for i, m in enumerate(mods):
            if isinstance(m, MLPBlock):
                if i + 1 < len(mods):
                    nxt = mods[i + 1]
                    lin = nxt[0] if isinstance(nxt, MLPBlock) else nxt
                    want = (isinstance(lin, nn.Linear) and wide_linear_supported(m[0], x) and m[0].out_features == lin.in_features
                            and _wide_consumer(lin, _rows_of(x)))
                else:
                    want = planes_out
                x = m(x, planes_out=want)
            else:
                x = m(x)
        return x


def _wide_consumer(linear, n_rows):
    """Would `linear` take a RowPlanes input of `n_rows` rows (wide_linear_supported without an input tensor)?"""
    return (switches.K22H and isinstance(linear, nn.Linear) and linear.in_features % 32 == 0 and linear.in_features >= 256
            and linear.out_features >= 256 and linear.out_features % 4 == 0 and n_rows >= switches.K22H_MIN_ROWS
            and linear.weight.is_cuda and linear.weight.dtype == torch.float32
            and not (torch.is_grad_enabled() and linear.weight.requires_grad))


def build_mlp(in_channel, hidden_dims, norm_cfg, is_head=False, act="relu", bias=False, dropout=0):
    """Sequential of [Linear(bias) -> norm -> act (-> Dropout)] blocks; with `is_head` the last entry is a
    plain Linear(bias=True).  Dense GEMMs stay on rocBLAS through torch."""
    layers = []
    last = in_channel
    for i, c in enumerate(hidden_dims):
        if is_head and i == len(hidden_dims) - 1:
            layers.append(PointLinear(last, c, bias=True))
        else:
            block = [PointLinear(last, c, bias=bias), build_norm_layer(norm_cfg, c)[1], get_activation_layer(act, c)]
            if dropout > 0:
                block.append(nn.Dropout(dropout))
            layers.append(MLPBlock(*block))
        last = c
    return MLPSequential(*layers)
