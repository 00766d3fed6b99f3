"""The BACKWARD half of BASELINE config 3 ("fwd+bwd") and config 4 ("bs = 2 per GPU") at the size bench.py times.

One training-mode forward + backward of `FSF.forward_train`'s graph (`FSF.forward_train_graph`: segmentor + fusion, camera / LiDAR
queries with their heads, query combination, the refine stage — FSF.py:806-903, :905-1044, fsd_bbox_head.py:96-197 — + bench.py's
dummy scalar loss = the sum of all head outputs: the step `bench.py --train` times) on the full 10-sweep frame and on the
Argoverse-2-shape frame (config 5), with EVERY autograd node of the HIP path checked in situ against a float64 restatement
evaluated on the very tensors the node received:

  * `_SparseConvFn` (34 layers): forward (K9c / K9b / fp32 kernel, whichever the dispatch picked), data gradient (the same kernels
    over the transposed rulebook) and weight gradient (K10 over the spconv-v1 pair lists);
  * `_NormActFn` (K12): act(LayerNorm(x)) backward — grad_x, grad_gamma, grad_beta;
  * `_BatchNormActFn` (K23): training-mode BatchNorm1d (+ ReLU) forward and backward through the batch statistics;
  * `_PointLinearFn`: forward product (K22 / library), input gradient, weight gradient (K10 identity pairing), bias gradient and
    the adjoint of the per-group addend (`fsf_gather_rows_add`'s adjoint: a segmented sum);
  * `_SegmentReduce` (max / mean / sum) forward and backward, `_GatherRows` forward and its segmented-sum adjoint;
  * `_SirProductFn` (K28): SIRLayer's concatenations + product with the position MLP, forward and the three adjoints.

The float64 restatements are plain torch on the device (index_add_, matmul, F.layer_norm / F.batch_norm under autograd) — the
same role `oracle/` plays for the forward; nothing here calls the HIP library to produce an expected value.
An end-to-end gradient comparison at this size is not meaningful (ReLU inputs within rounding of zero flip between any two
implementations and shift every gradient downstream, see test_plugin_gpu.py::test_stage1_gradients_vs_oracle): the per-node
checks are what carries the backward, as in round 2's 12 000-point test, now at 310 615 points.
"""
import collections

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


class _Checker:
    def __init__(self):
        self.seen = collections.defaultdict(list)  # kind -> [(shape, relative error)]

    def note(self, kind, shape, got, want, tol, denom=None):
        got, want = got.double(), want.double()
        scale = float(want.abs().max()) if denom is None else float(denom)
        err = float((got - want).abs().max()) / max(scale, 1e-30)
        self.seen[kind].append((tuple(shape), err))
        assert err <= tol, (kind, tuple(shape), err, tol)


def _install(monkeypatch, ck):
    from fullysparsefusion_amd.mmdet3d_plugin.ops import spconv as sp
    from fullysparsefusion_amd.mmdet3d_plugin.ops import sst_ops as so

    # ----------------------------------------------------------------------------------------------- sparse convolution
    conv_fwd, conv_bwd = sp._SparseConvFn.forward, sp._SparseConvFn.backward

    def conv_ref(feat, weight, table):
        kvol = table.size(1)
        w = weight.detach().reshape(kvol, weight.shape[-2], weight.shape[-1]).double()
        f64 = feat.detach().double()
        out = torch.zeros((table.size(0), w.size(2)), dtype=torch.float64, device=feat.device)
        for k in range(kvol):
            o = (table[:, k] >= 0).nonzero().squeeze(1)
            if o.numel():
                out.index_add_(0, o, f64[table[o, k].long()] @ w[k])
        return out

    def conv_forward(ctx, feat, weight, rb, inverse, split, planes=False):
        out = conv_fwd(ctx, feat, weight, rb, inverse, split, planes)
        ck.note("spconv.forward", (out.shape, feat.shape[1]), out, conv_ref(feat, weight, rb.table(inverse)), 1e-5)
        return out

    def conv_backward(ctx, grad):
        res = conv_bwd(ctx, grad)
        g_feat, g_w = res[:2]
        feat, weight = ctx.saved_tensors
        table = ctx.rb.table(ctx.inverse).long()
        kvol = table.size(1)
        w = weight.detach().reshape(kvol, weight.shape[-2], weight.shape[-1]).double()
        g64, f64 = grad.double(), feat.detach().double()
        want_feat = torch.zeros(feat.shape, dtype=torch.float64, device=feat.device)
        want_w = torch.zeros_like(w)
        for k in range(kvol):
            o = (table[:, k] >= 0).nonzero().squeeze(1)
            i = table[o, k]
            want_feat.index_add_(0, i, g64[o] @ w[k].t())
            want_w[k] = f64[i].t() @ g64[o]
        if g_feat is not None:
            ck.note("spconv.grad_input", (feat.shape, grad.shape[1]), g_feat, want_feat, 1e-5)
        if g_w is not None:
            ck.note("spconv.grad_weight", (feat.shape, grad.shape[1]), g_w.reshape(w.shape), want_w, 1e-5)
        return res

    monkeypatch.setattr(sp._SparseConvFn, "forward", staticmethod(conv_forward))
    monkeypatch.setattr(sp._SparseConvFn, "backward", staticmethod(conv_backward))

    # ------------------------------------------------------------------------------------------ LayerNorm + activation
    na_bwd = so._NormActFn.backward

    def norm_act_backward(ctx, grad):
        res = na_bwd(ctx, grad)
        x, gamma, beta = ctx.saved_tensors
        with torch.enable_grad():
            x64 = x.detach().double().requires_grad_(True)
            g64, b64 = gamma.detach().double().requires_grad_(True), beta.detach().double().requires_grad_(True)
            pre = F.layer_norm(x64, (x.size(1),), g64, b64, ctx.eps)
            y = F.gelu(pre) if ctx.act_code == "gelu" else torch.relu(pre)
            wx, wg, wb = torch.autograd.grad(y, [x64, g64, b64], grad.double())
        ok = torch.ones_like(pre, dtype=torch.bool) if ctx.act_code == "gelu" else (pre.detach().abs() > 1e-5)
        ok = ok.all(1, keepdim=True)  # (a row holding a ReLU input within rounding of zero: the flip moves its whole row's grad_x)
        ck.note("norm_act.grad_x", x.shape, res[0] * ok, wx * ok, 2e-5)
        # column sums over n rows of fp32 products: compared against the magnitude that was summed
        ck.note("norm_act.grad_gamma", x.shape, res[1], wg, 2e-5, denom=float((grad.double().abs() * 4).sum(0).max()))
        ck.note("norm_act.grad_beta", x.shape, res[2], wb, 2e-5, denom=float(grad.double().abs().sum(0).max()))
        return res

    monkeypatch.setattr(so._NormActFn, "backward", staticmethod(norm_act_backward))

    # --------------------------------------------------------------------------- training-mode BatchNorm1d (+ ReLU), K23
    bn_fwd, bn_bwd = so._BatchNormActFn.forward, so._BatchNormActFn.backward

    def bn_ref(x, weight, bias, eps, relu, grad=None):
        with torch.enable_grad():
            x64 = x.detach().double().requires_grad_(True)
            w64 = weight.detach().double().requires_grad_(True) if weight is not None else None
            b64 = bias.detach().double().requires_grad_(True) if bias is not None else None
            pre = F.batch_norm(x64, None, None, w64, b64, True, 0.0, eps)
            y = torch.relu(pre) if relu else pre
            if grad is None:
                return y.detach(), pre.detach()
            gs = torch.autograd.grad(y, [x64] + ([w64, b64] if w64 is not None else []), grad.double())
        return gs, pre.detach()

    def bn_forward(ctx, x, weight, bias, bn, relu):
        out = bn_fwd(ctx, x, weight, bias, bn, relu)
        ctx._chk = (weight, bias, bn.eps)
        want, _ = bn_ref(x, weight, bias, bn.eps, relu)
        ck.note("batch_norm.forward", x.shape, out, want, 1e-5)
        return out

    def bn_backward(ctx, grad):
        res = bn_bwd(ctx, grad)
        x = ctx.saved_tensors[0]
        weight, bias, eps = ctx._chk
        gs, pre = bn_ref(x, weight, bias, eps, ctx.relu, grad)
        # a ReLU input within rounding of zero may flip: such elements are left out of the element-wise comparison (their
        # effect on the column statistics is far below the tolerance)
        ok = (pre.abs() > 1e-5) if ctx.relu else torch.ones_like(pre, dtype=torch.bool)
        ck.note("batch_norm.grad_x", x.shape, res[0] * ok, gs[0] * ok, 2e-5)
        if weight is not None:
            # column sums of ~3e5 same-sign fp32 terms (the dummy loss makes every grad element equal): blocked fp32 summation is
            # good to ~1e-5 of the summed magnitude (measured 1.1e-5 ... 1.7e-5 on grad_beta)
            ck.note("batch_norm.grad_gamma", x.shape, res[1], gs[1], 1e-4, denom=float((grad.double().abs() * pre.abs()).sum(0).max()))
            ck.note("batch_norm.grad_beta", x.shape, res[2], gs[2], 1e-4, denom=float(grad.double().abs().sum(0).max()))
        return res

    monkeypatch.setattr(so._BatchNormActFn, "forward", staticmethod(bn_forward))
    monkeypatch.setattr(so._BatchNormActFn, "backward", staticmethod(bn_backward))

    # ------------------------------------------------------------------------------------------------ per-point Linear
    pl_fwd, pl_bwd = so._PointLinearFn.forward, so._PointLinearFn.backward

    def pl_forward(ctx, x, weight, bias, row_add=None, plan=None):
        out = pl_fwd(ctx, x, weight, bias, row_add, plan)
        rows = torch.randint(0, x.size(0), (4096,), device=x.device)
        want = F.linear(x.detach()[rows].double(), weight.detach().double(), None if bias is None else bias.detach().double())
        if row_add is not None:
            want = want + row_add.detach().double()[plan.inv[rows]]
        ck.note("point_linear.forward", (x.shape, weight.shape[0]), out[rows], want, 1e-5)
        return out

    def pl_backward(ctx, grad):
        res = pl_bwd(ctx, grad)
        x, weight = ctx.saved_tensors
        g64 = grad.double()
        if res[0] is not None:
            ck.note("point_linear.grad_input", (x.shape, weight.shape[0]), res[0], g64 @ weight.detach().double(), 1e-5)
        if res[1] is not None:
            # (a sum over n rows of fp32 products per entry, against the largest entry: 1.06e-5 measured on the [32 793, 1024] -> 1024
            # head layer of the two-frame batch — the rounding of 3e4 fp32 accumulations, K10p's split products are exact)
            ck.note("point_linear.grad_weight", (x.shape, weight.shape[0]), res[1], g64.t() @ x.detach().double(), 2e-5)
        if res[2] is not None:
            ck.note("point_linear.grad_bias", (x.shape, weight.shape[0]), res[2], g64.sum(0), 2e-5, denom=float(g64.abs().sum(0).max()))
        if res[3] is not None:
            want = torch.zeros((ctx.plan.m, grad.size(1)), dtype=torch.float64, device=grad.device).index_add_(0, ctx.plan.inv, g64)
            denom = torch.zeros_like(want).index_add_(0, ctx.plan.inv, g64.abs())
            ck.note("point_linear.grad_row_add", (x.shape, weight.shape[0]), res[3], want, 2e-5, denom=float(denom.max()))
        return res

    monkeypatch.setattr(so._PointLinearFn, "forward", staticmethod(pl_forward))
    monkeypatch.setattr(so._PointLinearFn, "backward", staticmethod(pl_backward))

    # ------------------------------------------------------------------------------- segmented reductions and gathers
    sr_fwd, sr_bwd = so._SegmentReduce.forward, so._SegmentReduce.backward

    def seg_ref(feat, plan, mode):
        m, c = plan.m, feat.size(1)
        idx = plan.inv[:, None].expand(-1, c)
        f64 = feat.detach().double()
        if mode == "max":
            return torch.full((m, c), -float("inf"), dtype=torch.float64, device=feat.device).scatter_reduce(0, idx, f64, "amax")
        s = torch.zeros((m, c), dtype=torch.float64, device=feat.device).index_add_(0, plan.inv, f64)
        if mode == "sum":
            return s
        cnt = torch.bincount(plan.inv, minlength=m).clamp(min=1).double()
        return s / cnt[:, None]

    def sr_forward(ctx, feat, plan, mode, short=False):
        out = sr_fwd(ctx, feat, plan, mode, short)
        want = seg_ref(feat, plan, mode)
        if mode == "max":
            assert torch.equal(out.double(), want), ("segment max differs", tuple(feat.shape))
            ck.seen["segment_reduce.forward.max"].append((tuple(feat.shape), 0.0))
        else:
            ck.note("segment_reduce.forward." + mode, feat.shape, out, want, 1e-5)
        return out

    def sr_backward(ctx, grad_out):
        res = sr_bwd(ctx, grad_out)
        plan, mode = ctx.plan, ctx.mode
        g64 = grad_out.double()
        if mode == "max":
            arg = ctx.saved_tensors[0]
            want = torch.zeros((plan.n, grad_out.size(1)), dtype=torch.float64, device=grad_out.device)
            want.scatter_(0, arg, g64)  # every (segment, channel) sends its gradient to its arg-max row (segments are non-empty)
            assert bool((plan.inv.gather(0, arg.reshape(-1)).reshape(arg.shape) ==
                         torch.arange(plan.m, device=arg.device)[:, None]).all()), "an arg-max row outside its segment"
        else:
            want = g64[plan.inv]
            if mode == "mean":
                want = want / torch.bincount(plan.inv, minlength=plan.m).clamp(min=1).double()[plan.inv][:, None]
        ck.note("segment_reduce.backward." + mode, (plan.n, grad_out.size(1)), res[0], want, 1e-6)
        return res

    monkeypatch.setattr(so._SegmentReduce, "forward", staticmethod(sr_forward))
    monkeypatch.setattr(so._SegmentReduce, "backward", staticmethod(sr_backward))

    gr_fwd, gr_bwd = so._GatherRows.forward, so._GatherRows.backward

    def gr_forward(ctx, src, plan):
        out = gr_fwd(ctx, src, plan)
        assert torch.equal(out, src.detach().float()[plan.inv])
        ck.seen["gather_rows.forward"].append((tuple(out.shape), 0.0))
        return out

    def gr_backward(ctx, grad_out):
        res = gr_bwd(ctx, grad_out)
        g64 = grad_out.double()
        want = torch.zeros((ctx.plan.m, grad_out.size(1)), dtype=torch.float64, device=grad_out.device).index_add_(0, ctx.plan.inv, g64)
        denom = torch.zeros_like(want).index_add_(0, ctx.plan.inv, g64.abs())
        ck.note("gather_rows.backward", grad_out.shape, res[0], want, 2e-5, denom=float(denom.max()))
        return res

    monkeypatch.setattr(so._GatherRows, "forward", staticmethod(gr_forward))
    monkeypatch.setattr(so._GatherRows, "backward", staticmethod(gr_backward))

    # ------------------------------------------------------------------------------------- SIRLayer input product (K28)
    from fullysparsefusion_amd.mmdet3d_plugin.models.voxel_encoders import voxel_encoder as ve

    sp_fwd, sp_bwd = ve._SirProductFn.forward, ve._SirProductFn.backward

    def sp_ref(points, feats, extra, h, normalizer, extra_div):
        nrm = torch.tensor(normalizer, dtype=torch.float64, device=points.device)
        parts = [points[:, :3].double() / nrm[None, :], points[:, 3:].double(), feats]
        if extra is not None:
            parts.append(extra / extra_div)
        return torch.cat(parts, 1) * h

    def sp_forward(ctx, points, feats, extra, h, normalizer, extra_div):
        out = sp_fwd(ctx, points, feats, extra, h, normalizer, extra_div)
        want = sp_ref(points.detach(), feats.detach().double(), None if extra is None else extra.detach().double(), h.detach().double(),
                      normalizer, extra_div)
        ck.note("sir_product.forward", out.shape, out, want, 1e-6)
        return out

    def sp_backward(ctx, grad):
        res = sp_bwd(ctx, grad)
        points, feats, extra, h = ctx.saved_tensors
        with torch.enable_grad():
            f64 = feats.detach().double().requires_grad_(True)
            e64 = extra.detach().double().requires_grad_(True) if extra is not None else None
            h64 = h.detach().double().requires_grad_(True)
            y = sp_ref(points.detach(), f64, e64, h64, ctx.normalizer, ctx.extra_div)
            gs = torch.autograd.grad(y, [f64, h64] + ([e64] if e64 is not None else []), grad.double())
        if res[1] is not None:
            ck.note("sir_product.grad_feats", feats.shape, res[1], gs[0], 1e-6)
        if res[3] is not None:
            ck.note("sir_product.grad_h", h.shape, res[3], gs[1], 1e-6)
        if res[2] is not None:
            ck.note("sir_product.grad_extra", extra.shape, res[2], gs[2], 1e-6)
        return res

    monkeypatch.setattr(ve._SirProductFn, "forward", staticmethod(sp_forward))
    monkeypatch.setattr(ve._SirProductFn, "backward", staticmethod(sp_backward))


def _run_graph_checked(device, monkeypatch, dataset, frames_per_gpu, min_points):
    import bench

    torch.manual_seed(0)
    model = bench.build_model(device, dataset).train()
    _, inp = bench.make_inputs(10, 0, device, frames=frames_per_gpu, dataset=dataset)
    n_pts = sum(int(p.shape[0]) for p in inp["points"])
    assert n_pts > min_points * frames_per_gpu
    ck = _Checker()
    _install(monkeypatch, ck)
    model.zero_grad(set_to_none=True)
    out = model.forward_train_graph(inp["points"], inp["img_metas"], inp["mask_data"], inp["mask_anno"])
    heads = bench.head_outputs(out)
    loss = bench.dummy_loss(out)
    loss.backward()
    torch.cuda.synchronize()
    monkeypatch.undo()
    seen = {k: len(v) for k, v in ck.seen.items()}
    worst = {k: max(e for _, e in v) for k, v in ck.seen.items()}
    print("nodes checked:", seen)
    print("worst relative error per kind:", {k: f"{v:.2e}" for k, v in worst.items()})
    return model, out, heads, ck, seen


def _assert_every_parameter_has_a_gradient(model):
    """The loss is the sum of every head output of `forward_train`'s graph: no parameter of the detector may be left without a
    gradient (round 4's loss stopped at the query features: 15.0 M of the 87.0 M parameters never saw one)."""
    missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing[:8]
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    for prefix in ("bbox_head.", "frustum_obj_head.", "combine_frustum_feat_mlp.", "combine_fsd_feat_mlp.", "refine_sir_layers.",
                   "lidar_img_mlp.", "position_encoder.", "out_proj.", "frustum_refined_head.", "refine_img_mlp."):
        norms = [float(p.grad.abs().sum()) for n, p in model.named_parameters() if n.startswith(prefix)]
        assert norms and sum(norms) > 0.0, prefix


@pytest.mark.parametrize("frames_per_gpu", [1, 2])
def test_every_autograd_node_of_the_10sweep_training_step_vs_float64(device, monkeypatch, frames_per_gpu):
    """frames_per_gpu = 1: BASELINE config 3 (10-sweep frame, bs 1, fwd + bwd); 2: config 4's per-rank batch (two distinct
    10-sweep frames in one batch: 6.2e5 points, batch index in every key)."""
    model, out, heads, ck, seen = _run_graph_checked(device, monkeypatch, "nuscenes", frames_per_gpu, 300000)
    # the graph reaches every head: segmentation (2) + camera / LiDAR query heads (2 x (cls, reg)) + one refine stage (cls, reg)
    assert len(heads) == 2 + 2 * 2 + 2 * model.num_extra_stages and model.num_extra_stages >= 1
    assert out["stage_results"][0]["cls_logits"][0].shape[0] == out["obj_feats"].shape[0] > 1000
    # every kind of node ran, at full size
    assert seen.get("spconv.forward", 0) == 34 and seen.get("spconv.grad_weight", 0) == 34 and seen.get("spconv.grad_input", 0) >= 33
    assert any(s[0][0] > 100000 * frames_per_gpu for s, _ in ck.seen["spconv.grad_weight"])
    for kind in ("norm_act.grad_x", "batch_norm.grad_x", "point_linear.grad_weight", "point_linear.grad_input",
                 "point_linear.grad_row_add", "segment_reduce.backward.max", "segment_reduce.backward.mean", "gather_rows.backward",
                 "sir_product.grad_feats", "sir_product.grad_h", "sir_product.grad_extra"):
        assert seen.get(kind, 0) > 0, (kind, seen)
    assert any(s[0] > 200000 * frames_per_gpu for s, _ in ck.seen["norm_act.grad_x"])
    assert any(s[0][0] > 300000 * frames_per_gpu for s, _ in ck.seen["point_linear.grad_weight"])
    _assert_every_parameter_has_a_gradient(model)


def test_every_autograd_node_of_the_av2_training_step_vs_float64(device, monkeypatch):
    """BASELINE config 5's per-rank step: the Argoverse-2 long-range shape (projects/configs/Argoverse2/FSF_AV2_config.py:84-94 —
    2048^2 x 32 grid, 4-stage 64-channel U-Net, 26 classes, 7 cameras with int32 id planes, no `unique_once`), ~150 k points,
    fwd + bwd of `forward_train`'s graph, every autograd node against float64."""
    model, out, heads, ck, seen = _run_graph_checked(device, monkeypatch, "av2", 1, 120000)
    assert len(heads) == 2 + 2 * 2 + 2 * model.num_extra_stages
    n_conv = seen.get("spconv.forward", 0)
    assert n_conv >= 20 and seen.get("spconv.grad_weight", 0) == n_conv and seen.get("spconv.grad_input", 0) >= n_conv - 1
    for kind in ("norm_act.grad_x", "batch_norm.grad_x", "point_linear.grad_weight", "point_linear.grad_input",
                 "segment_reduce.backward.max", "segment_reduce.backward.mean", "gather_rows.backward", "sir_product.grad_feats"):
        assert seen.get(kind, 0) > 0, (kind, seen)
    _assert_every_parameter_has_a_gradient(model)
