"""`naiveSyncBN1d` (NORM_LAYERS) — defined in the authors' mmdet3d fork [UNVENDORED], selected by
projects/configs/nuScenes/FSF_nuScenes_config.py:50,63,85.

Upstream behaviour: world_size == 1 or eval -> plain BatchNorm1d; otherwise per-rank
`mean` and `mean of squares` are all-reduced (SUM) and divided by world size — every rank weighted equally
regardless of its row count — and the backward all-reduces the matching gradients.  The collective runs on
whatever backend torch.distributed was initialised with: RCCL over xGMI on the GPU box ("nccl"), gloo in the
CPU tests.  The two [C] statistics travel as ONE packed [2C] message (latency-bound).
"""
from ... import switches
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from ..registry import NORM_LAYERS

# Instrumentation of the statistics collectives (C3): `SYNC_LOG` — a list the tests / bench.py install to record every collective
# this module issues, in issue order, as (direction, numel) — and `SYNC_COLLECTIVES` — bench.py's A/B switch: False skips the
# all-reduces on EVERY rank alike (the statistics are then per rank: a timing experiment, never a training mode), which is how
# the exposed time of the 2 x 38 latency-bound [2C] messages per step is measured.
SYNC_LOG = None
SYNC_COLLECTIVES = True


def _all_reduce_stats(t, direction, group=None):
    if SYNC_LOG is not None:
        SYNC_LOG.append((direction, int(t.numel())))
    if SYNC_COLLECTIVES:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    else:
        t.mul_(dist.get_world_size(group))


class _SyncStats(torch.autograd.Function):
    """all-reduce(mean) of a packed [2C] statistics vector; backward all-reduces the gradient the same way."""

    @staticmethod
    def forward(ctx, packed):
        out = packed.clone()
        _all_reduce_stats(out, "fwd")
        return out / dist.get_world_size()

    @staticmethod
    def backward(ctx, grad):
        g = grad.clone()
        _all_reduce_stats(g, "bwd")
        return g / dist.get_world_size()


def _column_stats(x):
    """(mean, biased variance) over the rows: K23's two-pass kernel on the GPU, torch on CPU tensors (gloo tests)."""
    if x.is_cuda and x.dtype == torch.float32:
        from ... import hip_ops

        return hip_ops.column_mean_var(x)
    return x.mean(0), x.var(0, unbiased=False)


class _SyncBatchNormAct(torch.autograd.Function):
    """naiveSyncBN1d (+ the ReLU that follows it) across ranks with the row passes on K23: local two-pass statistics ->
    ONE packed [2C] all-reduce (mean | mean of squares, every rank weighted equally, as upstream) -> fused normalise +
    activate; backward: K23's single-rank backward (which also yields sum(g) and sum(g * xhat)) + ONE packed [2C]
    all-reduce of the statistics' gradients + a per-column affine correction in x:

        gx = gx_local + [w s db/n - w s^2 mu dg/n + G_mu/n] + x [w s^2 dg/n + 2 G_q/n],   (G_mu, G_q) = allreduce(dL/dmu, dL/dq) / W
        dL/dmu = -w s db + (w dg / s) s^3 mu,   dL/dq = -(w dg / s) s^3 / 2            (db = sum g, dg = sum g xhat, local)

    which is exactly what autograd produces for the upstream formulation (tests/test_distributed_cpu.py checks it against
    that formulation at world size 2).  The statistics of a layer depend on the previous layer's normalised output, so the
    per-layer collectives cannot be merged across layers without changing the result; what is packed is the two statistics
    of a layer into one message per direction."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, relu, group):
        x = x.contiguous()
        n, c = x.shape
        world = dist.get_world_size(group)
        mean_l, var_l = _column_stats(x)
        packed = torch.cat([mean_l, var_l + mean_l * mean_l])
        _all_reduce_stats(packed, "fwd", group)
        packed /= world
        mean, meansqr = packed[:c], packed[c:]
        var = meansqr - mean * mean
        with torch.no_grad():
            bn.running_mean += bn.momentum * (mean - bn.running_mean)
            bn.running_var += bn.momentum * (var - bn.running_var)
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
        invstd = torch.rsqrt(var + bn.eps)
        scale = weight.detach() * invstd
        shift = bias.detach() - mean * scale
        if x.is_cuda and x.dtype == torch.float32:
            from ... import hip_ops

            y = hip_ops.batch_norm_act_forward(x, scale, shift, relu)
        else:
            y = torch.addcmul(shift, x, scale)
            y = torch.relu(y) if relu else y
        ctx.save_for_backward(x, mean, invstd, scale, shift, weight)
        ctx.relu, ctx.group, ctx.world = relu, group, world
        return y

    @staticmethod
    def backward(ctx, grad):
        x, mean, invstd, scale, shift, weight = ctx.saved_tensors
        n, c = x.shape
        grad = grad.contiguous()
        if x.is_cuda and x.dtype == torch.float32:
            from ... import hip_ops

            gx, dg, db = hip_ops.batch_norm_act_backward(x, grad, mean, invstd, scale, shift, ctx.relu)
        else:  # the same three quantities with torch ops (CPU tensors: the gloo tests)
            g = grad * (torch.addcmul(shift, x, scale) > 0) if ctx.relu else grad
            xhat = (x - mean) * invstd
            db, dg = g.sum(0), (g * xhat).sum(0)
            gx = scale * (g - db / n - xhat * (dg / n))
        w, s = weight.detach(), invstd
        dl_ds = w * dg / s
        g_stats = torch.cat([-w * s * db + dl_ds * s ** 3 * mean, -0.5 * dl_ds * s ** 3])
        _all_reduce_stats(g_stats, "bwd", ctx.group)
        g_stats /= ctx.world
        c0 = (w * s * db - w * s * s * mean * dg + g_stats[:c]) / n
        c1 = (w * s * s * dg + 2.0 * g_stats[c:]) / n
        gx = torch.addcmul(gx + c0, x, c1)
        return gx, dg, db, None, None, None


@NORM_LAYERS.register_module("naiveSyncBN1d")
class NaiveSyncBatchNorm1d(nn.BatchNorm1d):
    def forward_act(self, x, relu):
        """bn(x) followed by ReLU as one autograd node when the statistics are synced across ranks (K23 does the row passes)."""
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if self.training and distributed and x.dim() == 2 and self.affine and self.track_running_stats:
            return _SyncBatchNormAct.apply(x, self.weight, self.bias, self, bool(relu), None)
        y = self.forward(x)
        return torch.relu(y) if relu else y

    def forward(self, x):
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not (self.training and distributed):
            return super().forward(x)
        assert x.dim() == 2, "naiveSyncBN1d expects [rows, C]"
        if self.affine and self.track_running_stats and switches.SYNCBN_FUSED:
            return _SyncBatchNormAct.apply(x, self.weight, self.bias, self, False, None)
        mean = x.mean(0)
        meansqr = (x * x).mean(0)
        packed = _SyncStats.apply(torch.cat([mean, meansqr]))
        c = x.size(1)
        mean, meansqr = packed[:c], packed[c:]
        var = meansqr - mean * mean
        with torch.no_grad():
            self.running_mean += self.momentum * (mean.detach() - self.running_mean)
            self.running_var += self.momentum * (var.detach() - self.running_var)
            if self.num_batches_tracked is not None:
                self.num_batches_tracked += 1
        invstd = torch.rsqrt(var + self.eps)
        scale = self.weight * invstd
        bias = self.bias - mean * scale
        return x * scale[None, :] + bias[None, :]
