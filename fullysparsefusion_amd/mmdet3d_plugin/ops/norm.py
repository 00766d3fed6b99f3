"""`naiveSyncBN1d` (NORM_LAYERS) — defined in the authors' mmdet3d fork [UNVENDORED], selected by
projects/configs/nuScenes/FSF_nuScenes_config.py:50,63,85.

Upstream behaviour: world_size == 1 or eval -> plain BatchNorm1d; otherwise per-rank
`mean` and `mean of squares` are all-reduced (SUM) and divided by world size — every rank weighted equally
regardless of its row count — and the backward all-reduces the matching gradients.  The collective runs on
whatever backend torch.distributed was initialised with: RCCL over xGMI on the GPU box ("nccl"), gloo in the
CPU tests.  The two [C] statistics travel as ONE packed [2C] message (latency-bound).
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from ..registry import NORM_LAYERS


class _SyncStats(torch.autograd.Function):
    """all-reduce(mean) of a packed [2C] statistics vector; backward all-reduces the gradient the same way."""

    @staticmethod
    def forward(ctx, packed):
        out = packed.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        return out / dist.get_world_size()

    @staticmethod
    def backward(ctx, grad):
        g = grad.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g / dist.get_world_size()


@NORM_LAYERS.register_module("naiveSyncBN1d")
class NaiveSyncBatchNorm1d(nn.BatchNorm1d):
    def forward(self, x):
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not (self.training and distributed):
            return super().forward(x)
        assert x.dim() == 2, "naiveSyncBN1d expects [rows, C]"
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
