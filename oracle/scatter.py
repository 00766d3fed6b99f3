"""Oracle: scatter_v2 / torch.unique / torch_scatter semantics.  TEST INFRASTRUCTURE ONLY.

`scatter_v2` follows projects/mmdet3d_plugin/ops/sst_ops.py:150-177 line by line; the un-vendored
torch_scatter 2.0.2 calls inside it are restated with `Tensor.scatter_reduce` / `index_add_`
(mean = sum / clamp(count, 1); scatter_max -> (out, argmax), empty rows -> 0 / n).
Pinned: tests/golden/scatter_v2_*.npz were produced by the reference's own `scatter_v2` (imported in the build
container with the shim described in SURVEY.md App. D) and this file is checked against them.
"""
import torch


def unique_rows(coors, return_counts=True):
    """torch.unique(coors, return_inverse=True, return_counts=True, dim=0) (sst_ops.py:156)."""
    coors = torch.as_tensor(coors)
    return torch.unique(coors, return_inverse=True, return_counts=return_counts, dim=0)


def segment_sum(feat, inv, m):
    out = torch.zeros((m, feat.size(1)), dtype=feat.dtype)
    out.index_add_(0, inv, feat)
    return out


def segment_mean(feat, inv, m):
    s = segment_sum(feat, inv, m)
    cnt = torch.bincount(inv, minlength=m).clamp(min=1).to(feat.dtype)
    return s / cnt[:, None]


def segment_max(feat, inv, m):
    """torch_scatter.scatter_max: (out, arg); arg = first row index attaining the max (ties: the smallest
    index here; upstream's choice is atomics-order dependent), empty segments -> (0, n)."""
    n, c = feat.shape
    out = torch.full((m, c), float("-inf"), dtype=feat.dtype)
    out = out.scatter_reduce(0, inv[:, None].expand(n, c), feat, reduce="amax", include_self=True)
    is_max = feat == out[inv]
    cand = torch.where(is_max, torch.arange(n)[:, None].expand(n, c), torch.full((n, c), n))
    arg = torch.full((m, c), n, dtype=torch.int64)
    arg = arg.scatter_reduce(0, inv[:, None].expand(n, c), cand, reduce="amin", include_self=True)
    empty = torch.bincount(inv, minlength=m) == 0
    out = torch.where(empty[:, None], torch.zeros_like(out), out)
    return out, arg


def scatter_v2(feat, coors, mode, return_inv=True, min_points=0, unq_inv=None, new_coors=None):
    """sst_ops.py:150-177."""
    feat = torch.as_tensor(feat)
    coors = torch.as_tensor(coors)
    assert feat.size(0) == coors.size(0)
    if mode == "avg":
        mode = "mean"
    if unq_inv is None:
        new_coors, unq_inv, unq_cnt = torch.unique(coors, return_inverse=True, return_counts=True, dim=0)
    else:
        assert new_coors is not None
    if min_points > 0:
        cnt_per_point = unq_cnt[unq_inv]
        valid_mask = cnt_per_point >= min_points
        feat = feat[valid_mask]
        coors = coors[valid_mask]
        new_coors, unq_inv, unq_cnt = torch.unique(coors, return_inverse=True, return_counts=True, dim=0)
    m = new_coors.size(0)
    if mode == "max":
        new_feat, _ = segment_max(feat, unq_inv, m)
    elif mode == "mean":
        new_feat = segment_mean(feat, unq_inv, m)
    elif mode == "sum":
        new_feat = segment_sum(feat, unq_inv, m)
    else:
        raise NotImplementedError
    if not return_inv:
        return new_feat, new_coors
    return new_feat, new_coors, unq_inv


def ingroup_rank(group_inds):
    """Contract of TorchEx ingroup_indices (sst_ops.py:225-235): per group a permutation of 0..n_g-1.
    This oracle returns the stable rank (ascending original index)."""
    g = torch.as_tensor(group_inds).long()
    order = torch.argsort(g, stable=True)
    sg = g[order]
    n = g.numel()
    start = torch.zeros(n, dtype=torch.int64)
    if n:
        head = torch.ones(n, dtype=torch.bool)
        head[1:] = sg[1:] != sg[:-1]
        pos = torch.arange(n)
        start = torch.cummax(torch.where(head, pos, torch.zeros_like(pos)), 0)[0]
    out = torch.empty(n, dtype=torch.int64)
    out[order] = torch.arange(n) - start
    return out
