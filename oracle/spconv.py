"""Oracle: spconv v1 rulebooks and sparse convolution.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: mmdet3d.ops.spconv (spconv v1.x) is an un-vendored dependency of the reference
(README.md:23-26; imported at projects/mmdet3d_plugin/ops/sst_ops.py:5).  Restated from the published spconv v1
algorithm (`get_indice_pairs` / `indice_conv`):
  * an input site i feeds output site o through kernel offset k = (kz*KY + ky)*KX + kx iff
    o*stride = i + pad - k*dil (exactly divisible, o inside out_shape);
  * out_shape = (in + 2*pad - dil*(k-1) - 1) // stride + 1;
  * SubMConv3d: output sites == input sites (same rows), only pairs whose output site is active;
  * SparseConv3d: output sites = all reachable sites, rows in ascending linear (b,z,y,x) index;
  * SparseInverseConv3d on the same indice_key: the forward pairs with in/out swapped;
  * features: out[o] += in[i] @ W[kz,ky,kx] for every pair, weight layout [KZ,KY,KX,Cin,Cout].
`dense_conv3d_reference` is an INDEPENDENT oracle: a dense torch conv3d on the densified tensor restricted to
the output sites must give the same numbers.
"""
import numpy as np
import torch
import torch.nn.functional as F


def out_spatial_shape(shape, ksize, stride, padding, dilation):
    return [(shape[j] + 2 * padding[j] - dilation[j] * (ksize[j] - 1) - 1) // stride[j] + 1 for j in range(3)]


def _lin(b, z, y, x, shape):
    return ((b.astype(np.int64) * shape[0] + z) * shape[1] + y) * shape[2] + x


def build_rulebook(indices, batch_size, shape, ksize, stride, padding, dilation, subm):
    """indices int [m,4] (b,z,y,x).  Returns (out_indices int32 [m_out,4], pairs: list over kvol of
    (in_rows, out_rows) int64 arrays sorted by out row, out_shape)."""
    idx = np.asarray(indices).astype(np.int64)
    m = idx.shape[0]
    if subm:
        oshape = list(shape)
    else:
        oshape = out_spatial_shape(shape, ksize, stride, padding, dilation)
    b, z, y, x = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
    cand = []
    for kz in range(ksize[0]):
        for ky in range(ksize[1]):
            for kx in range(ksize[2]):
                nz = z + padding[0] - kz * dilation[0]
                ny = y + padding[1] - ky * dilation[1]
                nx = x + padding[2] - kx * dilation[2]
                ok = (nz >= 0) & (ny >= 0) & (nx >= 0)
                ok &= (nz % stride[0] == 0) & (ny % stride[1] == 0) & (nx % stride[2] == 0)
                oz, oy, ox = nz // stride[0], ny // stride[1], nx // stride[2]
                ok &= (oz < oshape[0]) & (oy < oshape[1]) & (ox < oshape[2])
                rows = np.nonzero(ok)[0]
                cand.append((rows, _lin(b[rows], oz[rows], oy[rows], ox[rows], oshape)))
    if subm:
        in_lin = _lin(b, z, y, x, shape)
        order = np.argsort(in_lin, kind="stable")
        sorted_lin = in_lin[order]
        out_indices = idx.astype(np.int32)

        def out_row(lin):
            pos = np.searchsorted(sorted_lin, lin)
            pos = np.clip(pos, 0, max(m - 1, 0))
            hit = (sorted_lin[pos] == lin) if m else np.zeros(lin.shape, bool)
            return np.where(hit, order[pos], -1)
    else:
        all_out = np.unique(np.concatenate([c[1] for c in cand])) if cand else np.zeros(0, np.int64)
        lin = all_out.copy()
        ox_ = lin % oshape[2]
        lin //= oshape[2]
        oy_ = lin % oshape[1]
        lin //= oshape[1]
        oz_ = lin % oshape[0]
        ob_ = lin // oshape[0]
        out_indices = np.stack([ob_, oz_, oy_, ox_], 1).astype(np.int32)

        def out_row(lin):
            return np.searchsorted(all_out, lin)
    pairs = []
    for rows, olin in cand:
        orow = out_row(olin)
        keep = orow >= 0
        i_r, o_r = rows[keep], orow[keep]
        srt = np.argsort(o_r, kind="stable")
        pairs.append((i_r[srt].astype(np.int64), o_r[srt].astype(np.int64)))
    return out_indices, pairs, oshape


def pairs_to_nbr(pairs, m_out):
    """Output-major neighbour table nbr[o, k] = input row or -1 (the HIP library's rulebook layout)."""
    nbr = np.full((m_out, len(pairs)), -1, dtype=np.int32)
    for k, (i_r, o_r) in enumerate(pairs):
        nbr[o_r, k] = i_r
    return nbr


def pairs_inverse_nbr(pairs, m_in):
    """Table of the inverse conv on the same indice_key: nbr_inv[i, k] = forward OUTPUT row paired with input i."""
    nbr = np.full((m_in, len(pairs)), -1, dtype=np.int32)
    for k, (i_r, o_r) in enumerate(pairs):
        nbr[i_r, k] = o_r
    return nbr


def indice_conv(feat, weight, pairs, m_out, inverse=False):
    """spconv v1 indice_conv: per-offset gather -> mm -> scatter-add.  weight [KZ,KY,KX,Cin,Cout] or [kvol,Cin,Cout]."""
    feat = torch.as_tensor(feat)
    if feat.dtype != torch.float64:  # float64 only when the caller asks for a higher-precision yardstick
        feat = feat.to(torch.float32)
    w = torch.as_tensor(weight).to(feat.dtype)
    w = w.reshape(-1, w.shape[-2], w.shape[-1])
    out = torch.zeros((m_out, w.shape[-1]), dtype=feat.dtype)
    for k, (i_r, o_r) in enumerate(pairs):
        if len(i_r) == 0:
            continue
        src, dst = (o_r, i_r) if inverse else (i_r, o_r)
        out.index_add_(0, torch.as_tensor(dst), feat[torch.as_tensor(src)] @ w[k])
    return out


def dense_conv3d_reference(feat, indices, batch_size, shape, weight, stride, padding, dilation, out_indices):
    """Independent oracle: densify, F.conv3d (cross-correlation, like spconv), sample at out_indices."""
    feat = torch.as_tensor(feat, dtype=torch.float64)
    idx = torch.as_tensor(np.asarray(indices)).long()
    cin = feat.shape[1]
    dense = torch.zeros((batch_size, cin, *shape), dtype=torch.float64)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feat
    w = torch.as_tensor(weight, dtype=torch.float64)  # [KZ,KY,KX,Cin,Cout]
    w = w.permute(4, 3, 0, 1, 2).contiguous()
    out = F.conv3d(dense, w, stride=tuple(stride), padding=tuple(padding), dilation=tuple(dilation))
    oi = torch.as_tensor(np.asarray(out_indices)).long()
    return out[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]].float()
