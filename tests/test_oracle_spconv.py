"""CPU: the spconv-v1 restatement against an independent dense conv3d oracle."""
import numpy as np
import pytest
import torch

from oracle import spconv as osp


def random_sparse(rng, batch, shape, density, cin):
    cells = batch * shape[0] * shape[1] * shape[2]
    m = max(1, int(cells * density))
    lin = rng.choice(cells, size=m, replace=False)
    lin.sort()
    x = lin % shape[2]
    y = (lin // shape[2]) % shape[1]
    z = (lin // (shape[2] * shape[1])) % shape[0]
    b = lin // (shape[2] * shape[1] * shape[0])
    idx = np.stack([b, z, y, x], 1).astype(np.int32)
    feat = rng.standard_normal((m, cin)).astype(np.float32)
    return idx, feat


@pytest.mark.parametrize("shape,stride,padding,subm", [
    ((8, 12, 10), (1, 1, 1), (1, 1, 1), True),
    ((8, 12, 10), (2, 2, 2), (1, 1, 1), False),
    ((5, 12, 10), (2, 2, 2), (0, 1, 1), False),   # the (0,1,1) padding of encoder stage 4 (cfg :66)
    ((3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ((4, 4, 4), (2, 2, 2), (1, 1, 1), False),
])
def test_indice_conv_equals_dense_conv(shape, stride, padding, subm):
    rng = np.random.default_rng(1)
    idx, feat = random_sparse(rng, 2, shape, 0.3, 6)
    w = rng.standard_normal((3, 3, 3, 6, 5)).astype(np.float32)
    out_idx, pairs, oshape = osp.build_rulebook(idx, 2, shape, (3, 3, 3), stride, padding, (1, 1, 1), subm)
    out = osp.indice_conv(feat, w, pairs, out_idx.shape[0])
    dense = osp.dense_conv3d_reference(feat, idx, 2, shape, w, stride, padding, (1, 1, 1), out_idx)
    np.testing.assert_allclose(out.numpy(), dense.numpy(), rtol=1e-4, atol=1e-4)
    if not subm:
        lin = ((out_idx[:, 0].astype(np.int64) * oshape[0] + out_idx[:, 1]) * oshape[1] + out_idx[:, 2]) * oshape[2] + out_idx[:, 3]
        assert (np.diff(lin) > 0).all()  # ascending linear order of the output sites
        # every output site of the dense conv with any contributing input is present
        assert out_idx.shape[0] == len(set(map(tuple, out_idx.tolist())))


def test_hand_checkable_3cube():
    """3^3 grid, two active sites: centre offset and one neighbour pair, checked by hand."""
    idx = np.array([[0, 1, 1, 1], [0, 1, 1, 2]], dtype=np.int32)
    out_idx, pairs, _ = osp.build_rulebook(idx, 1, (3, 3, 3), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True)
    nbr = osp.pairs_to_nbr(pairs, 2)
    assert nbr[0, 13] == 0 and nbr[1, 13] == 1     # centre offset: identity
    assert nbr[0, 14] == 1                           # row 0 sees row 1 at kx = 2 (x + 1)
    assert nbr[1, 12] == 0                           # row 1 sees row 0 at kx = 0 (x - 1)
    assert (nbr >= 0).sum() == 4


def test_inverse_conv_pairs_are_swapped():
    rng = np.random.default_rng(3)
    idx, feat = random_sparse(rng, 1, (6, 8, 8), 0.25, 4)
    out_idx, pairs, _ = osp.build_rulebook(idx, 1, (6, 8, 8), (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), False)
    coarse = rng.standard_normal((out_idx.shape[0], 4)).astype(np.float32)
    w = rng.standard_normal((27, 4, 3)).astype(np.float32)
    up = osp.indice_conv(coarse, w, pairs, idx.shape[0], inverse=True)
    nbr_inv = osp.pairs_inverse_nbr(pairs, idx.shape[0])
    ref = torch.zeros(idx.shape[0], 3)
    for k in range(27):
        has = nbr_inv[:, k] >= 0
        ref[has] += torch.from_numpy(coarse[nbr_inv[has, k]]) @ torch.from_numpy(w[k])
    np.testing.assert_allclose(up.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
