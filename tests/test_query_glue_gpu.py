"""K29 (csrc/query_glue.hip): every fused glue launch against the ATen chain it replaces — the chain the reference itself writes
(models/backbones/sir.py:65-85, detectors/single_stage_fsd.py:458-474, :951-960, detectors/FSF.py:313-329, :449-504, :657-692,
:961-1010, :1085-1094, roi_heads/bbox_heads/fsd_bbox_head.py:96-112) — bit for bit, and the detector methods that call them against
their own generic branches."""
import numpy as np
import pytest
import torch

from fullysparsefusion_amd import hip_ops
from fullysparsefusion_amd.mmdet3d_plugin.core.bbox import BasePointBBoxCoder

pytestmark = pytest.mark.gpu


def _gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("n,m,cols,lazy,with_index", [(1, 1, 3, False, False), (5000, 37, 5, True, True), (200001, 10397, 5, True, False),
                                                      (70000, 255, 8, False, True)])
def test_sorted_rows_equals_the_aten_chain(device, n, m, cols, lazy, with_index):
    g = _gen(n)
    inv = torch.randint(0, m, (n,), generator=g)
    inv[:m] = torch.arange(m)[:n]
    order = torch.argsort(inv, stable=True).to(torch.int32)
    wide = torch.randn(n, cols + 3, generator=g)
    fcl = torch.randn(n, 3, generator=g)
    centers = torch.randn(m, 3, generator=g)
    index = torch.randint(0, 3 * n, (n,), generator=g) if with_index else None
    inv, order, wide, fcl, centers = (t.to(device) for t in (inv, order, wide, fcl, centers))
    index = index.to(device) if index is not None else None
    points = wide[:, 1:1 + cols]  # a column slice: read through its row stride
    table = torch.zeros((m, 12), dtype=torch.float32, device=device)
    seg, pts_s, fcl_s, idx_s = hip_ops.sorted_rows(order, inv, points, f_cluster=None if lazy else fcl, centers=centers if lazy else None,
                                                   index=index, fill=table)
    o = order.long()
    assert torch.equal(seg, inv.index_select(0, o))
    assert torch.equal(pts_s, points.index_select(0, o))
    want_fcl = (points[:, :3] - centers[inv]) if lazy else fcl
    assert torch.equal(fcl_s, want_fcl.index_select(0, o))
    assert torch.equal(idx_s, index.index_select(0, o) if index is not None else o)
    assert bool((table == float("-inf")).all())


def test_compact_pairs_equals_five_index_selects(device):
    g = _gen(3)
    K, P = 4000, 90000
    means = torch.randn(K, 3, generator=g).to(device)
    centers = torch.randn(P, 3, generator=g).to(device)
    g_ids, p_ids, b_pts = (torch.randint(0, 1 << 40, (P,), generator=g).to(device) for _ in range(3))
    k_idx = torch.nonzero(torch.rand(K, generator=g) > 0.3).squeeze(1).to(device)
    v_idx = torch.nonzero(torch.rand(P, generator=g) > 0.2).squeeze(1).to(device)
    got = hip_ops.compact_pairs(means, k_idx, g_ids, p_ids, b_pts, centers, v_idx)
    want = (means[k_idx], g_ids[v_idx], p_ids[v_idx], b_pts[v_idx], centers[v_idx])
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    empty = hip_ops.compact_pairs(means, k_idx[:0], g_ids, p_ids, b_pts, centers, v_idx[:0])
    assert [t.shape[0] for t in empty] == [0, 0, 0, 0, 0]


@pytest.mark.parametrize("mf,ml", [(244, 10397), (0, 5), (3, 0), (1, 1)])
def test_combine_queries_equals_the_reference_expressions(device, mf, ml):
    g = _gen(mf * 7 + ml)
    fc, lc = torch.randn(mf, 3, generator=g).to(device), torch.randn(ml, 3, generator=g).to(device)
    fco = torch.randint(0, 300, (mf, 3), generator=g).to(device)
    lco = torch.randint(0, 5000, (ml, 3), generator=g).to(device)
    fp = torch.randn(mf, 8, generator=g).to(device)
    centers, coors, preds = hip_ops.combine_queries(fc, lc, fco, lco, fp, 1000)
    re = lco.clone()
    re[:, 0], re[:, 1] = lco[:, 1], lco[:, 0]
    re[:, 2] += 1000
    assert torch.equal(centers, torch.cat([fc, lc], 0))
    assert torch.equal(coors, torch.cat([fco, re], 0))
    assert torch.equal(preds, torch.cat([fp, fp.new_zeros((ml, 8))], 0))


@pytest.mark.parametrize("m,code", [(1, 8), (10641, 10), (333, 8)])
def test_decode_rois_equals_the_coder_and_the_batch_column(device, m, code):
    g = _gen(m)
    reg = (torch.randn(m, code, generator=g) * 0.7).to(device)
    centers = (torch.randn(m, 3, generator=g) * 30).to(device)
    coors = torch.randint(0, 4, (m, 3), generator=g).to(device)
    coder = BasePointBBoxCoder(code_size=code)
    want = torch.cat([coors[:, 0].unsqueeze(-1), coder.decode(reg, centers)], dim=-1)
    got = hip_ops.decode_rois(reg, centers, coors[:, 0], coder.EPS)
    assert got.dtype == want.dtype and torch.equal(got, want)


def test_refine_rows_equal_the_gather_and_the_head_concat(device):
    g = _gen(11)
    n, k, r = 60000, 25000, 900
    points = torch.randn(n, 5, generator=g).to(device)
    info = torch.randn(k, 13, generator=g).to(device)
    rois = torch.randn(r, 10, generator=g).to(device)
    pts_idx = torch.randint(0, n, (k,), generator=g).to(device)
    roi_idx = torch.sort(torch.randint(0, r, (k,), generator=g))[0].to(device)
    pts_out, fcl = hip_ops.refine_rows(info, points, pts_idx, roi_idx, rois[:, 1:4])
    ext = points[pts_idx]
    rel = ext[:, :3] - rois[:, 1:][:, :3][roi_idx]
    want = torch.cat([info[:, 3:6], info[:, 6:-1], info[:, -1][:, None], rel], dim=-1)
    assert torch.equal(pts_out, ext) and torch.equal(fcl, want)


def test_encode_preds_2d_equals_the_detector_methods(device):
    from conftest import build_test_fsf

    model = build_test_fsf()
    g = _gen(5)
    A, m = 60, 300
    anno = torch.rand(1, A, 8, generator=g)
    anno[0, :, :4] *= torch.tensor([1600.0, 900.0, 1600.0, 900.0])
    anno[0, :, 5] = torch.randint(0, 10, (A,), generator=g).float()
    anno[0, :, 6] = torch.randint(0, 6, (A,), generator=g).float()
    coors = torch.zeros((m, 3), dtype=torch.int64)
    coors[:, 2] = torch.randint(0, A + 1, (m,), generator=g)  # 0 = no object
    anno, coors = anno.to(device), coors.to(device)
    want_preds = model.get_single_cls_preds_2d(anno, coors)
    want_enc = model.encode_preds_2d(want_preds, 1600, 900)
    preds, enc = hip_ops.encode_preds_2d(anno[0], coors, model.num_classes, 1600, 900)
    assert torch.equal(preds, want_preds) and torch.equal(enc, want_enc)


def test_weighted_centres_equal_the_reference_expressions(device):
    g = _gen(9)
    n, m = 50000, 240
    points = torch.randn(n, 5, generator=g).to(device) * 20
    w = torch.rand(n, 1, generator=g).to(device)
    w[::7] = 0.0
    w[5] = float("nan")
    got = hip_ops.weighted_xyz(points, w, 1e-5)
    pw = w.clamp(min=1e-5)
    want = torch.cat([points[:, :3] * pw, pw], dim=-1)
    assert torch.equal(got[torch.arange(n, device=device) != 5], want[torch.arange(n, device=device) != 5]) and bool(torch.isnan(got[5]).all())
    mean = torch.rand(m, 4, generator=g).to(device) + 0.1
    assert torch.equal(hip_ops.centroid_divide(mean), mean[:, :3] / mean[:, 3:4])


def test_detector_glue_branches_equal_their_generic_forms(device):
    """FSF.combine_frustum_and_fsd / decode_stage_bboxes / get_cluster_delta_weighted with the K29 launches against the same methods
    on their ATen branches (entered by running them with gradients enabled, which the fused branches refuse)."""
    from conftest import build_test_fsf

    model = build_test_fsf().to(device)
    g = _gen(21)
    mf, ml = 120, 3000
    fc, lc = torch.randn(mf, 3, generator=g).to(device), torch.randn(ml, 3, generator=g).to(device)
    fco = torch.randint(0, 250, (mf, 3), generator=g).to(device)
    lco = torch.randint(0, 4000, (ml, 3), generator=g).to(device)
    fp = torch.randn(mf, 8, generator=g).to(device)
    ff = torch.randn(mf, model.lidar_img_input_dim, generator=g).to(device)
    lf = torch.randn(ml, model.lidar_input_dim, generator=g).to(device)
    res = lambda k: dict(cls_logits=[torch.randn(k, 10, generator=g).to(device)], reg_preds=[torch.randn(k, 10, generator=g).to(device)])  # noqa: E731
    fr, lr = res(mf), res(ml)
    with torch.no_grad():
        a = model.combine_frustum_and_fsd(fc, fco, fr, ff, fp, lc, lco, lr, lf)
    b = model.combine_frustum_and_fsd(fc, fco, fr, ff, fp, lc, lco, lr, lf)
    for i in (0, 1, 4):
        assert torch.equal(a[i], b[i])
    assert torch.equal(a[2]["reg_preds"][0], b[2]["reg_preds"][0])
    with torch.no_grad():
        ra = model.decode_stage_bboxes(a[0], a[1][:, 0], a[2]["reg_preds"])
    rb = model.decode_stage_bboxes(a[0], a[1][:, 0], a[2]["reg_preds"])
    assert torch.equal(ra, rb)
    n = 40000
    points = (torch.randn(n, 5, generator=g) * 10).to(device)
    coors = torch.zeros((n, 3), dtype=torch.int64)
    coors[:, 2] = torch.randint(1, 200, (n,), generator=g)
    coors = coors.to(device)
    w = torch.rand(n, 1, generator=g).to(device)
    with torch.no_grad():
        la, ca, ka = model.get_cluster_delta_weighted(points, coors, w)
    lb, cb, kb = model.get_cluster_delta_weighted(points, coors, w)
    assert torch.equal(ca, cb) and torch.equal(ka, kb) and torch.equal(la.materialize(), lb)


@pytest.mark.parametrize("n,act", [(37, "gelu"), (25000, "gelu"), (4001, "relu")])
def test_sir_input_direct_part_equals_the_materialised_concat(device, n, act):
    """fsf_sir_input_gather's direct_parts_mask (round 6): part 0 through the pooling index, part 1 as it stands — the refine stage's
    `cat([pts_feat[ext_pts_inds], pts_img_feat], -1)` (FSF.py:961-1010) never written — against the same kernel on the concatenation."""
    from fullysparsefusion_amd.mmdet3d_plugin.ops.sst_ops import GatheredRows

    torch.manual_seed(n)
    P = 30000
    wide = torch.randn(P, 132, device=device)
    feats = wide[:, :131]
    img = torch.randn(n, 32, device=device)
    idx = torch.randint(0, P, (n,), device=device)
    points, fcl, extra = torch.randn(n, 5, device=device), torch.randn(n, 13, device=device), torch.randn(n, 13, device=device)
    c = 5 + 131 + 32 + 13
    dims = [13, 16, 32, c]
    layers = [(torch.randn(dims[i + 1], dims[i], device=device) / dims[i] ** 0.5, torch.rand(dims[i + 1], device=device) + 0.5,
               torch.randn(dims[i + 1], device=device) * 0.1) for i in range(3)]
    norm = [20.0, 20.0, 4.0]
    mat = torch.cat([feats[idx], img], 1).contiguous()
    want = hip_ops.sir_input(points, mat, fcl, norm, (*layers, 1e-3), act, 10.0, extra=extra, extra_div=10.0)
    got = hip_ops.sir_input(points, [feats, img], fcl, norm, (*layers, 1e-3), act, 10.0, extra=extra, extra_div=10.0, feats_index=idx,
                            direct_parts=(1,))
    assert torch.equal(got, want)
    assert torch.equal(GatheredRows([feats, img], idx, direct=(1,)).materialize(), mat)


def test_row_planes_fall_back_to_rows_for_a_consumer_that_is_not_a_wide_linear(device):
    """planes handed to a layer K22h does not cover (a head whose first layer has another shape) used to raise; now the rows
    come back from the planes — (hi + lo) * inv_scale, the 22-bit rounding of the source relative to its row maximum."""
    from fullysparsefusion_amd.mmdet3d_plugin.ops import sst_ops

    torch.manual_seed(3)
    x = torch.randn(3000, 256, device=device) * torch.logspace(-6, 6, 3000, device=device)[:, None]
    rp = hip_ops.rows_to_planes(x)
    back = rp.rows()
    assert back.shape == x.shape and float(((back - x).abs() / x.abs().amax(1, keepdim=True)).max()) <= 2.0 ** -21
    lin = torch.nn.Linear(256, 12).to(device)  # 12 outputs: not a wide layer
    with torch.no_grad():
        got = sst_ops.point_linear(lin, rp)
        want = torch.nn.functional.linear(back, lin.weight, lin.bias)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
