"""K24 — the box tail of a cluster head as C-ABI calls (fsf_decode_cluster_boxes, fsf_class_rank_desc, fsf_nms_select around the capped
multi-class NMS) against the generic path it replaces (FrustumClusterHead._get_bboxes_single's ATen chain, itself checked against the
reference's vectors in test_refine_glue.py and against the oracle chain at full size in test_fullsize_gpu.py) and against numpy."""
import types

import numpy as np
import pytest
import torch

from fullysparsefusion_amd import hip_ops, switches
from fullysparsefusion_amd.mmdet3d_plugin.core.bbox import BasePointBBoxCoder, LiDARInstance3DBoxes, bbox3d2result, xywhr2xyxyr
from fullysparsefusion_amd.mmdet3d_plugin.models.dense_heads import cluster_heads

pytestmark = pytest.mark.gpu


def head_inputs(n, c, code, seed, device, spread=40.0):
    g = torch.Generator().manual_seed(seed)
    cls = torch.randn(n, c, generator=g) * 2.0
    reg = torch.randn(n, code, generator=g) * 0.5
    xyz = (torch.rand(n, 3, generator=g) - 0.5) * spread  # (dense enough for thousands of NMS suppressions)
    return cls.to(device), reg.to(device), xyz.to(device)


@pytest.mark.parametrize("n,c,code", [(1, 1, 8), (777, 3, 10), (10397, 10, 10), (5000, 10, 8)])
def test_decode_cluster_boxes_vs_the_aten_chain_and_float64(device, n, c, code):
    cls, reg, xyz = head_inputs(n, c, code, 1, device)
    wide = torch.cat([reg, cls, xyz, reg], 1)  # column slices of a wider tensor (what the heads hand over): read in place
    cls, reg, xyz = wide[:, code:code + c], wide[:, code + c + 3:], wide[:, code + c:code + c + 3]
    boxes, boxes_nms, scores_t = hip_ops.decode_cluster_boxes(cls, reg, xyz, 1e-6)
    coder = BasePointBBoxCoder(code_size=code)
    want_boxes = coder.decode(reg, xyz)
    want_nms = xywhr2xyxyr(LiDARInstance3DBoxes(want_boxes, box_dim=want_boxes.size(1)).bev)
    want_scores = cls.sigmoid().t().contiguous()
    assert boxes.shape == want_boxes.shape and boxes_nms.shape == (n, 5) and scores_t.shape == (c, n)
    # the same float32 operations in the same order (exp / atan2 / the reciprocal are the device library's in both)
    assert torch.equal(boxes, want_boxes)
    assert torch.equal(boxes_nms, want_nms)
    assert torch.equal(scores_t, want_scores)
    r64, x64 = reg.double().cpu().numpy(), xyz.double().cpu().numpy()
    ref = np.concatenate([r64[:, :3] + x64, np.exp(r64[:, 3:6]) - 1e-6, np.arctan2(r64[:, 6:7], r64[:, 7:8]), r64[:, 8:]], 1)
    assert np.abs(boxes.cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(scores_t.cpu().numpy() - (1 / (1 + np.exp(-cls.double().cpu().numpy()))).T).max() <= 1e-6


@pytest.mark.parametrize("n,c,thr", [(1, 1, 0.1), (5, 2, 0.99), (4097, 10, 0.3), (10397, 10, 0.1), (2048, 7, 0.0)])
def test_class_rank_desc_vs_numpy_stable_sort(device, n, c, thr):
    g = torch.Generator().manual_seed(n + c)
    st = torch.rand(c, n, generator=g)
    st[:, ::7] = st[:, :1].clone()  # ties: broken by ascending index
    if n > 3:
        st[0, :] = 0.0     # a class with nothing above the threshold
    order, rank, count = hip_ops.class_rank_desc(st.to(device), thr)
    s = st.numpy()
    for ci in range(c):
        valid = s[ci] > thr
        masked = np.where(valid, s[ci], -np.inf)
        want = np.argsort(-masked, kind="stable")
        np.testing.assert_array_equal(order[ci].cpu().numpy(), want)
        pos = np.empty(n, np.int64); pos[want] = np.arange(n)
        np.testing.assert_array_equal(rank[ci].cpu().numpy(), np.where(valid, pos, -1))
        assert int(count[ci]) == int(valid.sum())


def fake_head(cfg, classes, code):
    head = types.SimpleNamespace(as_rpn=False, training=False, test_cfg=cfg, tasks=[dict(class_names=classes)], box_code_size=code,
                                 bbox_coder=BasePointBBoxCoder(code_size=code), vis_dir=None, class_names=list(reversed(classes)),
                                 EMPTY_BOX_DIM=9)
    for name in ("_box_type", "_append_debug_columns", "_strip_debug_columns", "_box_tail_fused", "_label_lut"):
        setattr(head, name, types.MethodType(getattr(cluster_heads.FrustumClusterHead, name), head))
    return head


@pytest.mark.parametrize("n,c,code,max_num,spread,must_fuse", [
    (300, 3, 10, 500, 200.0, True), (10397, 10, 10, 500, 60.0, True), (9000, 26, 8, 500, 120.0, True), (6000, 10, 8, 300, 30.0, True), (50, 10, 10, 500, 5.0, True),
    (4000, 2, 10, 500, 25.0, False)])  # (the last: 2 000 boxes of a class piled on 25 m — a class may exhaust its mask window: full repeat)
def test_get_bboxes_single_fused_equals_the_generic_path(device, monkeypatch, n, c, code, max_num, spread, must_fuse):
    """The whole tail, fused (default) against generic (FSF_BOX_TAIL_FUSED=0): the same boxes, scores and labels in the same order —
    class-major when at most max_num boxes survive, the max_num best by score otherwise — and the same host-side result."""
    classes = [f"class{i}" for i in range(c)]
    cfg = dict(use_rotate_nms=True, nms_pre=-1, nms_thr=0.25, score_thr=0.3, min_bbox_size=0, max_num=max_num)
    head = fake_head(cfg, classes, code)
    cls, reg, xyz = head_inputs(n, c, code, 7, device, spread)
    reg[:, 3:6] = reg[:, 3:6].abs() + 0.3  # boxes a few metres across
    out = {}
    for on in (True, False):
        monkeypatch.setattr(switches, "BOX_TAIL_FUSED", on)
        with torch.no_grad():  # (get_bboxes runs under no_grad)
            b, s, l = cluster_heads.FrustumClusterHead._get_bboxes_single(head, 0, cls, None, reg, None, xyz,
                                                                          dict(box_type_3d=LiDARInstance3DBoxes))
        out[on] = (b, s, l, bbox3d2result(b, s, l))
    (bf, sf, lf, rf), (bg, sg, lg, rg) = out[True], out[False]
    fused_ran = getattr(bf, "_host_rows", None) is not None
    assert (fused_ran or not must_fuse) and getattr(bg, "_host_rows", None) is None
    assert len(bf.tensor) == len(bg.tensor) > 0

    def canon(b, s, l):
        """rows (box | score | label) in a canonical order, without the rows tied with the lowest score (a tie across the max_num cut
        may be broken either way: torch's sort is not stable, the selection's is)"""
        rows = np.concatenate([b.tensor.cpu().numpy(), s.cpu().numpy()[:, None], l.cpu().numpy()[:, None].astype(np.float32)], 1)
        rows = rows[rows[:, -2] > rows[:, -2].min()] if len(rows) == max_num else rows
        return rows[np.lexsort(rows.T[::-1])]

    np.testing.assert_array_equal(canon(bf, sf, lf), canon(bg, sg, lg))
    sc = sf.cpu().numpy()
    if len(sc) == max_num:  # the best max_num of more: by descending score
        assert (np.diff(sc) <= 0).all()
        if len(np.unique(sg.cpu().numpy())) == len(sg):  # no tied scores: the order is determined
            assert torch.equal(bf.tensor, bg.tensor) and torch.equal(sf, sg) and torch.equal(lf, lg)
    else:  # class-major, descending within a class: exactly the generic path's rows
        assert torch.equal(bf.tensor, bg.tensor) and torch.equal(sf, sg) and torch.equal(lf, lg)
    assert lf.dtype == torch.int64 and int(lf.max()) < c
    for k, col in (("scores_3d", -2), ("labels_3d", -1)):
        assert not rf[k].is_cuda and np.array_equal(rf[k].numpy().astype(np.float32), np.concatenate(
            [bf.tensor.cpu().numpy(), sf.cpu().numpy()[:, None], lf.cpu().numpy()[:, None].astype(np.float32)], 1)[:, col])
    assert torch.equal(rf["boxes_3d"].tensor, bf.tensor.cpu()) and not rf["boxes_3d"].tensor.is_cuda
    assert rf["labels_3d"].dtype == rg["labels_3d"].dtype and rf["scores_3d"].dtype == rg["scores_3d"].dtype
    if n >= 4000:
        assert len(bf.tensor) == max_num  # the selection's sort ran


def test_nms_select_reports_counts_and_the_incomplete_flag(device):
    n, c, d = 64, 3, 9
    boxes = torch.arange(n * d, dtype=torch.float32, device=device).view(n, d)
    st = torch.rand(c, n, device=device)
    order, rank, count = hip_ops.class_rank_desc(st, 0.5)
    keep = torch.arange(n, device=device).repeat(c, 1)     # every box above the threshold kept, in score order
    num = count.long()
    flag = torch.ones(1, dtype=torch.int32, device=device)
    buf = hip_ops.nms_select(boxes, st, order, keep, num, 64, 500, None, flag).cpu()
    meta = buf[500 * (d + 2):].view(torch.int32)
    total = int(num.sum())
    assert meta.tolist() == [total, total, 1, 0]
    rows = buf[:500 * (d + 2)].view(500, d + 2)[:total]
    s = st.cpu().numpy()
    want = np.concatenate([np.sort(s[ci][s[ci] > 0.5])[::-1] for ci in range(c)])
    np.testing.assert_array_equal(rows[:, d].numpy(), want)                                   # class-major, descending within a class
    np.testing.assert_array_equal(rows[:, d + 1].numpy(), np.repeat(np.arange(c), num.cpu().numpy()).astype(np.float32))
    with pytest.raises(RuntimeError):
        hip_ops.nms_select(boxes, st, order, keep, num, 8192, 500)  # 3 x 8192 boxes do not fit the selection's 16 384
