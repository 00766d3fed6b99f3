"""GPU parity tests: every C-ABI entry point of libfsf_hip.so against the CPU oracle and the committed golden
vectors.  Integer / index outputs are compared bit-exact; fp32 reductions within 1e-5 (sum order differs),
max bit-exact; sparse-conv features within 1e-4 (BASELINE.json north_star tolerance)."""
import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden
from oracle import project as oproj
from oracle import refine as orefine
from oracle import scatter as oscatter
from oracle import spconv as osp
from oracle import voxelize as ovox

pytestmark = pytest.mark.gpu

PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
VOXEL = (0.2, 0.2, 0.2)


@pytest.fixture(scope="module")
def ops(device):
    from fullysparsefusion_amd import hip_ops

    return hip_ops


def cloud(n, seed=0, oob_frac=0.0):
    rng = np.random.default_rng(seed)
    p = np.empty((n, 5), dtype=np.float32)
    lo, hi = (-50, 50) if oob_frac == 0 else (-53, 53)
    p[:, 0] = rng.uniform(lo, hi, n)
    p[:, 1] = rng.uniform(lo, hi, n)
    p[:, 2] = rng.uniform(-4.99, 2.99, n) if oob_frac == 0 else rng.uniform(-5.5, 3.5, n)
    p[:, 3] = rng.random(n)
    p[:, 4] = rng.integers(0, 10, n) * 0.05
    return p


# ------------------------------------------------------------------------------------------ voxelize
@pytest.mark.parametrize("n,oob", [(10000, 0.0), (307200, 0.0), (5000, 0.3), (1, 0.0)])
def test_voxelize_dynamic_bit_exact(ops, device, n, oob):
    pts = cloud(n, seed=n, oob_frac=oob)
    k = np.arange(513, dtype=np.float32)
    b = (k * np.float32(0.2) + np.float32(-51.2)).astype(np.float32)
    edge = np.zeros((3 * b.size, 5), dtype=np.float32)
    edge[:, 0] = np.concatenate([b, np.nextafter(b, np.float32(100)), np.nextafter(b, np.float32(-100))])
    edge[:, 1] = edge[::-1, 0]
    edge[:, 2] = np.clip(edge[:, 0] / 12.8, -5.0, 3.0)
    pts = np.concatenate([pts, edge], 0)
    grid = ovox.grid_size(VOXEL, PC_RANGE)
    want = ovox.dynamic_voxelize(pts, VOXEL, PC_RANGE)
    zyx, bzyx = ops.voxelize_dynamic(torch.from_numpy(pts).to(device), VOXEL, PC_RANGE, grid, batch_idx=3,
                                     want_zyx=True, want_bzyx=True)
    np.testing.assert_array_equal(zyx.cpu().numpy(), want)
    np.testing.assert_array_equal(bzyx.cpu().numpy()[:, 1:], want.astype(np.int64))
    assert (bzyx[:, 0] == 3).all()


@pytest.mark.parametrize("tag", ["v01", "v03", "v005", "v02"])
def test_voxelize_divfloor_golden(ops, device, tag):
    g = golden_cases(load_golden("divfloor.npz"))[tag]
    pts = torch.from_numpy(g["points"]).to(device)
    c = ops.voxelize_divfloor(pts, g["voxel"].tolist(), g["min"].tolist(), order="xyz")
    np.testing.assert_array_equal(c.cpu().numpy(), g["coors_xyz"])
    bidx = torch.arange(pts.size(0), device=device) % 2
    c4 = ops.voxelize_divfloor(pts, g["voxel"].tolist(), g["min"].tolist(), order="zyx", batch_idx=bidx)
    np.testing.assert_array_equal(c4.cpu().numpy()[:, 1:], g["coors_xyz"][:, ::-1])
    np.testing.assert_array_equal(c4.cpu().numpy()[:, 0], bidx.cpu().numpy())


# -------------------------------------------------------------------------------------------- unique
def check_unique(ops, device, coors_np, bounds=None):
    coors = torch.from_numpy(coors_np)
    want_c, want_inv, want_cnt = torch.unique(coors, return_inverse=True, return_counts=True, dim=0)
    kw = {}
    if bounds is not None:
        kw = dict(col_min=bounds[0], col_max=bounds[1])
    new_coors, plan = ops.unique_rows(coors.to(device), **kw)
    assert plan.m == want_c.size(0)
    np.testing.assert_array_equal(new_coors.cpu().numpy(), want_c.numpy())
    np.testing.assert_array_equal(plan.inv.cpu().numpy(), want_inv.numpy())
    np.testing.assert_array_equal(plan.cnt.cpu().numpy(), want_cnt.numpy())
    order = plan.order.cpu().numpy().astype(np.int64)
    offs = plan.seg_offsets.cpu().numpy()
    n = coors.size(0)
    assert sorted(order.tolist()) == list(range(n))
    np.testing.assert_array_equal(np.diff(offs), want_cnt.numpy())
    assert offs[0] == 0 and offs[-1] == n
    seg_of_sorted = want_inv.numpy()[order]
    assert (np.diff(seg_of_sorted) >= 0).all()
    # stable: ascending point index inside each segment
    same = np.diff(seg_of_sorted) == 0
    assert (np.diff(order)[same] > 0).all()
    return plan


@pytest.mark.parametrize("case", ["k4_avg", "k3_avg", "k1_avg", "single_row", "all_same"])
def test_unique_rows_golden_keys(ops, device, case):
    g = golden_cases(load_golden("scatter_v2.npz"))[case]
    check_unique(ops, device, g["coors"])


def test_unique_rows_voxel_grid_full_size(ops, device):
    pts = cloud(307200, seed=7)
    _, coors = ovox.voxelize_batch([pts[:150000], pts[150000:]], VOXEL, PC_RANGE)
    check_unique(ops, device, coors, bounds=([0, 0, 0, 0], [1, 39, 511, 511]))
    check_unique(ops, device, coors)  # data-dependent bounds path


def test_unique_rows_wide_and_negative_keys(ops, device):
    rng = np.random.default_rng(5)
    c = np.stack([rng.integers(-5, 5, 50000), rng.integers(-100000, 100000, 50000), rng.integers(0, 3, 50000)], 1)
    check_unique(ops, device, c.astype(np.int64))
    c1 = rng.integers(-(2 ** 40), 2 ** 40, (20000, 1)).astype(np.int64)
    check_unique(ops, device, c1)


def test_unique_rows_empty(ops, device):
    new_coors, plan = ops.unique_rows(torch.zeros((0, 4), dtype=torch.int64, device=device))
    assert plan.m == 0 and new_coors.shape == (0, 4)


def test_unique_rows_out_of_bounds_is_reported(ops, device):
    from fullysparsefusion_amd._lib import FsfHipError

    c = torch.tensor([[0, 1, 2, 3], [0, 1, 2, 600]], dtype=torch.int64, device=device)
    with pytest.raises(FsfHipError):
        ops.unique_rows(c, col_min=[0, 0, 0, 0], col_max=[1, 39, 511, 511])


# ------------------------------------------------------------------------------------ segment reduce
@pytest.mark.parametrize("c", [3, 4, 5, 11, 64, 128, 131, 260])
@pytest.mark.parametrize("mode", ["sum", "mean", "max"])
def test_segment_reduce_vs_oracle(ops, device, c, mode):
    rng = np.random.default_rng(c)
    n = 20000
    # skewed segments: singletons, ~3-point voxels and one 6000-row group
    keys = rng.integers(0, 5000, n)
    keys[:6000] = 17
    keys = keys[rng.permutation(n)]
    feat = rng.standard_normal((n, c)).astype(np.float32)
    coors = torch.from_numpy(keys.astype(np.int64))[:, None]
    _, plan = ops.unique_rows(coors.to(device))
    want_c, want_inv = torch.unique(coors, return_inverse=True, dim=0)
    m = want_c.size(0)
    f = torch.from_numpy(feat)
    if mode == "max":
        out, arg = ops.segment_reduce(f.to(device), plan, "max", return_argmax=True)
        want, want_arg = oscatter.segment_max(f, want_inv, m)
        np.testing.assert_array_equal(out.cpu().numpy(), want.numpy())
        np.testing.assert_array_equal(arg.cpu().numpy(), want_arg.numpy())
    else:
        out = ops.segment_reduce(f.to(device), plan, mode)
        want = oscatter.segment_sum(f.double(), want_inv, m)
        if mode == "mean":
            want = want / torch.bincount(want_inv, minlength=m).clamp(min=1)[:, None]
        np.testing.assert_allclose(out.cpu().numpy(), want.float().numpy(), rtol=1e-5, atol=2e-5)
    # run-to-run determinism (no atomics)
    out2 = ops.segment_reduce(f.to(device), plan, mode)
    first = out if mode != "max" else out
    assert torch.equal(first, out2)


@pytest.mark.parametrize("mode", ["sum", "mean", "max"])
def test_segment_reduce_short_vs_oracle_and_chunked(ops, device, mode):
    """fsf_segment_reduce_short (thread per (segment, channel), several tensors in one launch) against the oracle and against
    the chunked kernel: max / argmax bit for bit, sums to the fp32 summation order; empty segments -> 0 / n; a strided view
    and a 600-row segment among the voxels."""
    rng = np.random.default_rng(3)
    n, m = 30000, 9000
    inv = rng.integers(0, m, n)
    inv[inv % 11 == 5] = 1        # unused ids (empty segments)
    inv[:600] = 7                 # one long segment
    inv = inv[rng.permutation(n)]
    plan = ops.segment_plan_from_inverse(torch.from_numpy(inv).to(device), m)
    widths = [131, 33, 11, 5, 64]
    wide = torch.from_numpy(rng.standard_normal((n, 70)).astype(np.float32)).to(device)
    feats = [torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32)).to(device) for c in widths[:-1]] + [wide[:, 4:68]]
    outs = ops.segment_reduce_short(feats, plan, mode)
    tinv = torch.from_numpy(inv)
    cnt = torch.bincount(tinv, minlength=m)
    for f, o in zip(feats, outs):
        chunked = ops.segment_reduce(f, plan, mode)
        if mode == "max":
            want, _ = oscatter.segment_max(f.cpu(), tinv, m)
            np.testing.assert_array_equal(o.cpu().numpy(), want.numpy())
            assert torch.equal(o, chunked)
        else:
            want = oscatter.segment_sum(f.cpu().double(), tinv, m)
            if mode == "mean":
                want = want / cnt.clamp(min=1)[:, None]
            # (segment 1 collects ~2 700 rows: a start-to-end fp32 sum of that length carries ~1e-4 of rounding at |sum| ~ 1e2)
            np.testing.assert_allclose(o.cpu().numpy(), want.float().numpy(), rtol=1e-5, atol=2e-4 if mode == "sum" else 2e-5)
            np.testing.assert_allclose(o.cpu().numpy(), chunked.cpu().numpy(), rtol=1e-5, atol=2e-4 if mode == "sum" else 2e-5)
        assert not o[cnt.to(device) == 0].any()
    # aligned multiples of four take the float4 variant; argmax with a single tensor
    f4 = [feats[4].contiguous(), torch.from_numpy(rng.standard_normal((n, 128)).astype(np.float32)).to(device)]
    o4 = ops.segment_reduce_short(f4, plan, mode)
    for f, o in zip(f4, o4):
        ref = ops.segment_reduce(f, plan, mode)
        if mode == "max":
            assert torch.equal(o, ref)
        else:
            np.testing.assert_allclose(o.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=2e-4 if mode == "sum" else 2e-5)
    if mode == "max":
        (o,), arg = ops.segment_reduce_short([f4[1]], plan, "max", return_argmax=True)
        ref, ref_arg = ops.segment_reduce(f4[1], plan, "max", return_argmax=True)
        assert torch.equal(o, ref) and torch.equal(arg, ref_arg)
    assert all(torch.equal(a, b) for a, b in zip(outs, ops.segment_reduce_short(feats, plan, mode)))  # deterministic


@pytest.mark.parametrize("case", sorted(golden_cases(load_golden("scatter_v2.npz"))))
def test_segment_reduce_reference_golden(ops, device, case):
    g = golden_cases(load_golden("scatter_v2.npz"))[case]
    if int(g["min_points"]) > 0:
        pytest.skip("min_points path is exercised through scatter_v2 (test_plugin_ops)")
    new_coors, plan = ops.unique_rows(torch.from_numpy(g["coors"]).to(device))
    out = ops.segment_reduce(torch.from_numpy(g["feat"]).to(device), plan, str(g["mode"]))
    np.testing.assert_array_equal(new_coors.cpu().numpy(), g["new_coors"])
    np.testing.assert_array_equal(plan.inv.cpu().numpy(), g["inv"])
    if str(g["mode"]) == "max":
        np.testing.assert_array_equal(out.cpu().numpy(), g["new_feat"])
    else:
        np.testing.assert_allclose(out.cpu().numpy(), g["new_feat"], rtol=1e-5, atol=1e-5)


def test_segment_plan_from_inverse_with_empty_segments(ops, device):
    rng = np.random.default_rng(0)
    m, n = 300, 5000
    inv = rng.integers(0, m, n)
    inv[inv % 7 == 3] = 0  # leave some segment ids unused
    plan = ops.segment_plan_from_inverse(torch.from_numpy(inv).to(device), m, return_counts=True)
    cnt = np.bincount(inv, minlength=m)
    np.testing.assert_array_equal(plan.cnt.cpu().numpy(), cnt)
    np.testing.assert_array_equal(np.diff(plan.seg_offsets.cpu().numpy()), cnt)
    f = torch.from_numpy(rng.standard_normal((n, 16)).astype(np.float32))
    out, arg = ops.segment_reduce(f.to(device), plan, "max", return_argmax=True)
    want, want_arg = oscatter.segment_max(f, torch.from_numpy(inv), m)
    np.testing.assert_array_equal(out.cpu().numpy(), want.numpy())
    np.testing.assert_array_equal(arg.cpu().numpy(), want_arg.numpy())
    s = ops.segment_reduce(f.to(device), plan, "mean")
    np.testing.assert_allclose(s.cpu().numpy(), oscatter.segment_mean(f, torch.from_numpy(inv), m).numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mode", ["sum", "mean", "max"])
def test_segment_reduce_backward(ops, device, mode):
    rng = np.random.default_rng(11)
    n, c = 4000, 24
    keys = torch.from_numpy(rng.integers(0, 700, (n, 1)))
    f = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32)).requires_grad_(True)
    _, inv = torch.unique(keys, return_inverse=True, dim=0)
    m = int(inv.max()) + 1
    go = torch.from_numpy(rng.standard_normal((m, c)).astype(np.float32))
    if mode == "max":
        want_out, _ = oscatter.segment_max(f, inv, m)
        ref = f.new_zeros((m, c)).scatter_reduce(0, inv[:, None].expand(n, c), f, reduce="amax", include_self=False)
    elif mode == "mean":
        ref = f.new_zeros((m, c)).scatter_reduce(0, inv[:, None].expand(n, c), f, reduce="mean", include_self=False)
    else:
        ref = f.new_zeros((m, c)).index_add(0, inv, f)
    ref.backward(go)
    _, plan = ops.unique_rows(keys.to(device))
    arg = None
    if mode == "max":
        _, arg = ops.segment_reduce(f.detach().to(device), plan, "max", return_argmax=True)
    g = ops.segment_reduce_backward(go.to(device), plan, mode, argmax=arg)
    np.testing.assert_allclose(g.cpu().numpy(), f.grad.numpy(), rtol=1e-6, atol=1e-6)


def test_gather_rows(ops, device):
    rng = np.random.default_rng(2)
    for c in (3, 64, 128, 131):
        src = torch.from_numpy(rng.standard_normal((999, c)).astype(np.float32))
        idx = torch.from_numpy(rng.integers(0, 999, 12345))
        out = ops.gather_rows(src.to(device), idx.to(device))
        assert torch.equal(out.cpu(), src[idx])


def test_voxel2point_golden(ops, device):
    g = load_golden("neck.npz")
    out, valid = ops.voxel2point(torch.from_numpy(g["points"]).to(device), torch.from_numpy(g["coors"]).to(device),
                                 torch.from_numpy(g["voxel_feats"]).to(device), torch.from_numpy(g["inv"]).to(device),
                                 g["voxel_size"].tolist(), g["pc_range"][:3].tolist(), padding=-1.0)
    np.testing.assert_array_equal(valid.cpu().numpy(), g["mask"])
    np.testing.assert_array_equal(out.cpu().numpy()[g["mask"]], g["out"])  # fp32 bit-exact (same op order)
    assert (~g["mask"]).sum() > 0


# ---------------------------------------------------------------------------------------- projection
@pytest.mark.parametrize("tag", ["nusc_small", "nusc_mid", "av2_small"])
def test_project_gather_golden(ops, device, tag):
    g = golden_cases(load_golden("project.npz"))[tag]
    ids, p2d = ops.project_gather_mask(torch.from_numpy(g["points"]).to(device), torch.from_numpy(g["lidar2img"]).to(device),
                                       torch.from_numpy(g["mask"]).to(device), return_pts_2d=True)
    np.testing.assert_array_equal(p2d.cpu().numpy(), g["pts_2d"])
    np.testing.assert_array_equal(ids.cpu().numpy(), g["obj_id"])
    if "score" in g:
        score, cam_ids = ops.cam_select_score(ids, torch.from_numpy(g["mask_anno"]).to(device), return_ids=True)
        np.testing.assert_array_equal(cam_ids.cpu().numpy(), g["cam_ids"])
        np.testing.assert_array_equal(score.cpu().numpy(), g["score"])
        if g["mask"].shape[1] <= ops.PROJECT_SCORE_MAX_CLS:  # the fused kernel against the reference's own outputs
            fs, fi, ffg = ops.project_score(torch.from_numpy(g["points"]).to(device), torch.from_numpy(g["lidar2img"]).to(device),
                                            torch.from_numpy(g["mask"]).to(device), torch.from_numpy(g["mask_anno"]).to(device),
                                            return_ids=True)
            np.testing.assert_array_equal(fi.cpu().numpy(), g["cam_ids"])
            np.testing.assert_array_equal(fs.cpu().numpy(), g["score"])
            np.testing.assert_array_equal(ffg.cpu().numpy(), g["obj_id"].sum((-2, -1)) > 0)


def test_project_gather_full_size_vs_oracle(ops, device):
    """BASELINE config-3 shape: 3e5 points x 6 cams x 10 classes on a 900x1600 u8 mask."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from fullysparsefusion_amd.synthetic import make_lidar2img, make_mask_data

    rng = np.random.default_rng(0)
    n = 300000
    pts = cloud(n, seed=3)[:, :3].copy()
    L = make_lidar2img(6)
    mask, anno = make_mask_data(rng, 6, 10, 900, 1600, 250)
    want, want_p2d = oproj.points_in_mask(pts, mask, L)
    ids, p2d = ops.project_gather_mask(torch.from_numpy(pts).to(device), torch.from_numpy(L).to(device),
                                       torch.from_numpy(mask).to(device), return_pts_2d=True)
    np.testing.assert_array_equal(p2d.cpu().numpy(), want_p2d)
    np.testing.assert_array_equal(ids.cpu().numpy(), want)
    assert (want > 0).any(-1).any(-1).mean() > 0.05
    wi, ws = oproj.cam_select_score(want, anno)
    score, cam_ids = ops.cam_select_score(ids, torch.from_numpy(anno).to(device), return_ids=True)
    np.testing.assert_array_equal(cam_ids.cpu().numpy(), wi)
    np.testing.assert_array_equal(score.cpu().numpy(), ws)
    # the fused kernel (no [n, 6, 10] int64 tensor in between): same scores, ids and foreground flag
    fscore, fids, ffg = ops.project_score(torch.from_numpy(pts).to(device), torch.from_numpy(L).to(device),
                                          torch.from_numpy(mask).to(device), torch.from_numpy(anno).to(device), return_ids=True)
    np.testing.assert_array_equal(fscore.cpu().numpy(), ws)
    np.testing.assert_array_equal(fids.cpu().numpy(), wi)
    np.testing.assert_array_equal(ffg.cpu().numpy(), want.sum((-2, -1)) > 0)


def test_project_gather_av2_shape_vs_oracle(ops, device):
    """BASELINE config-5 shape: 7 ring cameras, one int32 instance-id plane per camera (ids exceed 255), 1550x2048."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from fullysparsefusion_amd.synthetic import make_lidar2img, make_mask_data

    rng = np.random.default_rng(5)
    n = 150000
    r = rng.uniform(1.0, 200.0, n)
    a = rng.uniform(-np.pi, np.pi, n)
    pts = np.stack([r * np.cos(a), r * np.sin(a), rng.normal(-1.0, 1.0, n)], 1).astype(np.float32)
    L = make_lidar2img(7, fx=1780.0, cx=1024.0, cy=775.0)
    mask, anno = make_mask_data(rng, 7, 1, 1550, 2048, 600, dtype=np.int32)
    assert mask.max() > 255
    want, want_p2d = oproj.points_in_mask(pts, mask, L)
    ids, p2d = ops.project_gather_mask(torch.from_numpy(pts).to(device), torch.from_numpy(L).to(device),
                                       torch.from_numpy(mask).to(device), return_pts_2d=True)
    np.testing.assert_array_equal(p2d.cpu().numpy(), want_p2d)
    np.testing.assert_array_equal(ids.cpu().numpy(), want)
    assert want.max() > 255 and (want > 0).any(-1).any(-1).mean() > 0.05
    wi, ws = oproj.cam_select_score(want, anno)
    fscore, fids, ffg = ops.project_score(torch.from_numpy(pts).to(device), torch.from_numpy(L).to(device),
                                          torch.from_numpy(mask).to(device), torch.from_numpy(anno).to(device), return_ids=True)
    np.testing.assert_array_equal(fscore.cpu().numpy(), ws)
    np.testing.assert_array_equal(fids.cpu().numpy(), wi)
    np.testing.assert_array_equal(ffg.cpu().numpy(), want.sum((-2, -1)) > 0)


@pytest.mark.parametrize("n,ng,empty_group,with_nan", [(250003, 6, 4, False), (3000, 6, None, True), (1, 3, 1, False), (70001, 32, 31, False)])
def test_group_pairs_equal_the_torch_expression(ops, device, n, ng, empty_group, with_nan):
    """K27 against `fg = score > thresh; fg[0] |= ~fg.any(0); fg.t().nonzero()` (single_stage_fsd.py:826-838): the same pairs in the same
    order; a group nobody passes (keeps point 0), NaN scores (never pass), one point, 32 groups, a row-strided score matrix."""
    torch.manual_seed(n + ng)
    buf = torch.rand((n, ng + 3), device=device)
    score = buf[:, :ng]
    thresh = torch.rand(ng, device=device) * 0.5 + 0.45
    if empty_group is not None:
        score[:, empty_group] = 0.0
    if with_nan:
        score[::7, 1] = float("nan")
    fg = score > thresh[None, :]
    fg[0] |= ~fg.any(0)
    want = fg.t().nonzero(as_tuple=False)
    g_ids, p_ids = ops.group_pairs(score, thresh, keep_one=True)
    assert torch.equal(g_ids, want[:, 0]) and torch.equal(p_ids, want[:, 1])
    if empty_group is not None:
        assert int((g_ids == empty_group).sum()) == 1 and int(p_ids[g_ids == empty_group][0]) == 0
    g2, p2 = ops.group_pairs(score, thresh, keep_one=False)
    want2 = (score > thresh[None, :]).t().nonzero(as_tuple=False)
    assert torch.equal(g2, want2[:, 0]) and torch.equal(p2, want2[:, 1])
    if ng <= 16:  # class scores in, one or two member classes per group (gather_group_by_names folded in): the reference's column sums
        nc = 2 * ng - 1
        cls = torch.softmax(torch.randn(n, nc + 1, device=device) * 3.0, 1)[:, :-1]  # (a row-strided view, as in the detector)
        cols = [[2 * g, 2 * g + 1] if g < ng - 1 else [2 * g] for g in range(ng)]
        member = torch.zeros((ng, nc), device=device)
        for g, c in enumerate(cols):
            member[g, c] = 1.0
        grouped = torch.stack([cls[:, c].sum(1) for c in cols], dim=1)
        assert torch.equal(grouped, cls @ member.t())
        th = torch.full((ng,), 0.3, device=device)
        fg3 = grouped > th[None, :]
        fg3[0] |= ~fg3.any(0)
        want3 = fg3.t().nonzero(as_tuple=False)
        g3, p3 = ops.group_pairs(cls, th, keep_one=True, group_cols=cols)
        assert torch.equal(g3, want3[:, 0]) and torch.equal(p3, want3[:, 1])


def overlap_rows_reference(obj):
    """extract_fg_pts + double_overlap_pts + get_sir_coors of the reference (FSF.py:299-308, :260-297, :373-376) on an [n, cells]
    id tensor, as index lists: (src_pt, id) per output row."""
    fg_idx = (obj.sum(-1) > 0).nonzero().squeeze(1)
    rows = obj[fg_idx]
    k = (rows > 0).sum(-1)
    src, ids = [fg_idx], [rows.max(-1)[0]]
    for kk in range(2, int(k.max()) + 1 if k.numel() else 0):
        sel = k == kk
        if int(sel.sum()) == 0:
            continue
        top = rows[sel].topk(kk, dim=-1)[0]
        for j in range(1, kk):
            src.append(fg_idx[sel])
            ids.append(top[:, j])
    return torch.cat(src), torch.cat(ids)


@pytest.mark.parametrize("case", ["frame_like", "dense_overlaps", "no_foreground", "int32_masks"])
def test_overlap_rows_equal_the_reference_loop(ops, device, case):
    """K26 (fsf_overlap_plan + fsf_overlap_rows, fed by fsf_project_score's count / largest id) against the reference's
    extract_fg_pts + double_overlap_pts + get_sir_coors on the [n, cams, classes] tensor of fsf_project_gather_mask: the same rows in
    the same order.  Cases: a frame-like mask set; every pixel of every camera inside several class planes (k of 8 and more, duplicate ids
    in one point's cells); no point inside any mask; int32 id planes."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from fullysparsefusion_amd.synthetic import make_lidar2img, make_mask_data

    rng = np.random.default_rng(11)
    n, ncam, ncls, H, W = 60000, 6, 10, 450, 800
    pts = cloud(n, seed=5)[:, :3].copy()
    L = make_lidar2img(ncam, fx=630.0, cx=400.0, cy=225.0)
    if case == "frame_like":
        mask, _ = make_mask_data(rng, ncam, ncls, H, W, 120)
    elif case == "int32_masks":
        mask, _ = make_mask_data(rng, ncam, ncls, H, W, 400, dtype=np.int32)
    elif case == "no_foreground":
        mask = np.zeros((ncam, ncls, H, W), dtype=np.uint8)
    else:
        mask = np.zeros((ncam, ncls, H, W), dtype=np.uint8)
        for c in range(ncam):
            for k in range(ncls):
                if rng.random() < 0.9:  # a plane of 64 x 64 tiles with ids from a SMALL set: duplicates inside a point's cells
                    tiles = rng.integers(0, 6, size=((H + 63) // 64, (W + 63) // 64)).astype(np.uint8)
                    mask[c, k] = np.kron(tiles, np.ones((64, 64), dtype=np.uint8))[:H, :W]
    anno = np.zeros((int(mask.max()) + 1, 9), dtype=np.float32)
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    obj = ops.project_gather_mask(t(pts), t(L), t(mask)).reshape(n, -1)
    _, fg_bool, (fg, count, max_id) = ops.project_score(t(pts), t(L), t(mask), t(anno), return_overlap=True)
    assert torch.equal(fg_bool, obj.sum(-1) > 0) and torch.equal(count.long(), (obj > 0).sum(-1))
    assert torch.equal(max_id.long(), obj.max(-1)[0])
    num_fg, num_multi, num_extra, ws = ops.overlap_plan(fg, count, ncam * ncls)
    want_src, want_id = overlap_rows_reference(obj)
    k = (obj > 0).sum(-1)
    assert num_fg == int((obj.sum(-1) > 0).sum()) and num_multi == int((k >= 2).sum()) and num_extra == int((k - 1).clamp(min=0).sum())
    if case == "no_foreground":
        assert num_fg == 0
        return
    assert num_multi > 0
    if case == "dense_overlaps":
        assert int(k.max()) >= 8 and bool(((obj.sort(-1)[0][:, 1:] == obj.sort(-1)[0][:, :-1]) & (obj.sort(-1)[0][:, 1:] > 0)).any())
    batch = torch.full((n,), 3, dtype=torch.int64, device=device)
    for b in (None, batch):
        src, coors = ops.overlap_rows(t(pts), t(L), t(mask), max_id, b, ws, num_fg, num_multi, num_extra)
        assert torch.equal(src, want_src)
        assert torch.equal(coors[:, 2], want_id) and bool((coors[:, 1] == 0).all()) and bool((coors[:, 0] == (0 if b is None else 3)).all())


# ----------------------------------------------------------------------------------------- rulebooks
def sparse_sites(rng, batch, shape, m):
    cells = batch * shape[0] * shape[1] * shape[2]
    lin = np.sort(rng.choice(cells, size=m, replace=False))
    x = lin % shape[2]
    y = (lin // shape[2]) % shape[1]
    z = (lin // (shape[2] * shape[1])) % shape[0]
    b = lin // (shape[2] * shape[1] * shape[0])
    return np.stack([b, z, y, x], 1).astype(np.int32)


def surface_sites(rng, batch, shape, m):
    """LiDAR-like occupancy: a thin, x-y dense sheet (ground) plus scattered columns."""
    z0 = shape[0] // 3
    sites = set()
    y0 = max(0, shape[1] - 300)  # long-range grids: keep the sheet dense and put it at the far corner (large coordinates)
    x0 = max(0, shape[2] - 300)
    while len(sites) < m:
        b = int(rng.integers(batch))
        y = int(rng.integers(y0, shape[1]))
        x = int(rng.integers(x0, shape[2]))
        z = z0 + int(rng.integers(0, 2)) if rng.random() < 0.8 else int(rng.integers(shape[0]))
        sites.add((b, z, y, x))
    arr = np.array(sorted(sites), dtype=np.int32)
    return arr


@pytest.mark.parametrize("shape,m", [((8, 12, 10), 400), ((40, 128, 128), 20000), ((3, 3, 3), 2), ((32, 2048, 2048), 60000)])
def test_rulebook_subm_bit_exact(ops, device, shape, m):
    rng = np.random.default_rng(m)
    idx = sparse_sites(rng, 2, shape, m) if m < 20000 else surface_sites(rng, 2, shape, m)
    _, pairs, _ = osp.build_rulebook(idx, 2, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True)
    want = osp.pairs_to_nbr(pairs, idx.shape[0])
    nbr = ops.rulebook_subm(torch.from_numpy(idx).to(device), 2, shape)
    np.testing.assert_array_equal(nbr.cpu().numpy(), want)
    # spconv v1 pair-list form (canonical: ascending output row per offset)
    ip, num = ops.rulebook_to_pairs(nbr)
    ip, num = ip.cpu().numpy(), num.cpu().numpy()
    for k, (i_r, o_r) in enumerate(pairs):
        assert num[k] == len(i_r)
        np.testing.assert_array_equal(ip[k, 0, : num[k]], i_r)
        np.testing.assert_array_equal(ip[k, 1, : num[k]], o_r)


@pytest.mark.parametrize("shape,padding,m", [((8, 12, 10), (1, 1, 1), 400), ((5, 12, 10), (0, 1, 1), 300),
                                             ((40, 128, 128), (1, 1, 1), 20000), ((4, 4, 4), (1, 1, 1), 9),
                                             ((32, 2048, 2048), (1, 1, 1), 60000), ((4, 256, 256), (0, 1, 1), 30000)])
def test_rulebook_strided_bit_exact(ops, device, shape, padding, m):
    rng = np.random.default_rng(m + 1)
    idx = sparse_sites(rng, 2, shape, m) if m < 20000 else surface_sites(rng, 2, shape, m)
    out_idx, pairs, oshape = osp.build_rulebook(idx, 2, shape, (3, 3, 3), (2, 2, 2), padding, (1, 1, 1), False)
    o, nbr, nbr_inv, got_shape = ops.rulebook_strided(torch.from_numpy(idx).to(device), 2, shape, (3, 3, 3), (2, 2, 2), padding)
    assert list(got_shape) == list(oshape)
    np.testing.assert_array_equal(o.cpu().numpy(), out_idx)  # ascending linear (b,z,y,x), bit-exact
    np.testing.assert_array_equal(nbr.cpu().numpy(), osp.pairs_to_nbr(pairs, out_idx.shape[0]))
    np.testing.assert_array_equal(nbr_inv.cpu().numpy(), osp.pairs_inverse_nbr(pairs, idx.shape[0]))


# --------------------------------------------------------------------------------------- sparse conv
@pytest.mark.parametrize("cin,cout", [(16, 16), (64, 64), (64, 128), (128, 128), (256, 128), (128, 256), (32, 20)])
def test_spconv_forward_subm_vs_oracle(ops, device, cin, cout):
    rng = np.random.default_rng(cin * 1000 + cout)
    shape = (16, 48, 48)
    idx = surface_sites(rng, 2, shape, 3000)
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    w = (rng.standard_normal((27, cin, cout)) / np.sqrt(cin * 6)).astype(np.float32)
    _, pairs, _ = osp.build_rulebook(idx, 2, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True)
    want = osp.indice_conv(feat, w, pairs, idx.shape[0])
    nbr = ops.rulebook_subm(torch.from_numpy(idx).to(device), 2, shape)
    wt = ops.spconv_transpose_weight(torch.from_numpy(w).to(device))
    assert torch.equal(wt.cpu(), torch.from_numpy(w).permute(0, 2, 1).contiguous())
    out = ops.spconv_forward(torch.from_numpy(feat).to(device), wt, nbr)
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)
    # fused epilogue: BN affine + residual + ReLU
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    res = torch.from_numpy(rng.standard_normal((idx.shape[0], cout)).astype(np.float32))
    out2 = ops.spconv_forward(torch.from_numpy(feat).to(device), wt, nbr, scale=scale.to(device), shift=shift.to(device),
                              residual=res.to(device), relu=True)
    want2 = torch.relu(want * scale + shift + res)
    np.testing.assert_allclose(out2.cpu().numpy(), want2.numpy(), rtol=1e-4, atol=1e-4)
    assert torch.equal(out2, ops.spconv_forward(torch.from_numpy(feat).to(device), wt, nbr, scale=scale.to(device),
                                                shift=shift.to(device), residual=res.to(device), relu=True))


@pytest.mark.parametrize("m,cin,cout", [(3000, 16, 16), (3000, 64, 64), (3000, 64, 128), (40000, 128, 128), (20000, 256, 128),
                                        (3000, 128, 256), (1500, 48, 32),
                                        (1517, 512, 512)])  # the deepest U-Net level: 12 row blocks x 4 slices x 16 k ranges, XCD-aware grid
def test_spconv_forward_split_vs_oracle_and_fp32_kernel(ops, device, m, cin, cout):
    """K9b (row-stationary, exact bf16 split on the bf16 matrix cores) against the CPU oracle at small sizes and against the
    fp32-pipe kernel at large ones, with the fused epilogue; error vs float64 on sampled rows no larger than fp32's."""
    rng = np.random.default_rng(m + cin + cout)
    shape = (16, 48, 48) if m <= 3000 else (40, 512, 512)
    idx = surface_sites(rng, 2 if m <= 3000 else 1, shape, m)
    n = idx.shape[0]
    feat = (rng.standard_normal((n, cin)) * np.exp(rng.standard_normal((n, 1)))).astype(np.float32)
    w = (rng.standard_normal((27, cin, cout)) / np.sqrt(cin * 6)).astype(np.float32)
    nbr = ops.rulebook_subm(torch.from_numpy(idx).to(device), 2 if m <= 3000 else 1, shape)
    f, wd = torch.from_numpy(feat).to(device), torch.from_numpy(w).to(device)
    planes = ops.spconv_prepare_weight_split(wd)
    out = ops.spconv_forward_split(f, planes, 27, cout, nbr)
    ref = ops.spconv_forward(f, ops.spconv_transpose_weight(wd), nbr) if cin % 16 == 0 else None
    if m <= 3000:
        _, pairs, _ = osp.build_rulebook(idx, 2, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True)
        want = osp.indice_conv(feat, w, pairs, n)
        np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)
    rows = torch.from_numpy(rng.choice(n, size=min(n, 512), replace=False)).to(device)
    nb = nbr.index_select(0, rows).long()
    gathered = torch.where((nb >= 0)[:, :, None], f.double()[nb.clamp(min=0)], torch.zeros((), dtype=torch.float64, device=device))
    want64 = torch.einsum("rkc,kcd->rd", gathered, wd.double())
    err = float((out.index_select(0, rows).double() - want64).abs().max())
    scale_ = max(1.0, float(want64.abs().max()))
    assert err <= 2e-5 * scale_
    if ref is not None:
        err32 = float((ref.index_select(0, rows).double() - want64).abs().max())
        assert err <= max(2.0 * err32, 2e-6 * scale_), (err, err32)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(device)
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(device)
    res = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(device)
    out2 = ops.spconv_forward_split(f, planes, 27, cout, nbr, scale=sc, shift=sh, residual=res, relu=True)
    want2 = torch.relu(out * sc + sh + res)
    assert float((out2 - want2).abs().max()) <= 1e-5 * max(1.0, float(want2.abs().max()))
    assert torch.equal(out2, ops.spconv_forward_split(f, planes, 27, cout, nbr, scale=sc, shift=sh, residual=res, relu=True))


def decode_planes(pl):
    """Planes -> f32 [m + 1, c] = (hi + lo) * inverse row scale, on the host (float64 arithmetic)."""
    m, c = pl.m, pl.c
    raw = pl.data.cpu().numpy()[: (m + 1) * c * 4].view(np.float16).reshape(m + 1, c // 8, 2, 8).astype(np.float64)
    inv = pl.scales.cpu().numpy().astype(np.float64)  # [m + 1, nchunk]
    val = (raw[:, :, 0, :] + raw[:, :, 1, :]).reshape(m + 1, c)
    chunk = np.minimum(np.arange(c) // 128, inv.shape[1] - 1)
    return val * inv[:, chunk]


@pytest.mark.parametrize("m,c", [(1, 8), (1000, 64), (4097, 128), (300, 256), (257, 72)])
def test_to_planes_is_a_22_bit_row_scaled_split(ops, device, m, c):
    """fsf_to_planes: x = (hi + lo) / s_row with s_row a power of two per (row, 128-channel chunk); hi + lo reproduces x to
    2^-21 of the chunk's largest magnitude (22 bits of every element that matters at the row's scale), the planes stay
    inside the f16 range whatever the row's magnitude (1e-30 ... 1e30), row m is zeros with scale 1, strided input rows."""
    rng = np.random.default_rng(m + c)
    x = rng.standard_normal((m, c)) * np.exp(rng.standard_normal((m, 1)) * 3.0)
    x[0] *= 1e30 if m > 1 else 1.0
    if m > 2:
        x[1] *= 1e-30
        x[2] = 0.0
    x = x.astype(np.float32)
    buf = torch.full((m, c + 12), float("nan"), device=device)
    view = buf[:, 4:4 + c] if c % 4 == 0 else buf[:, :c]
    view.copy_(torch.from_numpy(x).to(device))
    for src in (torch.from_numpy(x).to(device), view):
        pl = ops.to_planes(src)
        got = decode_planes(pl)
        assert np.isfinite(pl.data.cpu().numpy()[: (m + 1) * c * 4].view(np.float16).astype(np.float32)).all()
        assert not got[m].any() and (pl.scales[m].cpu().numpy() == 1.0).all()
        for ch in range((c + 127) // 128):
            sl = slice(128 * ch, min(c, 128 * ch + 128))
            amax = np.abs(x[:, sl]).max(1, keepdims=True).astype(np.float64)
            err = np.abs(got[:m, sl] - x[:, sl].astype(np.float64))
            assert (err <= amax * 2.0 ** -21 + 1e-300).all()
            inv = pl.scales[:m, ch].cpu().numpy().astype(np.float64)
            assert (np.log2(inv) == np.round(np.log2(inv))).all()  # powers of two: the scaling is exact
            top = amax[:, 0] / inv
            assert ((top >= 2.0 ** 13) & (top < 2.0 ** 14) | (amax[:, 0] == 0)).all()


@pytest.mark.parametrize("m,cins,cout", [(70000, (128,), 128), (70000, (128, 128), 128), (90000, (64,), 64), (66000, (64,), 128),
                                         (20000, (128,), 128), (5000, (64, 128), 128), (3000, (128,), 256), (300, (32,), 64), (17, (128,), 128)])
def test_spconv_forward_planes_vs_float64(ops, device, m, cins, cout):
    """K9c (pre-split f16 planes, 3 MFMAs per fp32-equivalent product, cell skipping) against float64 on sampled rows — error
    of the size of the fp32-pipe kernel's own —, against the K9b kernel, with one and two sources (the decoder's channel
    concatenation), the fused epilogue, the plane-form output (what the next layer consumes) and bitwise determinism.  Rows
    of very different magnitude share an MFMA (per-row scales)."""
    rng = np.random.default_rng(m + sum(cins) + cout)
    small = m <= 5000
    shape = (16, 48, 48) if small else (40, 512, 512)
    bs = 2 if small else 1
    idx = surface_sites(rng, bs, shape, m)
    n = idx.shape[0]
    cin = sum(cins)
    feat = (rng.standard_normal((n, cin)) * np.exp(rng.standard_normal((n, 1)) * 1.5)).astype(np.float32)
    w = (rng.standard_normal((27, cin, cout)) / np.sqrt(cin * 6)).astype(np.float32)
    nbr = ops.rulebook_subm(torch.from_numpy(idx).to(device), bs, shape)
    f, wd = torch.from_numpy(feat).to(device), torch.from_numpy(w).to(device)
    assert ops.spconv_planes_supported(cins, cout, 27)
    wpl = ops.spconv_prepare_weight_planes(wd)
    srcs, c0 = [], 0
    for c in cins:
        srcs.append(ops.to_planes(f[:, c0:c0 + c]))  # a column slice: row-strided input
        c0 += c
    out, opl = ops.spconv_forward_planes(srcs, wpl, 27, cout, nbr, want_planes=True)
    rows = torch.from_numpy(rng.choice(n, size=min(n, 512), replace=False)).to(device)
    nb = nbr.index_select(0, rows).long()
    gathered = torch.where((nb >= 0)[:, :, None], f.double()[nb.clamp(min=0)], torch.zeros((), dtype=torch.float64, device=device))
    want64 = torch.einsum("rkc,kcd->rd", gathered, wd.double())
    scale_ = max(1.0, float(want64.abs().max()))
    err = float((out.index_select(0, rows).double() - want64).abs().max())
    assert err <= 1e-5 * scale_, (err, scale_)
    if cin % 16 == 0:
        ref = ops.spconv_forward(f, ops.spconv_transpose_weight(wd), nbr)
        err32 = float((ref.index_select(0, rows).double() - want64).abs().max())
        assert err <= max(4.0 * err32, 4e-6 * scale_), (err, err32)
        assert float((out - ref).abs().max()) <= 2e-5 * scale_  # every row, against the fp32-pipe kernel
    # the plane-form output decodes to the fp32 output (22-bit split of it)
    dec = decode_planes(opl)
    o64 = out.cpu().numpy().astype(np.float64)
    for ch in range((cout + 127) // 128):
        sl = slice(128 * ch, min(cout, 128 * ch + 128))
        amax = np.abs(o64[:, sl]).max(1, keepdims=True)
        assert (np.abs(dec[:n, sl] - o64[:, sl]) <= amax * 2.0 ** -21 + 1e-300).all()
    assert not dec[n].any()
    # fused epilogue + residual + ReLU, planes only / fp32 only outputs, determinism
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(device)
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(device)
    res = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(device)
    out2, opl2 = ops.spconv_forward_planes(srcs, wpl, 27, cout, nbr, scale=sc, shift=sh, residual=res, relu=True, want_planes=True)
    want2 = torch.relu(out * sc + sh + res)
    assert float((out2 - want2).abs().max()) <= 1e-5 * max(1.0, float(want2.abs().max()))
    out3, none = ops.spconv_forward_planes(srcs, wpl, 27, cout, nbr, scale=sc, shift=sh, residual=res, relu=True)
    assert none is None and torch.equal(out2, out3)
    none, opl3 = ops.spconv_forward_planes(srcs, wpl, 27, cout, nbr, scale=sc, shift=sh, residual=res, relu=True, want_out=False,
                                           want_planes=True)
    assert none is None and torch.equal(opl3.data, opl2.data) and torch.equal(opl3.scales, opl2.scales)
    # a chain: the plane-form output feeds the next layer exactly like a fresh conversion of the fp32 output
    if cout in (64, 128):
        w2 = torch.from_numpy((rng.standard_normal((27, cout, 64)) / np.sqrt(cout * 6)).astype(np.float32)).to(device)
        wpl2 = ops.spconv_prepare_weight_planes(w2)
        a_, _ = ops.spconv_forward_planes([opl2], wpl2, 27, 64, nbr)
        b_, _ = ops.spconv_forward_planes([ops.to_planes(out2)], wpl2, 27, 64, nbr)
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("m", [3000, 60000])
def test_spconv_forward_planes_strided_and_inverse_tables(ops, device, m):
    """K9c only sees a neighbour table: the stride-2 convolution (m_in != m_out, 3-9 pairs per output row) and its inverse
    (the transposed table) against the fp32-pipe kernel and float64 on sampled rows."""
    rng = np.random.default_rng(m)
    small = m <= 5000
    shape = (16, 48, 48) if small else (40, 512, 512)
    idx = surface_sites(rng, 1, shape, m)
    n = idx.shape[0]
    out_idx, nbr, nbr_inv, oshape = ops.rulebook_strided(torch.from_numpy(idx).to(device), 1, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    mo = out_idx.size(0)
    for table, rows_in, rows_out in ((nbr, n, mo), (nbr_inv, mo, n)):
        feat = (rng.standard_normal((rows_in, 128)) * np.exp(rng.standard_normal((rows_in, 1)))).astype(np.float32)
        w = (rng.standard_normal((27, 128, 128)) / np.sqrt(128 * 4)).astype(np.float32)
        f, wd = torch.from_numpy(feat).to(device), torch.from_numpy(w).to(device)
        out, _ = ops.spconv_forward_planes([ops.to_planes(f)], ops.spconv_prepare_weight_planes(wd), 27, 128, table)
        ref = ops.spconv_forward(f, ops.spconv_transpose_weight(wd), table)
        assert out.shape == (rows_out, 128)
        scale_ = max(1.0, float(ref.abs().max()))
        assert float((out - ref).abs().max()) <= 2e-5 * scale_
        rows = torch.from_numpy(rng.choice(rows_out, size=min(rows_out, 256), replace=False)).to(device)
        nb = table.index_select(0, rows).long()
        g = torch.where((nb >= 0)[:, :, None], f.double()[nb.clamp(min=0)], torch.zeros((), dtype=torch.float64, device=device))
        want = torch.einsum("rkc,kcd->rd", g, wd.double())
        assert float((out.index_select(0, rows).double() - want).abs().max()) <= 1e-5 * scale_


@pytest.mark.parametrize("m,cin,cout", [(100000, 128, 128), (36000, 256, 128), (7500, 256, 256), (1500, 512, 512)])
def test_spconv_forward_full_size_properties(ops, device, m, cin, cout):
    """BASELINE-size layers (the persistent work-queue kernel with 1..9 offset splits, stealing across XCD queues, in-kernel
    fold by the last arriver): size-independent properties instead of the CPU oracle —
    bitwise determinism call after call, linearity in the features, the identity kernel, and a dense-torch check of a
    random sample of output rows."""
    rng = np.random.default_rng(m + cin)
    shape = (40, 512, 512)
    idx = surface_sites(rng, 1, shape, m)
    n = idx.shape[0]
    nbr = ops.rulebook_subm(torch.from_numpy(idx).to(device), 1, shape)
    g = torch.Generator(device="cpu").manual_seed(m)
    feat = torch.randn(n, cin, generator=g).to(device)
    feat2 = torch.randn(n, cin, generator=g).to(device)
    w = (torch.randn(27, cin, cout, generator=g) / (cin * 6) ** 0.5).to(device)
    wt = ops.spconv_transpose_weight(w)
    out = ops.spconv_forward(feat, wt, nbr)
    for _ in range(3):  # the queue order differs from call to call; the result may not
        assert torch.equal(out, ops.spconv_forward(feat, wt, nbr))
    # linearity: conv(a x + y) = a conv(x) + conv(y) up to fp32 rounding
    lhs = ops.spconv_forward(feat * 0.5 + feat2, wt, nbr)
    rhs = out * 0.5 + ops.spconv_forward(feat2, wt, nbr)
    assert float((lhs - rhs).abs().max()) <= 2e-4 * max(1.0, float(rhs.abs().max()))
    # a random sample of output rows against the gather -> matmul definition in float64
    rows = torch.from_numpy(rng.choice(n, size=min(n, 512), replace=False)).to(device)
    nb = nbr.index_select(0, rows).long()                                  # [r, 27]
    gathered = torch.where((nb >= 0)[:, :, None], feat.double()[nb.clamp(min=0)], torch.zeros((), dtype=torch.float64, device=device))
    want = torch.einsum("rkc,kcd->rd", gathered, w.double())
    got = out.index_select(0, rows).double()
    assert float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
    # identity kernel (centre offset = I, cin == cout): output == input, exactly
    if cin == cout:
        wi = torch.zeros(27, cin, cout, device=device)
        wi[13] = torch.eye(cin, device=device)
        assert torch.equal(ops.spconv_forward(feat, ops.spconv_transpose_weight(wi), nbr), feat)


def test_spconv_forward_strided_and_inverse_vs_dense(ops, device):
    rng = np.random.default_rng(9)
    shape = (9, 24, 24)
    idx = surface_sites(rng, 1, shape, 900)
    cin, cout = 32, 48
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / 10).astype(np.float32)
    for padding in [(1, 1, 1), (0, 1, 1)]:
        o, nbr, nbr_inv, oshape = ops.rulebook_strided(torch.from_numpy(idx).to(device), 1, shape, (3, 3, 3), (2, 2, 2), padding)
        wt = ops.spconv_transpose_weight(torch.from_numpy(w.reshape(27, cin, cout)).to(device))
        out = ops.spconv_forward(torch.from_numpy(feat).to(device), wt, nbr)
        dense = osp.dense_conv3d_reference(feat, idx, 1, shape, w, (2, 2, 2), padding, (1, 1, 1), o.cpu().numpy())
        np.testing.assert_allclose(out.cpu().numpy(), dense.numpy(), rtol=1e-4, atol=1e-4)
        # inverse conv back to the fine sites with the swapped pairs
        w2 = (rng.standard_normal((27, cout, 16)) / 10).astype(np.float32)
        out_idx, pairs, _ = osp.build_rulebook(idx, 1, shape, (3, 3, 3), (2, 2, 2), padding, (1, 1, 1), False)
        want_up = osp.indice_conv(out.cpu().numpy(), w2, pairs, idx.shape[0], inverse=True)
        up = ops.spconv_forward(out, ops.spconv_transpose_weight(torch.from_numpy(w2).to(device)), nbr_inv)
        np.testing.assert_allclose(up.cpu().numpy(), want_up.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 128), (64, 128), (128, 64), (256, 128), (128, 320), (48, 32), (16, 16)])
@pytest.mark.parametrize("kind", ["subm", "strided", "inverse"])
def test_spconv_backward_vs_oracle_autograd(ops, device, kind, cin, cout):
    """K10: d/d feat = conv over the transposed rulebook, d/d W = pair-list outer products; oracle = autograd through
    the per-offset gather / mm / index_add restatement.  fp32, rtol 1e-4 of the gradient scale."""
    rng = np.random.default_rng(cin * 7 + cout + len(kind))
    shape = (12, 40, 40)
    idx = surface_sites(rng, 2, shape, 2500 if kind == "subm" else 1800)
    subm = kind == "subm"
    stride = (1, 1, 1) if subm else (2, 2, 2)
    out_idx, pairs, _ = osp.build_rulebook(idx, 2, shape, (3, 3, 3), stride, (1, 1, 1), (1, 1, 1), subm)
    dev_idx = torch.from_numpy(idx).to(device)
    if subm:
        nbr = ops.rulebook_subm(dev_idx, 2, shape)
        table, table_t, flip = nbr, nbr, True
        m_in, m_out = idx.shape[0], idx.shape[0]
    else:
        _, nbr, nbr_inv, _ = ops.rulebook_strided(dev_idx, 2, shape, (3, 3, 3), stride, (1, 1, 1))
        if kind == "strided":
            table, table_t, flip = nbr, nbr_inv, False
            m_in, m_out = idx.shape[0], out_idx.shape[0]
        else:
            table, table_t, flip = nbr_inv, nbr, False
            m_in, m_out = out_idx.shape[0], idx.shape[0]
    feat = torch.from_numpy(rng.standard_normal((m_in, cin)).astype(np.float32)).requires_grad_()
    w = torch.from_numpy((rng.standard_normal((27, cin, cout)) / np.sqrt(cin * 6)).astype(np.float32)).requires_grad_()
    gout = torch.from_numpy(rng.standard_normal((m_out, cout)).astype(np.float32))
    want = osp.indice_conv(feat, w, pairs, m_out, inverse=kind == "inverse")
    want.backward(gout)
    # data gradient: the forward kernel over the transposed table with the un-transposed (k-flipped for SubM) weight
    wd = w.detach().to(device)
    g_feat = ops.spconv_forward(gout.to(device), wd.flip(0).contiguous() if flip else wd, table_t)
    np.testing.assert_allclose(g_feat.cpu().numpy(), feat.grad.numpy(), rtol=1e-4, atol=1e-4)
    # weight gradient
    ip, num = ops.rulebook_to_pairs(table)
    g_w = ops.spconv_backward_weight(feat.detach().to(device), gout.to(device), ip, num)
    tol = 1e-4 * float(w.grad.abs().max())
    np.testing.assert_allclose(g_w.cpu().numpy(), w.grad.numpy(), rtol=1e-4, atol=tol)
    assert torch.equal(g_w, ops.spconv_backward_weight(feat.detach().to(device), gout.to(device), ip, num))  # deterministic


@pytest.mark.parametrize("cin,cmid,cout", [(128, 128, 128), (64, 128, 64), (256, 128, 128)])
def test_conv_modules_training_on_the_plane_kernel_vs_oracle_autograd(ops, device, cin, cmid, cout, monkeypatch):
    """Training mode of the conv MODULES at >= 4096 rows (ops/spconv.py `_SparseConvFn`, planes=True): the forward and the data
    gradient of a SubM -> strided -> inverse chain run on K9c (features / grad_out through fsf_to_planes, the step's weights
    through fsf_spconv_prepare_weight_planes), the weight gradients on K10; oracle = autograd through the per-offset
    gather / mm / index_add restatement.  Also equal within tolerance to the FSF_TRAIN_PLANES=0 path."""
    from fullysparsefusion_amd import hip_ops
    from fullysparsefusion_amd.mmdet3d_plugin.ops import spconv as sp

    rng = np.random.default_rng(cin + 3 * cmid + 7 * cout)
    shape = (12, 96, 96)
    idx = surface_sites(rng, 2, shape, 16000)
    m = idx.shape[0]
    torch.manual_seed(cin)
    mods = [sp.SubMConv3d(cin, cmid, 3, padding=1, bias=False, indice_key="subm"),
            sp.SparseConv3d(cmid, cmid, 3, stride=2, padding=1, bias=False, indice_key="down"),
            sp.SparseInverseConv3d(cmid, cout, 3, indice_key="down", bias=False)]
    for mod in mods:
        mod.to(device).train()
    feat = torch.from_numpy(rng.standard_normal((m, cin)).astype(np.float32))
    probe = torch.from_numpy(rng.standard_normal((m, cout)).astype(np.float32))
    calls = []
    real = hip_ops.spconv_forward_planes
    monkeypatch.setattr(hip_ops, "spconv_forward_planes", lambda *a, **k: (calls.append(a[4].size(0)), real(*a, **k))[1])

    def run():
        f = feat.to(device).requires_grad_()
        x = sp.SparseConvTensor(f, torch.from_numpy(idx).to(device), list(shape), 2)
        for mod in mods:
            mod.zero_grad()
            x = mod(x)
        (x.features * probe.to(device)).sum().backward()
        return [x.features.detach().cpu(), f.grad.cpu()] + [mod.weight.grad.cpu().clone() for mod in mods]

    got = run()
    m_down = osp.build_rulebook(idx, 2, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), False)[0].shape[0]
    assert m_down >= 4096, m_down
    assert len(calls) == 6 and sorted(calls) == sorted([m, m_down, m, m, m_down, m]), calls  # 3 forwards + 3 data gradients
    from fullysparsefusion_amd import switches

    monkeypatch.setattr(switches, "TRAIN_PLANES", False)
    other = run()
    assert len(calls) == 6
    # oracle
    _, pairs_subm, _ = osp.build_rulebook(idx, 2, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True)
    _, pairs_down, _ = osp.build_rulebook(idx, 2, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), False)
    f = feat.clone().requires_grad_()
    ws = [mod.weight.detach().cpu().reshape(27, mod.in_channels, mod.out_channels).clone().requires_grad_() for mod in mods]
    y = osp.indice_conv(f, ws[0], pairs_subm, m)
    y = osp.indice_conv(y, ws[1], pairs_down, m_down)
    y = osp.indice_conv(y, ws[2], pairs_down, m, inverse=True)
    (y * probe).sum().backward()
    want = [y.detach(), f.grad] + [w.grad for w in ws]
    for name, a, b, c in zip(["out", "grad_in", "gw_subm", "gw_down", "gw_inv"], got, other, want):
        c = c.reshape(a.shape)
        scale = float(c.abs().max())
        assert float((a - c).abs().max()) <= 1e-4 * scale, (name, float((a - c).abs().max()), scale)
        assert float((a - b).abs().max()) <= 1e-4 * scale, (name, "vs FSF_TRAIN_PLANES=0")


def test_spconv_backward_weight_large_and_empty_offsets(ops, device):
    """Many pair-range splits (partials folded in split order), offsets with zero pairs (isolated sites) and a pair
    count that is not a multiple of the 32-pair stage."""
    rng = np.random.default_rng(77)
    shape = (24, 300, 300)
    idx = surface_sites(rng, 1, shape, 70001)
    cin = cout = 64
    nbr = ops.rulebook_subm(torch.from_numpy(idx).to(device), 1, shape)
    ip, num = ops.rulebook_to_pairs(nbr)
    feat = torch.from_numpy(rng.standard_normal((idx.shape[0], cin)).astype(np.float32)).to(device)
    gout = torch.from_numpy(rng.standard_normal((idx.shape[0], cout)).astype(np.float32)).to(device)
    g_w = ops.spconv_backward_weight(feat, gout, ip, num)
    nb = nbr.long()
    for k in [0, 4, 13, 22, 26]:
        sel = (nb[:, k] >= 0).nonzero().squeeze(1)
        want = feat[nb[sel, k]].double().t() @ gout[sel].double()
        err = (g_w[k].double() - want).abs().max().item()
        assert err <= 1e-4 * max(1.0, want.abs().max().item()), (k, err)
    # isolated sites: only the centre offset has pairs, every other slice must come back exactly zero
    lone = np.stack([np.zeros(50), np.full(50, 3), np.arange(50) * 5, np.arange(50) * 5], 1).astype(np.int32)
    nbr1 = ops.rulebook_subm(torch.from_numpy(lone).to(device), 1, shape)
    ip1, num1 = ops.rulebook_to_pairs(nbr1)
    assert num1.cpu().tolist() == [0] * 13 + [50] + [0] * 13
    g1 = ops.spconv_backward_weight(feat[:50], gout[:50], ip1, num1)
    assert torch.count_nonzero(g1[:13]) == 0 and torch.count_nonzero(g1[14:]) == 0
    np.testing.assert_allclose(g1[13].cpu().numpy(), (feat[:50].t() @ gout[:50]).cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_spconv_asymmetric_weight_detects_transposes(ops, device):
    """A = I-like check with an asymmetric B (guide §3): one active site, identity features."""
    idx = np.array([[0, 1, 1, 1]], dtype=np.int32)
    cin = cout = 16
    feat = np.eye(1, cin, 3, dtype=np.float32)  # e_3
    w = np.zeros((27, cin, cout), dtype=np.float32)
    w[13] = np.arange(cin * cout, dtype=np.float32).reshape(cin, cout)
    nbr = ops.rulebook_subm(torch.from_numpy(idx).to(device), 1, (3, 3, 3))
    out = ops.spconv_forward(torch.from_numpy(feat).to(device), ops.spconv_transpose_weight(torch.from_numpy(w).to(device)), nbr)
    np.testing.assert_array_equal(out.cpu().numpy()[0], w[13][3])


# ------------------------------------------------------------------------------------- in-group rank
def test_ingroup_rank(ops, device):
    rng = np.random.default_rng(4)
    g = torch.from_numpy(rng.integers(0, 300, 50000))
    r = ops.ingroup_rank(g.to(device)).cpu()
    assert torch.equal(r, oscatter.ingroup_rank(g))


# ------------------------------------------------------------------------------------ norm + activation
@pytest.mark.parametrize("c", [3, 16, 32, 64, 128, 133, 180, 256, 512, 640, 1000, 1024])
@pytest.mark.parametrize("act", ["gelu", "relu", None])
def test_norm_act_vs_torch(ops, device, c, act):
    torch.manual_seed(c)
    n = 5000
    x = torch.randn(n, c) * 3 + 0.5
    ln = torch.nn.LayerNorm(c, eps=1e-3)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.normal_()
    f = {"gelu": torch.nn.functional.gelu, "relu": torch.relu, None: lambda t: t}[act]
    w64, b64 = ln.weight.detach().double(), ln.bias.detach().double()
    want = f(torch.nn.functional.layer_norm(x.double(), (c,), w64, b64, 1e-3)).float()
    got = ops.norm_act(x.to(device), ln.weight.detach().to(device), ln.bias.detach().to(device), 1e-3, "ln", act, inplace=False)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    scale, shift = torch.rand(c) + 0.5, torch.randn(c)
    want2 = f(x * scale + shift)
    got2 = ops.norm_act(x.to(device), scale.to(device), shift.to(device), 0.0, "affine", act, inplace=False)
    # GELU's 1 + erf(x/sqrt2) cancels for x << 0: absolute error of a few 1e-6 on tiny outputs, both sides in fp32
    np.testing.assert_allclose(got2.cpu().numpy(), want2.numpy(), rtol=1e-6, atol=5e-6)


@pytest.mark.gpu
def test_gelu_form_vs_float64_erf_on_1e7_points(ops, device):
    """The library's one GELU (csrc/common.h: max(y, 0) - t 2^P(t), t = |y| / sqrt 2, P a degree-8 fit of log2 erfc(t) - 1/2;
    tools/fit_gelu_poly.py) against float64 erf on 1e7 points: a dense sweep of [-10, 10], normal draws at three widths and every
    magnitude down to the denormals.  Bound: 6e-8 beyond the rounding of the fp32 result (the rational form it replaced sat at
    2.2e-7); far tails: 0 <= -GELU(y) <= 1e-8 for y <= -6, GELU(y) == y for y >= 6; no NaN for finite inputs of any size."""
    g = torch.Generator().manual_seed(11)
    parts = [torch.linspace(-10, 10, 6_000_000, dtype=torch.float64).float(), torch.randn(1_500_000, generator=g),
             torch.randn(1_500_000, generator=g) * 3, torch.randn(500_000, generator=g) * 0.05,
             (torch.rand(500_000, generator=g) * 2 - 1) * torch.pow(10.0, -torch.rand(500_000, generator=g) * 44)]
    x = torch.cat(parts).view(-1, 128)
    assert x.numel() >= 9_999_000
    one, zero = torch.ones(128, device=device), torch.zeros(128, device=device)
    got = ops.norm_act(x.to(device), one, zero, 0.0, "affine", "gelu", inplace=False).cpu().double()
    xd = x.double()
    want = 0.5 * xd * (1 + torch.erf(xd / 2 ** 0.5))
    half_ulp = torch.from_numpy(np.spacing(np.abs(want.float().numpy()))).double() * 0.5
    beyond = ((got - want).abs() - half_ulp).clamp_min(0)
    assert beyond.max().item() <= 6e-8, (beyond.max().item(), x.view(-1)[beyond.argmax()].item())
    far = torch.tensor([[-1e30, -1e10, -40.0, -13.0, -6.0, 6.0, 13.0, 40.0, 1e10, 1e30, 3e38, -3e38] + [0.0] * 116])
    y = ops.norm_act(far.to(device), one, zero, 0.0, "affine", "gelu", inplace=False).cpu()[0]
    assert torch.isfinite(y).all()
    assert (y[:5] <= 0).all() and (y[:5] >= -1e-8).all() and y[11] <= 0 and y[11] >= -1e-8
    assert torch.equal(y[5:11], far[0, 5:11]) and (y[12:] == 0).all()


# ------------------------------------------------------------------------------- refine-stage ops (K17 / K20)
def random_rois(rng, r, spread=40.0):
    ctr = rng.uniform(-spread, spread, (r, 2))
    z = rng.uniform(-2.5, -0.5, (r, 1))
    wlh = np.stack([rng.uniform(0.5, 2.5, r), rng.uniform(0.8, 6.0, r), rng.uniform(1.0, 3.0, r)], 1)
    rz = rng.uniform(-np.pi, np.pi, (r, 1))
    return np.concatenate([ctr, z, wlh, rz], 1).astype(np.float32)


@pytest.mark.parametrize("r,p,max_inbox,max_all", [(37, 20000, 512, 50000), (300, 60000, 16, 50000), (300, 60000, 512, 700),
                                                   (1, 5000, 512, 50000), (700, 3000, 512, 50000),
                                                   # several chunks of RoI groups: cap inside an early chunk / never reached
                                                   (2000, 20000, 32, 3000), (2000, 20000, 512, 10 ** 6), (1100, 8000, 4, 900)])
def test_dynamic_point_pool_vs_oracle(ops, device, r, p, max_inbox, max_all):
    rng = np.random.default_rng(r * 31 + p)
    rois = random_rois(rng, r)
    # points: half uniform, half clustered around RoI centres so boxes are well populated
    which = rng.integers(0, r, p // 2)
    near = rois[which, :3] + rng.normal(0, 1.5, (p // 2, 3)) + np.array([0, 0, 1.0])
    pts = np.concatenate([near, np.concatenate([rng.uniform(-45, 45, (p - p // 2, 2)), rng.uniform(-3, 2, (p - p // 2, 1))], 1)])
    pts = np.concatenate([pts, rng.random((p, 2))], 1).astype(np.float32)  # stride-5 rows like the real points
    extra = [1.0, 1.0, 1.0]
    # full (uncapped) oracle membership with boundary margins; a (point, roi) pair within 1e-4 of a box face may
    # legitimately flip under device sinf/cosf rounding
    wp, wr, wf, near_pairs = orefine.dynamic_point_pool(rois, pts[:, :3], extra, 10 ** 9, 10 ** 9, return_margin=True)
    risky = near_pairs[near_pairs[:, 2] < 1e-4]
    gp, gr, gf = ops.dynamic_point_pool(torch.from_numpy(rois).to(device), torch.from_numpy(pts).to(device), extra,
                                        max_inbox, max_all)
    gp, gr, gf = gp.cpu().numpy(), gr.cpu().numpy(), gf.cpu().numpy()
    if len(risky) == 0:
        cp, cr, cf = orefine.dynamic_point_pool(rois, pts[:, :3], extra, max_inbox, max_all)
        np.testing.assert_array_equal(gr, cr)
        np.testing.assert_array_equal(gp, cp)
        np.testing.assert_array_equal(gf[:, :3], cf[:, :3])
        np.testing.assert_allclose(gf[:, 3:12], cf[:, 3:12], atol=2e-5)
        np.testing.assert_array_equal(gf[:, 12], cf[:, 12])
    else:  # the canonical order and caps must still hold, membership must agree away from the faces
        risky_set = {(int(a), int(b)) for a, b, _ in risky}
        full = {(int(a), int(b)) for a, b in zip(wr, wp)}
        got = {(int(a), int(b)) for a, b in zip(gr, gp)}
        assert all(pair in full or pair in risky_set for pair in got)
    # structural invariants (dynamic_point_roi_extractor.py:83-92) at any size
    key = gr * (p + 1) + gp
    assert (np.diff(key) > 0).all()                       # ascending (roi, point), no duplicates
    assert len(gp) <= max_all and (np.bincount(gr, minlength=r) <= max_inbox).all()
    roi = rois[gr]
    np.testing.assert_array_equal(gf[:, :3], pts[gp, :3])
    np.testing.assert_allclose(gf[:, 6] + gf[:, 9], roi[:, 4], atol=1e-5)   # length
    np.testing.assert_allclose(gf[:, 7] + gf[:, 10], roi[:, 3], atol=1e-5)  # width
    np.testing.assert_allclose(gf[:, 8] + gf[:, 11], roi[:, 5], atol=1e-5)  # height
    assert (np.abs(gf[:, 3]) < roi[:, 4] + extra[0] + 1e-5).all() and (np.abs(gf[:, 4]) < roi[:, 3] + extra[1] + 1e-5).all()
    assert len(gp) > 0


@pytest.mark.parametrize("max_inbox,max_all", [(512, 50000), (2048, 10 ** 6), (64, 3000)])
def test_dynamic_point_pool_binned_equals_brute_force(ops, device, max_inbox, max_all, monkeypatch):
    """The cell-binned path == the P x R brute-force passes, bit for bit: RoIs with far more hits than the LDS list (bisection on
    the point index), a box larger than the cell table walk allows, boxes and points beyond the table's border cells, NaN points."""
    rng = np.random.default_rng(max_inbox)
    r, p = 1500, 200000
    rois = random_rois(rng, r, spread=45.0)
    rois[0] = [0, 0, -2, 30, 60, 4, 0.3]            # ~1/4 of the scene: tens of thousands of hits
    rois[1] = [3000.0, -2500.0, -2, 4, 9, 3, 1.0]   # beyond the border cells (they collect everything outside +-2 km)
    rois[2] = [1e6, 1e6, 0, 1e7, 1e7, 1e7, 0.0]     # covers everything: whole-array walk
    rois[3, :2] = [2.0, 2.0]
    rois[3, 3:6] = [6.0, 12.0, 4.0]                 # dense centre: > 2048 hits
    pts = np.concatenate([rng.normal(0, 6, (p // 2, 2)), rng.uniform(-3, 1, (p // 2, 1))], 1)
    pts = np.concatenate([pts, np.concatenate([rng.uniform(-50, 50, (p - p // 2, 2)), rng.uniform(-3, 1, (p - p // 2, 1))], 1)])
    pts[:40, :2] = [3000.0, -2500.0] + rng.normal(0, 1, (40, 2))
    pts[40:45] = np.nan
    pts = pts[rng.permutation(p)].astype(np.float32)
    d_rois, d_pts = torch.from_numpy(rois.astype(np.float32)).to(device), torch.from_numpy(pts).to(device)
    old = ops.set_option(ops.OPT_POOL_BRUTE, 1)  # (fsf_set_option: the library reads no environment variable per call)
    try:
        bp, br, bf = ops.dynamic_point_pool(d_rois, d_pts, [1.0, 1.0, 1.0], max_inbox, max_all)
        ops.set_option(ops.OPT_POOL_BRUTE, 0)
        gp, gr, gf = ops.dynamic_point_pool(d_rois, d_pts, [1.0, 1.0, 1.0], max_inbox, max_all)
    finally:
        ops.set_option(ops.OPT_POOL_BRUTE, old)
    assert gp.numel() == bp.numel() and gp.numel() > 0
    assert torch.equal(gp, bp) and torch.equal(gr, br) and torch.equal(gf, bf)
    assert int((gr == 0).sum()) == min(max_inbox, max_all)


def test_dynamic_point_pool_batched_and_empty(ops, device):
    rng = np.random.default_rng(5)
    rois = random_rois(rng, 40, spread=10.0)
    pts = np.concatenate([rng.uniform(-12, 12, (8000, 2)), rng.uniform(-3, 2, (8000, 1))], 1).astype(np.float32)
    pb = np.sort(rng.integers(0, 2, 8000)).astype(np.int64)
    rb = np.sort(rng.integers(0, 2, 40)).astype(np.float32)
    rois8 = np.concatenate([rb[:, None], rois], 1)
    gp, gr, gf = ops.dynamic_point_pool(torch.from_numpy(rois8).to(device), torch.from_numpy(pts).to(device), [0.5, 0.5, 0.5],
                                        512, roi_batch_col=0, box_col=1, pts_batch=torch.from_numpy(pb).to(device))
    want_p, want_r = [], []
    for b in range(2):  # the reference loops over samples and offsets the indices (dynamic_point_roi_extractor.py:43-72)
        pm, rm = np.nonzero(pb == b)[0], np.nonzero(rb == b)[0]
        p_, r_, _ = orefine.dynamic_point_pool(rois[rm], pts[pm], [0.5, 0.5, 0.5], 512)
        want_p.append(pm[p_])
        want_r.append(rm[r_])
    np.testing.assert_array_equal(gr.cpu().numpy(), np.concatenate(want_r))
    np.testing.assert_array_equal(gp.cpu().numpy(), np.concatenate(want_p))
    # nothing inside any box
    far = torch.full((100, 3), 500.0, device=device)
    gp, gr, gf = ops.dynamic_point_pool(torch.from_numpy(rois).to(device), far, [0.5, 0.5, 0.5], 512)
    assert gp.numel() == 0 and gf.shape == (0, 13)


@pytest.mark.parametrize("c,hf,wf", [(32, 57, 100), (64, 113, 200), (3, 900, 1600), (130, 29, 50)])
def test_project_gather_bilinear_vs_torch_grid_sample(ops, device, c, hf, wf):
    """fsf_project_gather_bilinear (north_star's per-point bilinear image-feature gather) against
    F.grid_sample(bilinear, align_corners=False, zeros) at the reference projection (oracle.project.gather_bilinear), 1e-4:
    NCHW and channels-last feature maps, per-camera and camera-summed outputs, the visible-camera count."""
    from fullysparsefusion_amd import synthetic

    rng = np.random.default_rng(c + hf)
    f = synthetic.make_frame(num_sweeps=1, seed=3)
    pts = f["points"][:20000, 5:8].copy()
    pts[:50] = rng.uniform(-60, 60, (50, 3))  # some far / behind-camera / out-of-image points
    L = f["lidar2img"]
    feat = rng.standard_normal((6, c, hf, wf)).astype(np.float32)
    want, valid = oproj.gather_bilinear(pts, L, feat, 900, 1600)
    d_pts, d_L, d_feat = torch.from_numpy(pts).to(device), torch.from_numpy(L).to(device), torch.from_numpy(feat).to(device)
    got = ops.project_gather_bilinear(d_pts, d_L, d_feat, (900, 1600))
    assert got.shape == (pts.shape[0], 6, c)
    scale = max(1.0, float(np.abs(want).max()))
    assert float(np.abs(got.cpu().numpy() - want).max()) <= 1e-4 * scale
    assert not got.cpu().numpy()[~valid].any() and valid.sum() > 1000
    cl = d_feat.permute(0, 2, 3, 1).contiguous()
    got_cl, count = ops.project_gather_bilinear(d_pts, d_L, cl, (900, 1600), channels_last=True, return_count=True)
    assert float((got_cl - got).abs().max()) <= 1e-5 * scale
    np.testing.assert_array_equal(count.cpu().numpy(), valid.sum(1).astype(np.uint8))
    for layout_feat, last in ((d_feat, False), (cl, True)):
        red = ops.project_gather_bilinear(d_pts, d_L, layout_feat, (900, 1600), channels_last=last, reduce_cams=True)
        assert float(np.abs(red.cpu().numpy() - want.sum(1)).max()) <= 2e-4 * scale


def test_ops_dynamic_point_pool_reference_signature(ops, device):
    """`ops.dynamic_point_pool(rois, pts, extra_wlh, max_inbox_point, max_all_pts=50000)` by the reference's name
    (projects/mmdet3d_plugin/ops/__init__.py:1, dynamic_point_pool_op.py:10-51): same rows as the oracle, the fake
    (-1, -1, zeros) row when nothing is inside, non-differentiable outputs, None gradients."""
    from fullysparsefusion_amd.mmdet3d_plugin.ops import dynamic_point_pool

    rng = np.random.default_rng(11)
    rois = random_rois(rng, 30, spread=10.0)
    pts = np.concatenate([rng.uniform(-12, 12, (6000, 2)), rng.uniform(-3, 2, (6000, 1))], 1).astype(np.float32)
    wp, wr, wf, margin = orefine.dynamic_point_pool(rois, pts, [0.5, 0.5, 0.5], 64, return_margin=True)
    r_t = torch.from_numpy(rois).to(device).requires_grad_(True)
    p_t = torch.from_numpy(pts).to(device).requires_grad_(True)
    gp, gr, gf = dynamic_point_pool(r_t, p_t, [0.5, 0.5, 0.5], 64)
    assert gp.dtype == torch.int64 and gr.dtype == torch.int64 and gf.dtype == torch.float32 and gf.shape[1] == 13
    assert not gp.requires_grad and not gr.requires_grad and not gf.requires_grad
    if not (margin[:, 2] < 1e-4).any():
        np.testing.assert_array_equal(gp.cpu().numpy(), wp)
        np.testing.assert_array_equal(gr.cpu().numpy(), wr)
        np.testing.assert_allclose(gf.cpu().numpy(), wf, atol=1e-5)
    # max_all_pts caps the output in (roi, point) order
    cp, cr, _ = dynamic_point_pool(r_t, p_t, [0.5, 0.5, 0.5], 64, 100)
    assert cp.numel() == min(100, gp.numel()) and torch.equal(cp, gp[:100]) and torch.equal(cr, gr[:100])
    # nothing inside any box: the reference's fake non-empty row
    far = torch.full((100, 3), 500.0, device=device)
    fp, fr, ff = dynamic_point_pool(torch.from_numpy(rois).to(device), far, [0.5, 0.5, 0.5], 64)
    assert fp.tolist() == [-1] and fr.tolist() == [-1] and ff.shape == (1, 13) and float(ff.abs().sum()) == 0.0


@pytest.mark.parametrize("n,rotated", [(1, True), (70, True), (500, True), (500, False), (1500, True)])
def test_nms_bev_vs_oracle(ops, device, n, rotated):
    rng = np.random.default_rng(n + int(rotated))
    # clustered boxes so that suppression chains exist
    nclu = max(1, n // 6)
    ctr = rng.uniform(-40, 40, (nclu, 2))[rng.integers(0, nclu, n)] + rng.normal(0, 0.7, (n, 2))
    wl = np.stack([rng.uniform(1.5, 2.5, n), rng.uniform(3.5, 5.5, n)], 1)
    yaw = rng.uniform(-np.pi, np.pi, (n, 1))
    boxes = np.concatenate([ctr - wl / 2, ctr + wl / 2, yaw], 1).astype(np.float32)  # xywhr2xyxyr layout, score order
    thresh = 0.25
    iou = orefine.iou_bev_matrix(boxes, rotated) if n <= 500 else None
    keep = ops.nms_bev(torch.from_numpy(boxes).to(device), thresh, rotated).cpu().numpy()
    assert (np.diff(keep) > 0).all() and (len(keep) == 0 or keep[0] == 0)
    if iou is not None:
        borderline = np.abs(iou - thresh) < 1e-4
        if not borderline.any():
            np.testing.assert_array_equal(keep, orefine.nms_from_iou(iou, thresh))
        kept = np.zeros(n, bool)
        kept[keep] = True
        # greedy-NMS invariants under the float64 IoU, away from the threshold: kept boxes do not suppress each other and
        # every dropped box is suppressed by an earlier kept one
        for i in keep:
            assert not (kept[i + 1:] & (iou[i, i + 1:] > thresh + 1e-4)).any()
        for j in np.nonzero(~kept)[0]:
            assert (iou[keep[keep < j], j] > thresh - 1e-4).any()
    else:  # large: self-consistency via a second, permuted-but-equivalent call and idempotence
        again = ops.nms_bev(torch.from_numpy(boxes[keep]).to(device), thresh, rotated).cpu().numpy()
        np.testing.assert_array_equal(again, np.arange(len(keep)))


def test_nms_bev_multiclass_equals_per_class_calls(ops, device):
    """One batched call == one fsf_nms_bev call per class on the class's own score order (incl. an empty class)."""
    rng = np.random.default_rng(21)
    n, c = 900, 5
    ctr = rng.uniform(-40, 40, (150, 2))[rng.integers(0, 150, n)] + rng.normal(0, 0.7, (n, 2))
    wl = np.stack([rng.uniform(1.5, 2.5, n), rng.uniform(3.5, 5.5, n)], 1)
    boxes = torch.from_numpy(np.concatenate([ctr - wl / 2, ctr + wl / 2, rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)).to(device)
    scores = torch.from_numpy(rng.random((c, n)).astype(np.float32)).to(device)
    scores[3] = 0.0  # nothing above the threshold
    valid = scores > 0.3
    order = torch.where(valid, scores, scores.new_full((), float("-inf"))).sort(dim=1, descending=True, stable=True)[1]
    count = valid.sum(1, dtype=torch.int32)
    pos = torch.arange(n, device=device, dtype=torch.int32).expand(c, n)
    rank = torch.where(valid, torch.empty_like(pos).scatter_(1, order, pos), pos.new_full((), -1))
    keep, num = ops.nms_bev_multiclass(boxes, rank, count, 0.25, True)
    for k in range(c):
        nk = int(count[k])
        want = ops.nms_bev(boxes[order[k, :nk]], 0.25, True)
        assert int(num[k]) == want.numel()
        assert torch.equal(keep[k, : int(num[k])], want)
    assert int(num[3]) == 0
    # capped: every class scan stops after its first `cap` keeps == the head of the uncapped list
    for cap in (1, 37, 64, 100000):
        keep_c, num_c = ops.nms_bev_multiclass(boxes, rank, count, 0.25, True, max_keep=cap)
        for k in range(c):
            m = min(int(num[k]), cap)
            assert int(num_c[k]) == m
            assert torch.equal(keep_c[k, :m], keep[k, :m])
        keep_w, num_w, flag = ops.nms_bev_multiclass(boxes, rank, count, 0.25, True, max_keep=cap, windowed=True)
        assert int(flag) == 0  # (900 boxes: the window of max(4 cap, 2048) covers every class)
        assert torch.equal(num_w, num_c) and all(torch.equal(keep_w[k, :int(num_c[k])], keep_c[k, :int(num_c[k])]) for k in range(c))


def test_nms_bev_multiclass_window_and_incomplete_flag(ops, device):
    """Windowed per-class masks (each class's best max(4 cap, 2048) boxes): with spread-out boxes the first `cap` keeps lie inside
    the window and the result equals the full-mask call; with 6 000 copies of a few boxes a class keeps fewer than `cap` inside its
    window although it has more boxes — the flag is raised (the caller repeats on full masks)."""
    rng = np.random.default_rng(2)
    n, c, cap = 6000, 3, 100
    def make(ctr):
        wl = np.stack([rng.uniform(1.5, 2.5, n), rng.uniform(3.5, 5.5, n)], 1)
        return torch.from_numpy(np.concatenate([ctr - wl / 2, ctr + wl / 2, rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)).to(device)
    scores = torch.from_numpy(rng.random((c, n)).astype(np.float32)).to(device)
    valid = scores > 0.05
    order = torch.where(valid, scores, scores.new_full((), float("-inf"))).sort(dim=1, descending=True, stable=True)[1]
    count = valid.sum(1, dtype=torch.int32)
    pos = torch.arange(n, device=device, dtype=torch.int32).expand(c, n)
    rank = torch.where(valid, torch.empty_like(pos).scatter_(1, order, pos), pos.new_full((), -1))
    spread = make(rng.uniform(-300, 300, (n, 2)))
    full_k, full_n = ops.nms_bev_multiclass(spread, rank, count, 0.25, True, max_keep=cap)
    win_k, win_n, flag = ops.nms_bev_multiclass(spread, rank, count, 0.25, True, max_keep=cap, windowed=True)
    assert int(flag) == 0 and torch.equal(win_n, full_n)
    assert all(torch.equal(win_k[k, :int(full_n[k])], full_k[k, :int(full_n[k])]) for k in range(c))
    stacked = make(rng.uniform(-20, 20, (40, 2))[rng.integers(0, 40, n)] + rng.normal(0, 0.05, (n, 2)))
    full_k, full_n = ops.nms_bev_multiclass(stacked, rank, count, 0.25, True, max_keep=cap)
    assert int(full_n.max()) < cap                       # ~40 survivors per class: the cap is never reached
    _, win_n, flag = ops.nms_bev_multiclass(stacked, rank, count, 0.25, True, max_keep=cap, windowed=True)
    assert int(flag) == 1                                # 6 000 boxes per class > the 2 048-box window


# ------------------------------------------------------------------------------------------ K21 SIR-layer input
@pytest.mark.parametrize("p,cf,ce,r,act", [(5, 175, 0, 3, "gelu"), (5, 128, 0, 3, "gelu"), (5, 163, 13, 13, "gelu"),
                                           (4, 128, 0, 3, "gelu"), (5, 40, 0, 3, "relu"), (5, 251, 0, 3, "gelu")])
def test_sir_input_vs_torch_composition(ops, device, p, cf, ce, r, act):
    """cat -> xyz normalisation -> rel_mlp (3 x Linear/LayerNorm/act) -> product, against the same steps in torch fp64."""
    import torch.nn.functional as F

    torch.manual_seed(p * 100 + cf)
    n, c = 5003, p + cf + ce
    big = torch.randn(n, p + 3, device=device)
    points = big[:, :p]                      # a row-strided view, like points[:, :5] of the [N, 8] input
    feats = torch.randn(n, cf, device=device)
    fcl = torch.randn(n, r, device=device) * torch.tensor([3.0] + [0.5] * (r - 1), device=device)
    fcl[:7] = 0.0                            # singleton clusters: f_cluster exactly zero
    extra = fcl if ce else None
    dims = [r, 16, 32, c]
    layers = []
    for i in range(3):
        layers.append((torch.randn(dims[i + 1], dims[i], device=device) / dims[i] ** 0.5,
                       torch.rand(dims[i + 1], device=device) + 0.5, torch.randn(dims[i + 1], device=device) * 0.1))
    norm = [20.0, 20.0, 4.0]
    out = ops.sir_input(points, feats, fcl, norm, (*layers, 1e-3), act, 10.0, extra=extra, extra_div=10.0)
    d = lambda t: t.double()  # noqa: E731
    x = torch.cat([d(points[:, :3]) / torch.tensor(norm, device=device, dtype=torch.float64), d(points[:, 3:]), d(feats)] +
                  ([d(extra) / 10.0] if ce else []), 1)
    h = d(fcl) / 10.0
    for w, g, b in layers:
        h = F.layer_norm(F.linear(h, d(w)), (w.size(0),), d(g), d(b), 1e-3)
        h = F.gelu(h) if act == "gelu" else F.relu(h)
    want = x * h
    assert out.shape == (n, c)
    err = (out.double() - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), err
    assert torch.equal(out, ops.sir_input(points, feats, fcl, norm, (*layers, 1e-3), act, 10.0, extra=extra, extra_div=10.0))


@pytest.mark.parametrize("n,k,c,norm,act,bias", [(50021, 256, 128, "ln", "gelu", False), (20000, 180, 128, "ln", "relu", False),
                                                 (7001, 133, 128, "ln", "gelu", True), (30000, 11, 64, "affine", "relu", False),
                                                 (513, 128, 128, "none", "none", True), (1, 64, 32, "ln", "gelu", False),
                                                 (40000, 128, 64, "affine", "gelu", True),
                                                 (10641, 1024, 1024, "none", "none", True), (5000, 768, 1024, "affine", "relu", False),
                                                 (3001, 128, 132, "none", "none", True),
                                                 # >= 65 536 rows (sizes the weight-resident experiment K22r, profiles/r5_k22r_*, was checked at)
                                                 (100003, 131, 128, "affine", "relu", False), (80000, 180, 128, "ln", "gelu", True),
                                                 (65536, 64, 64, "ln", "relu", False)])
def test_linear_norm_act_split_bf16_is_fp32_accurate(ops, device, n, k, c, norm, act, bias):
    """K22: Linear -> LayerNorm / affine -> act with the product formed from the exact 3-way bf16 split (six cross terms).
    Against float64: the error must be of the size of an fp32 GEMM's own error (compared with torch's fp32 F.linear on the
    same data), on inputs with a wide dynamic range; row-strided input; deterministic."""
    import torch.nn.functional as F

    torch.manual_seed(n + k)
    kpad = (k + 3) // 4 * 4
    xbuf = torch.full((n, kpad + 4), float("nan"), device=device)  # NaN padding: columns >= k must never be read into the sum
    x = xbuf[:, :k]
    x.copy_(torch.randn(n, k, device=device) * torch.exp(torch.randn(n, 1, device=device) * 2.0))
    w = torch.randn(c, k, device=device) / k ** 0.5
    b = torch.randn(c, device=device) if bias else None
    g = torch.rand(c, device=device) + 0.5
    be = torch.randn(c, device=device) * 0.1
    planes = ops.linear_prepare_weight(w)
    assert ops.linear_norm_act_supported(x, c)
    out = ops.linear_norm_act(x, planes, c, bias=b, norm=norm, gamma=g if norm != "none" else None,
                              beta=be if norm != "none" else None, eps=1e-3, act=act)

    def tail(y):
        if norm == "ln":
            y = F.layer_norm(y, (c,), g.to(y.dtype), be.to(y.dtype), 1e-3)
        elif norm == "affine":
            y = y * g.to(y.dtype) + be.to(y.dtype)
        return F.gelu(y) if act == "gelu" else F.relu(y) if act == "relu" else y

    want = tail(F.linear(x.double(), w.double(), b.double() if bias else None))
    ref32 = tail(F.linear(x.contiguous(), w, b))                      # what the library path computes in fp32
    scale = max(1.0, float(want.abs().max()))
    err = float((out.double() - want).abs().max())
    err32 = float((ref32.double() - want).abs().max())
    assert err <= max(2.0 * err32, 2e-6 * scale), (err, err32)
    assert err <= 2e-5 * scale
    assert torch.equal(out, ops.linear_norm_act(x, planes, c, bias=b, norm=norm, gamma=g if norm != "none" else None,
                                                beta=be if norm != "none" else None, eps=1e-3, act=act))


@pytest.mark.parametrize("n,k,c,norm,act,addend", [(50021, 128, 128, "ln", "gelu", True), (20000, 180, 128, "ln", "relu", False),
                                                    (7001, 133, 64, "ln", "gelu", True), (30000, 11, 64, "affine", "relu", False),
                                                    (4099, 256, 256, "none", "none", False), (513, 96, 36, "ln", "gelu", True),
                                                    (1, 64, 128, "ln", "gelu", True), (100003, 131, 128, "affine", "relu", True)])
def test_linear_f16x3_in_kernel_split_vs_float64(ops, device, n, k, c, norm, act, addend):
    """K22f (fsf_linear_f16w_norm_act_grouped): fp32 x split IN the kernel into f16 hi | lo per row with a running power-of-two unit,
    W as f16 planes, three MFMA passes.  Against float64 on four kinds of rows — plain normal; a wide dynamic range inside every row;
    rows whose magnitude RISES along k by 1e6 (the unit falls mid-row: the accumulators are rescaled); rows whose first 32 columns are
    zero and the rest tiny (the first scale is the cap, later chunks lower it) — the error stays within 2 x that of torch's fp32
    F.linear (or 2e-6 of the output scale), like the exact bf16 x 6 form, and the two forms agree to 1e-5 of the scale."""
    torch.manual_seed(n + k)
    F = torch.nn.functional
    w = torch.randn(c, k) / k ** 0.5
    b = torch.randn(c)
    g, be = torch.rand(c) + 0.5, torch.randn(c)
    table = torch.randn(257, c) * 2 if addend else None
    idx = torch.randint(0, 257, (n,)) if addend else None
    kinds = [torch.randn(n, k), torch.randn(n, k) * torch.exp(torch.randn(n, k) * 4),
             torch.randn(n, k) * torch.logspace(-3, 3, k)[None, :],
             torch.cat([torch.zeros(n, min(32, k)), torch.randn(n, max(k - 32, 0)) * 1e-12], 1)[:, :k]]
    for kind, x in enumerate(kinds):
        x = x.contiguous()
        xs_ = torch.zeros(n, (k + 3) // 4 * 4)
        xs_[:, :k] = x
        xd = xs_.to(device)[:, :k]
        def ref(dt):
            z = F.linear(x.to(dt), w.to(dt), b.to(dt))
            if addend:
                z = z + table.to(dt)[idx]
            if norm == "ln":
                z = F.layer_norm(z, (c,), g.to(dt), be.to(dt), 1e-3)
            elif norm == "affine":
                z = z * g.to(dt) + be.to(dt)
            return F.gelu(z) if act == "gelu" else F.relu(z) if act == "relu" else z
        want, ref32 = ref(torch.float64), ref(torch.float32)
        kw = dict(bias=b.to(device), norm=norm, gamma=g.to(device) if norm != "none" else None, beta=be.to(device) if norm != "none" else None,
                  eps=1e-3, act=act, row_add=table.to(device) if addend else None, row_add_index=idx.to(device) if addend else None)
        got = {}
        for fmt in ("f16x3", "bf16x6"):
            planes = ops.linear_prepare_weight(w.to(device), fmt=fmt)
            assert ops.linear_weight_is_f16(planes) == (fmt == "f16x3")
            with pytest.raises(ops.FsfHipError, match="format tag"):  # (round 6: the format is a tag on the tensor, never inferred from its size)
                ops.linear_weight_is_f16(planes.clone())
            got[fmt] = ops.linear_norm_act(xd, planes, c, **kw).cpu()
        scale = float(want.abs().max())
        err32 = float((ref32.double() - want).abs().max())
        for fmt, y in got.items():
            err = float((y.double() - want).abs().max())
            assert torch.isfinite(y).all() and err <= max(2.0 * err32, 2e-6 * scale), (kind, fmt, err, err32, scale)
        assert float((got["f16x3"] - got["bf16x6"]).abs().max()) <= 1e-5 * scale, kind


def test_linear_f16x3_addend_magnitudes(ops, device):
    """K22f keeps the per-row addend in the accumulators in the unit s_x * s_w; the first scale of a row is capped so that s_x * s_w <=
    2^40: addends up to 1e20 beside inputs of any size come through to fp32 accuracy, rows of denormal-small inputs contribute nothing
    measurable beside an O(1) addend."""
    torch.manual_seed(5)
    n, k, c = 4096, 128, 128
    w = (torch.randn(c, k) / k ** 0.5).to(device)
    for xmag, amag in [(1.0, 1e20), (1e-30, 1.0), (1e20, 1e-20), (1e-6, 1e6)]:
        x = (torch.randn(n, k) * xmag).to(device)
        table = (torch.randn(64, c) * amag).to(device)
        idx = torch.randint(0, 64, (n,), device=device)
        y = ops.linear_norm_act(x, ops.linear_prepare_weight(w, fmt="f16x3"), c, row_add=table, row_add_index=idx)
        want = x.double() @ w.double().t() + table.double()[idx]
        assert torch.isfinite(y).all()
        assert float((y.double() - want).abs().max()) <= 4e-6 * float(want.abs().max()), (xmag, amag)


@pytest.mark.parametrize("n,cin,cout", [(101119, 256, 128), (1517, 1024, 512), (33, 128, 64), (1, 8, 4)])
def test_channel_group_sum_add_equals_torch(ops, device, n, cin, cout):
    """The U-Net decoder's `features.view(n, C, -1).sum(2) + merge` in one pass: bit-identical to the two torch ops."""
    torch.manual_seed(n)
    f = torch.randn(n, cin, device=device)
    m = torch.randn(n, cout, device=device)
    want = m + f.view(n, cout, -1).sum(dim=2)
    assert torch.equal(ops.channel_group_sum_add(f, cout, add=m), want)
    assert torch.equal(ops.channel_group_sum_add(f, cout), f.view(n, cout, -1).sum(dim=2))


@pytest.mark.parametrize("n,ca,cb", [(101119, 128, 128), (7441, 256, 256), (36508, 128, 128), (33, 8, 24), (1, 16, 8)])
def test_channel_pair_sum_add2_equals_the_op_on_the_concatenation(ops, device, n, ca, cb):
    """The decoder shortcut read from the two tensors `decoder_layer_forward` concatenates: bit-identical to torch's
    `cat([a, b], 1).view(n, C, 2).sum(2) + merge` (and so to fsf_channel_group_sum_add) without writing the concatenation;
    widths that are not multiples of 8 are refused."""
    torch.manual_seed(n + ca)
    a, b = torch.randn(n, ca, device=device), torch.randn(n, cb, device=device)
    cout = (ca + cb) // 2
    m = torch.randn(n, cout, device=device)
    cat = torch.cat([a, b], 1)
    assert ops.channel_pair_sum_add2_supported(a, b)
    assert torch.equal(ops.channel_pair_sum_add2(a, b, add=m), m + cat.view(n, cout, 2).sum(dim=2))
    assert torch.equal(ops.channel_pair_sum_add2(a, b), cat.view(n, cout, 2).sum(dim=2))
    assert torch.equal(ops.channel_pair_sum_add2(a, b, add=m), ops.channel_group_sum_add(cat, cout, add=m))
    assert not ops.channel_pair_sum_add2_supported(torch.zeros(n, 12, device=device), torch.zeros(n, 4, device=device))


@pytest.mark.parametrize("n,m,c", [(92639, 92639, 64), (5000, 777, 128), (300, 1, 256), (10, 0, 32)])
def test_to_planes_rows_equals_to_planes_of_the_gathered_rows(ops, device, n, m, c):
    """fsf_to_planes_rows (round 6): the planes of feat[row_index] — bit for bit what fsf_to_planes makes of the gathered rows (row-strided
    source included), without writing them."""
    torch.manual_seed(n + c)
    wide = torch.randn(n, c + 8, device=device) * torch.exp(torch.randn(n, 1, device=device))
    feat = wide[:, :c]
    idx = torch.randperm(n, device=device)[:m] if m <= n else torch.randint(0, n, (m,), device=device)
    want = ops.to_planes(feat.index_select(0, idx).contiguous())
    got = ops.to_planes(feat, row_index=idx)
    assert got.m == want.m == m and got.c == c and torch.equal(got.data, want.data) and torch.equal(got.scales, want.scales)


@pytest.mark.parametrize("n,ca,cb", [(101119, 128, 128), (5003, 64, 64), (1, 128, 128), (0, 128, 128), (777, 256, 256), (4099, 32, 96)])
def test_channel_pair_sum_add2_planes_equals_sum_then_to_planes(ops, device, n, ca, cb):
    """fsf_channel_pair_sum_add2_planes (round 6): the decoder shortcut leaving as planes for the level's upsampling convolution — the
    planes and the scales fsf_to_planes makes of fsf_channel_pair_sum_add2's rows, bit for bit (zero row behind the last one included),
    with and without the addend."""
    torch.manual_seed(n + ca)
    a, b = torch.randn(n, ca, device=device) * 3, torch.randn(n, cb, device=device)
    if n > 3:
        a[1] = 0.0
        b[1] = 0.0          # an all-zero row: scale of the empty maximum
        a[2] *= 1e-30       # tiny values next to ordinary ones
    cout = (ca + cb) // 2
    m = torch.randn(n, cout, device=device)
    for add in (m, None):
        want = ops.to_planes(ops.channel_pair_sum_add2(a, b, add=add))
        got = ops.channel_pair_sum_add2_planes(a, b, add=add)
        assert got.m == want.m == n and got.c == want.c == cout
        assert torch.equal(got.data, want.data) and torch.equal(got.scales, want.scales)


@pytest.mark.parametrize("n,w,k", [(20000, 60, 2), (3000, 60, 4), (17, 7, 7), (1, 128, 5)])
def test_row_topk_desc_equals_torch_topk(ops, device, n, w, k):
    torch.manual_seed(n + w)
    x = torch.zeros((n, w), dtype=torch.int64, device=device)
    sel = torch.rand(n, w, device=device) < 0.08
    x[sel] = torch.randint(1, 251, (int(sel.sum()),), device=device)
    x[0, :] = 5  # ties
    got = ops.row_topk_desc(x, k)
    assert torch.equal(got, x.topk(k, dim=-1)[0])


@pytest.mark.parametrize("n,cin,cout", [(50000, 180, 128), (20001, 136, 128), (70000, 256, 128), (33, 64, 64)])
def test_linear_weight_gradient_through_identity_pairs(ops, device, n, cin, cout):
    """X^T dY via K10's identity pairing == the autograd weight gradient of nn.Linear (float64 yardstick)."""
    torch.manual_seed(n)
    x = torch.randn(n, cin, device=device)
    g = torch.randn(n, cout, device=device)
    got = ops.linear_backward_weight(x, g)
    want = x.double().t() @ g.double()
    assert got.shape == (cin, cout)
    assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max()) * max(1.0, (n / 1e4) ** 0.5)
    assert torch.equal(got, ops.linear_backward_weight(x, g))


@pytest.mark.parametrize("c", [16, 32, 64, 128, 133, 180, 256, 512, 768, 1024])
@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_norm_act_backward_vs_torch_autograd(ops, device, c, act):
    """Fused LayerNorm + activation backward (grad_x, grad_gamma, grad_beta) against float64 autograd."""
    import torch.nn.functional as F

    torch.manual_seed(c)
    n = 30011 if c <= 256 else 16331  # (the 1024-wide LayerNorm + GELU of the query / refine heads: 16 k rows per step)
    x = torch.randn(n, c, device=device) * 2 + 0.5
    g = torch.rand(c, device=device) + 0.5
    b = torch.randn(c, device=device) * 0.2
    go = torch.randn(n, c, device=device)
    gx, dg, db = ops.norm_act_backward(x, go, g, b, 1e-3, act)
    xd, gd, bd = x.double().requires_grad_(), g.double().requires_grad_(), b.double().requires_grad_()
    y = F.layer_norm(xd, (c,), gd, bd, 1e-3)
    y = F.gelu(y) if act == "gelu" else F.relu(y)
    y.backward(go.double())
    assert float((gx.double() - xd.grad).abs().max()) <= 2e-5 * max(1.0, float(xd.grad.abs().max()))
    assert float((dg.double() - gd.grad).abs().max()) <= 1e-4 * max(1.0, float(gd.grad.abs().max()))
    assert float((db.double() - bd.grad).abs().max()) <= 1e-4 * max(1.0, float(bd.grad.abs().max()))
    gx2, dg2, db2 = ops.norm_act_backward(x, go, g, b, 1e-3, act)
    assert torch.equal(gx, gx2) and torch.equal(dg, dg2) and torch.equal(db, db2)  # deterministic


# ------------------------------------------------------------------------ K23: column statistics / training BatchNorm
@pytest.mark.parametrize("n,c", [(1, 5), (17, 3), (5000, 1), (4097, 16), (30011, 64), (200003, 128), (60000, 131), (9000, 300)])
def test_column_stats_vs_float64(ops, device, n, c):
    torch.manual_seed(n + c)
    x = torch.randn(n, c, device=device) * 3 + 100.0  # mean^2 >> var: E[x^2] - E[x]^2 would lose the variance
    s = ops.column_sum(x)
    want = x.double().sum(0)
    assert float((s.double() - want).abs().max()) <= 2e-6 * float(want.abs().max()) * max(1.0, (n / 1e4) ** 0.5)
    mean, var = ops.column_mean_var(x)
    assert float((mean.double() - x.double().mean(0)).abs().max()) <= 2e-6 * 100.0 * max(1.0, (n / 1e4) ** 0.5)
    wv = x.double().var(0, unbiased=False)
    assert float((var.double() - wv).abs().max()) <= 1e-4 * max(1.0, float(wv.max()))
    assert torch.equal(s, ops.column_sum(x))
    m2, v2 = ops.column_mean_var(x)
    assert torch.equal(mean, m2) and torch.equal(var, v2)


@pytest.mark.parametrize("n,c", [(2, 8), (30011, 64), (101119, 128), (7441, 256), (1517, 512), (5000, 131)])
@pytest.mark.parametrize("relu", [True, False])
def test_batch_norm_act_training_vs_torch_autograd(ops, device, n, c, relu):
    """Training-mode BatchNorm1d (+ ReLU) forward, running statistics and backward against float64 autograd."""
    from fullysparsefusion_amd.mmdet3d_plugin.ops.sst_ops import batch_norm_act_training

    torch.manual_seed(n + c)
    x = (torch.randn(n, c, device=device) * 2 + 0.7).requires_grad_()
    go = torch.randn(n, c, device=device)
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(device).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(device).double().train()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    y = batch_norm_act_training(bn, x, relu)
    assert y is not None
    y.backward(go)
    xd = x.detach().double().requires_grad_()
    yd = ref(xd)
    yd = torch.relu(yd) if relu else yd
    yd.backward(go.double())
    tol = lambda t: 2e-5 * max(1.0, float(t.abs().max()))
    assert float((y.double() - yd).abs().max()) <= tol(yd)
    assert float((x.grad.double() - xd.grad).abs().max()) <= tol(xd.grad)
    assert float((bn.weight.grad.double() - ref.weight.grad).abs().max()) <= 1e-4 * max(1.0, float(ref.weight.grad.abs().max()))
    assert float((bn.bias.grad.double() - ref.bias.grad).abs().max()) <= 1e-4 * max(1.0, float(ref.bias.grad.abs().max()))
    assert float((bn.running_mean.double() - ref.running_mean).abs().max()) <= 1e-6
    assert float((bn.running_var.double() - ref.running_var).abs().max()) <= 1e-5
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("n,c,relu", [(30011, 64, True), (101119, 128, True), (5000, 131, False)])
def test_fused_syncbn_relu_on_k23_vs_float64(ops, device, n, c, relu):
    """naiveSyncBN1d across ranks with the row passes on K23 (ops/norm.py::_SyncBatchNormAct): a one-rank process group on
    the device exercises the very code the ranks run (statistics kernel, packed all-reduce, fused normalise + ReLU, the
    single-rank backward kernel + the statistics-gradient correction); with one rank the result must equal training-mode
    BatchNorm1d, checked against float64 autograd.  (World size 2 is covered on CPU / gloo, tests/test_distributed_cpu.py.)"""
    import os
    import socket

    import torch.distributed as dist

    from fullysparsefusion_amd.mmdet3d_plugin.ops.norm import NaiveSyncBatchNorm1d, _SyncBatchNormAct

    own = not dist.is_initialized()
    if own:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]))
        s.close()
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        torch.manual_seed(n + c)
        x = (torch.randn(n, c, device=device) * 2 + 0.7).requires_grad_()
        go = torch.randn(n, c, device=device)
        bn = NaiveSyncBatchNorm1d(c, eps=1e-3, momentum=0.01).to(device).train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_()
        ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(device).double().train()
        ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
        y = _SyncBatchNormAct.apply(x, bn.weight, bn.bias, bn, relu, None)
        y.backward(go)
        xd = x.detach().double().requires_grad_()
        yd = ref(xd)
        yd = torch.relu(yd) if relu else yd
        yd.backward(go.double())
        tol = lambda t: 5e-5 * max(1.0, float(t.detach().abs().max()))  # noqa: E731  (the naive E[x^2] - E[x]^2 variance)
        assert float((y.detach().double() - yd.detach()).abs().max()) <= tol(yd)
        assert float((x.grad.double() - xd.grad).abs().max()) <= tol(xd.grad)
        assert float((bn.weight.grad.double() - ref.weight.grad).abs().max()) <= 2e-4 * max(1.0, float(ref.weight.grad.abs().max()))
        assert float((bn.bias.grad.double() - ref.bias.grad).abs().max()) <= 2e-4 * max(1.0, float(ref.bias.grad.abs().max()))
        assert float((bn.running_mean.double() - ref.running_mean).abs().max()) <= 1e-6
    finally:
        if own:
            dist.destroy_process_group()


@pytest.mark.parametrize("cin,cout,bias", [(3, 16, False), (16, 32, False), (32, 133, False), (128, 131, True), (10, 128, True)])
def test_point_linear_thin_layers_gradients(ops, device, cin, cout, bias):
    """The K10 weight-gradient route with padded channel counts + the K23 bias gradient == autograd of nn.Linear."""
    from fullysparsefusion_amd.mmdet3d_plugin.ops.sst_ops import PointLinear

    torch.manual_seed(cin * cout)
    n = 40003
    lin = PointLinear(cin, cout, bias=bias).to(device)
    x = torch.randn(n, cin, device=device, requires_grad=True)
    go = torch.randn(n, cout, device=device)
    lin(x).backward(go)
    want_w = go.double().t() @ x.detach().double()
    assert float((lin.weight.grad.double() - want_w).abs().max()) <= 2e-5 * float(want_w.abs().max())
    assert float((x.grad.double() - go.double() @ lin.weight.detach().double()).abs().max()) <= 1e-4
    if bias:
        want_b = go.double().sum(0)
        assert float((lin.bias.grad.double() - want_b).abs().max()) <= 2e-5 * float(want_b.abs().max())


@pytest.mark.parametrize("n,g,cl,cr,c,norm,act", [(50000, 700, 128, 128, 128, "ln", "gelu"), (30011, 9000, 64, 64, 64, "affine", "relu"),
                                                  (4099, 1, 128, 128, 128, "ln", "gelu"), (20000, 20000, 64, 32, 128, "none", "none")])
def test_linear_norm_act_grouped_equals_linear_of_concat(ops, device, n, g, cl, cr, c, norm, act):
    """fsf_linear_norm_act_grouped: x W_left^T + (groups W_right^T)[inv] == Linear(cat([x, groups[inv]], 1)) (float64
    yardstick), and the plugin's GroupedConcat path == the materialised concat through the same layer."""
    import torch.nn.functional as F

    from fullysparsefusion_amd.mmdet3d_plugin.ops import sst_ops

    torch.manual_seed(n + g)
    x = torch.randn(n, cl, device=device) * 2
    grp = torch.randn(g, cr, device=device) * 2
    inv = torch.randint(0, g, (n,), device=device)
    inv[:min(n, g)] = torch.arange(min(n, g), device=device)
    w = torch.randn(c, cl + cr, device=device) / (cl + cr) ** 0.5
    gam, bet = torch.rand(c, device=device) + 0.5, torch.randn(c, device=device) * 0.1
    planes = ops.linear_prepare_weight(w[:, :cl].contiguous())
    table = F.linear(grp, w[:, cl:].contiguous())
    out = ops.linear_norm_act(x, planes, c, norm=norm, gamma=gam if norm != "none" else None,
                              beta=bet if norm != "none" else None, eps=1e-3, act=act, row_add=table, row_add_index=inv)

    def tail(y):
        if norm == "ln":
            y = F.layer_norm(y, (c,), gam.to(y.dtype), bet.to(y.dtype), 1e-3)
        elif norm == "affine":
            y = y * gam.to(y.dtype) + bet.to(y.dtype)
        return F.gelu(y) if act == "gelu" else F.relu(y) if act == "relu" else y

    cat = torch.cat([x, grp[inv]], 1)
    want = tail(F.linear(cat.double(), w.double()))
    ref32 = tail(F.linear(cat, w))
    scale = max(1.0, float(want.abs().max()))
    err, err32 = float((out.double() - want).abs().max()), float((ref32.double() - want).abs().max())
    assert err <= max(3.0 * err32, 3e-6 * scale), (err, err32)
    if norm != "none":
        lin = torch.nn.Linear(cl + cr, c, bias=False).to(device)
        with torch.no_grad():
            lin.weight.copy_(w)
        if norm == "ln":
            nm = torch.nn.LayerNorm(c, eps=1e-3).to(device)
            with torch.no_grad():
                nm.weight.copy_(gam), nm.bias.copy_(bet)
        else:
            nm = torch.nn.BatchNorm1d(c, eps=1e-3).to(device).eval()
            with torch.no_grad():
                nm.weight.copy_(gam), nm.bias.copy_(bet), nm.running_var.uniform_(0.5, 2.0), nm.running_mean.normal_()
        a = torch.nn.GELU() if act == "gelu" else torch.nn.ReLU()
        with torch.no_grad():
            gc = sst_ops.GroupedConcat(x, grp, inv)
            got = sst_ops._grouped_linear_norm_act(lin, nm, a, gc)
            ref = sst_ops.linear_norm_act(lin, nm, a, gc.materialize())
        assert got is not None
        assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("n,k,ns,sc,off,norm,act", [(10641, 1024, 5, 128, 0, "ln", "gelu"), (3001, 128, 5, 128, 128, "ln", "gelu"),
                                                    (777, 128, 5, 16, 128, "none", "none"), (5000, 96, 3, 64, 96, "affine", "relu"),
                                                    (1, 64, 2, 32, 0, "ln", "relu"), (260, 70, 4, 20, 72, "ln", "gelu")])
def test_linear_norm_act_sliced_equals_separate_calls(ops, device, n, k, ns, sc, off, norm, act):
    """fsf_linear_norm_act_sliced: nslice independent layers in one launch == the same layers one fsf_linear_norm_act call at a
    time (bit for bit: per slice the arithmetic is the same), and both == float64 to fp32 accuracy."""
    import torch.nn.functional as F

    torch.manual_seed(n + k)
    width = (ns - 1) * off + k
    x = torch.randn(n, (width + 3) // 4 * 4, device=device) * 2
    w = torch.randn(ns * sc, k, device=device) / k ** 0.5
    bias = torch.randn(ns * sc, device=device) * 0.1
    gam, bet = torch.rand(ns * sc, device=device) + 0.5, torch.randn(ns * sc, device=device) * 0.1
    kw = dict(norm=norm, gamma=gam if norm != "none" else None, beta=bet if norm != "none" else None, eps=1e-3, act=act)
    planes = ops.linear_prepare_weight_sliced(w, ns, sc)
    out = ops.linear_norm_act_sliced(x, k, off, planes, ns, sc, bias=bias, **kw)
    assert out.shape == (n, ns * sc)
    for s in range(ns):
        sl = slice(s * sc, (s + 1) * sc)
        xs = x[:, s * off:s * off + k]
        if (s * off) % 4 == 0 and (n == 1 or True):
            one = ops.linear_norm_act(xs, ops.linear_prepare_weight(w[sl].contiguous(), fmt="bf16x6"), sc, bias=bias[sl].contiguous(), norm=norm,
                                      gamma=gam[sl].contiguous() if norm != "none" else None,
                                      beta=bet[sl].contiguous() if norm != "none" else None, eps=1e-3, act=act)
            assert torch.equal(out[:, sl], one), s
        y = F.linear(xs.double(), w[sl].double(), bias[sl].double())
        y32 = F.linear(xs, w[sl], bias[sl])
        def tail(t, g=gam[sl], b=bet[sl]):
            if norm == "ln":
                t = F.layer_norm(t, (sc,), g.to(t.dtype), b.to(t.dtype), 1e-3)
            elif norm == "affine":
                t = t * g.to(t.dtype) + b.to(t.dtype)
            return F.gelu(t) if act == "gelu" else F.relu(t) if act == "relu" else t
        want, ref32 = tail(y), tail(y32)
        scale = max(1.0, float(want.abs().max()))
        err, err32 = float((out[:, sl].double() - want).abs().max()), float((ref32.double() - want).abs().max())
        assert err <= max(3.0 * err32, 3e-6 * scale), (s, err, err32)


def test_separate_head_branches_in_one_launch_per_layer(ops, device):
    """FSDSeparateHead at inference: the attribute MLPs run layer by layer as sliced launches; results == the per-attribute
    modules (fused blocks bit for bit; the 2..10-wide output layer to fp32 GEMM accuracy)."""
    from fullysparsefusion_amd.mmdet3d_plugin.models.dense_heads.cluster_heads import FSDSeparateHead

    torch.manual_seed(5)
    attrs = dict(center=(3, 2, 128), dim=(3, 2, 128), rot=(2, 2, 128), vel=(2, 2, 128), score=(10, 2, 128))
    head = FSDSeparateHead(1024, attrs, norm_cfg=dict(type="LN"), act="gelu").to(device).eval()
    with torch.no_grad():
        for p in head.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    x = torch.randn(10397, 1024, device=device)
    with torch.no_grad():
        got = head(x)
        assert head.__dict__["_fsf_sliced"][1] is not None
        ref = {a: getattr(head, a)(x) for a in attrs}
        hid = {a: getattr(head, a)[1](getattr(head, a)[0](x)) for a in attrs}
    for a, (d, _, _) in attrs.items():
        assert got[a].shape == (x.size(0), d)
        want = torch.nn.functional.linear(hid[a].double(), getattr(head, a)[2].weight.double(), getattr(head, a)[2].bias.double())
        scale = max(1.0, float(want.abs().max()))
        assert float((got[a].double() - want).abs().max()) <= max(3e-6 * scale, 3.0 * float((ref[a].double() - want).abs().max())), a
    # a training-mode / grad-enabled call keeps the per-attribute modules
    head.train()
    out = head(x[:64].requires_grad_(True))
    assert out["center"].requires_grad


def test_gather_rows_strided_source_and_seg_head_stack(ops, device):
    """gather_rows from a column block of a wider buffer (no contiguous copy), and VoteSegHead's two Linears as one launch."""
    from fullysparsefusion_amd.mmdet3d_plugin.models.decode_heads.segmentation_head import VoteSegHead

    torch.manual_seed(1)
    wide = torch.randn(5000, 44, device=device)
    idx = torch.randint(0, 5000, (12345,), device=device)
    for lo, hi in ((0, 11), (11, 44), (8, 40)):
        got = ops.gather_rows(wide[:, lo:hi], idx)
        assert torch.equal(got, wide[:, lo:hi][idx])
    head = VoteSegHead(131, 10, hidden_dims=[128, 128], dropout_ratio=0.0, norm_cfg=dict(type="naiveSyncBN1d"),
                       act_cfg=dict(type="ReLU")).to(device).eval()
    with torch.no_grad():
        head.voting.bias.normal_()
        x = torch.randn(30011, 131, device=device)
        logits, votes = head(x)
        assert head.__dict__.get("_fsf_stacked") is not None and logits.shape == (30011, 11) and votes.shape == (30011, 33)
        feat = head.pre_seg_conv(x)
        for got, lin in ((logits, head.conv_seg), (votes, head.voting)):
            want = torch.nn.functional.linear(feat.double(), lin.weight.double(), lin.bias.double())
            ref = torch.nn.functional.linear(feat, lin.weight, lin.bias)
            scale = max(1.0, float(want.abs().max()))
            assert float((got.double() - want).abs().max()) <= max(3e-6 * scale, 3.0 * float((ref.double() - want).abs().max()))


def test_vote_centers_keys_equals_the_torch_expressions(ops, device):
    """fsf_vote_centers_keys against the expressions it replaces (single_stage_fsd.py group_sample + cluster-voxel keys), bit for
    bit: one- and two-class groups, exact ties (both classes weighted 1/2), voxel boundaries, negative coordinates, strided rows."""
    torch.manual_seed(11)
    P, nc, bsz = 60000, 10, 2
    groups = [[0], [1, 2], [3, 4], [5], [6, 7], [8, 9]]
    logits = torch.randn(P, nc + 1, device=device)
    logits[::7, 2] = logits[::7, 1]                    # ties inside group 1
    offs = torch.randn(P, (nc + 1) * 3, device=device) * 2
    wide = torch.randn(P, 8, device=device) * 30
    points = wide[:, 1:6]                              # a row-strided view
    points[:100, :3] = torch.tensor([-3.2, 0.0, 1.6], device=device)  # exact multiples of the voxel sizes
    batch = torch.randint(0, bsz, (P,), device=device)
    n = 200000
    p_ids = torch.randint(0, P, (n,), device=device)
    g_ids = torch.randint(0, len(groups), (n,), device=device).sort()[0]
    vs_rows = [[0.4, 0.4, 0.8], [0.8, 0.8, 1.6], [1.6, 1.6, 3.2], [0.2, 0.2, 0.4], [0.4, 0.4, 0.4], [1.0, 1.0, 6.0]]
    rmin = [-54.0, -54.0, -5.0]
    masks = [sum(1 << c for c in cols) for cols in groups]
    centers, keys, b = ops.vote_centers_keys(logits, offs, points, batch, g_ids, p_ids, nc, masks, vs_rows, rmin, bsz)
    member = torch.zeros((len(groups), nc), dtype=torch.bool, device=device)
    for gi, cols in enumerate(groups):
        member[gi, cols] = True
    logit = logits.index_select(0, p_ids)[:, :nc]
    mem = member.index_select(0, g_ids)
    masked = torch.where(mem, logit, logit.new_full((), float("-inf")))
    w = ((masked - masked.max(1)[0][:, None]).abs() < 1e-6) & mem
    assert int((w.sum(1) == 2).sum()) > 100            # the ties are really there
    w = w.float()
    w = w / w.sum(1)[:, None]
    offset = offs.reshape(-1, nc + 1, 3).index_select(0, p_ids)[:, :nc, :]
    want_c = points[:, :3].index_select(0, p_ids) + (offset * w[:, :, None]).sum(dim=1)
    vsize = torch.tensor(vs_rows, device=device)
    want_v = torch.div(want_c - torch.tensor(rmin, device=device)[None, :], vsize.index_select(0, g_ids), rounding_mode="floor").long()
    want_b = batch.index_select(0, p_ids)
    assert torch.equal(centers, want_c)
    assert torch.equal(keys[:, 1:], want_v) and torch.equal(keys[:, 0], g_ids * bsz + want_b) and torch.equal(b, want_b)


def test_round2_entry_points_on_empty_and_tiny_inputs(ops, device):
    """No rows / one row through the entry points added this round: nothing is launched on empty inputs, shapes stay right."""
    z128 = torch.empty((0, 128), device=device)
    w = torch.randn(256, 128, device=device)
    pl = ops.linear_prepare_weight_sliced(w, 2, 128)
    assert ops.linear_norm_act_sliced(z128, 128, 0, pl, 2, 128).shape == (0, 256)
    one = ops.linear_norm_act_sliced(torch.randn(1, 128, device=device), 128, 0, pl, 2, 128)
    assert one.shape == (1, 256) and bool(torch.isfinite(one).all())
    # short-segment reduce: no rows, segments all empty
    plan = ops.segment_plan_from_inverse(torch.empty((0,), dtype=torch.int64, device=device), 3)
    outs = ops.segment_reduce_short([torch.empty((0, 5), device=device), torch.empty((0, 8), device=device)], plan, "mean")
    assert [tuple(o.shape) for o in outs] == [(3, 5), (3, 8)] and not any(bool(o.any()) for o in outs)
    # gather from a strided source with no indices / vote centres with no pairs
    wide = torch.randn(10, 12, device=device)
    assert ops.gather_rows(wide[:, 2:9], torch.empty((0,), dtype=torch.int64, device=device)).shape == (0, 7)
    e = torch.empty((0,), dtype=torch.int64, device=device)
    c, k, b = ops.vote_centers_keys(torch.randn(4, 11, device=device), torch.randn(4, 33, device=device), torch.randn(4, 5, device=device),
                                    torch.zeros(4, dtype=torch.int64, device=device), e, e, 10, [1, 6], [[0.4, 0.4, 0.8]] * 2, [-54, -54, -5], 1)
    assert c.shape == (0, 3) and k.shape == (0, 4) and b.shape == (0,)
    # pooling through the cell-binned path with a single point / a single RoI, and with none inside
    rois = torch.tensor([[0.0, 0.0, -1.0, 2.0, 4.0, 2.0, 0.3]], device=device)
    p1 = torch.tensor([[0.1, 0.2, 0.0]], device=device)
    gp, gr, gf = ops.dynamic_point_pool(rois, p1, [0.5, 0.5, 0.5], 512)
    assert gp.tolist() == [0] and gr.tolist() == [0] and gf.shape == (1, 13)
    gp, gr, gf = ops.dynamic_point_pool(rois, p1 + 100.0, [0.5, 0.5, 0.5], 512)
    assert gp.numel() == 0 and gf.shape == (0, 13)
    # windowed multi-class NMS with nothing above the threshold
    boxes = torch.tensor([[0.0, 0.0, 1.0, 1.0, 0.0], [5.0, 5.0, 6.0, 6.0, 0.0]], device=device)
    rank = torch.full((2, 2), -1, dtype=torch.int32, device=device)
    keep, num, flag = ops.nms_bev_multiclass(boxes, rank, torch.zeros(2, dtype=torch.int32, device=device), 0.25, True, max_keep=10, windowed=True)
    assert num.tolist() == [0, 0] and int(flag) == 0


@pytest.mark.parametrize("n,act", [(50003, "gelu"), (17, "relu")])
def test_sir_input_gathered_parts_equal_materialised_rows(ops, device, n, act):
    """fsf_sir_input_gather: three feature tensors side by side, rows through an index == the same kernel on the gathered and
    concatenated [n, 11 + 33 + 131] matrix, bit for bit (strided part views included)."""
    torch.manual_seed(n)
    P = 30000
    both = torch.randn(P, 44, device=device)
    logits, votes = both[:, :11], both[:, 11:]
    wide = torch.randn(P, 132, device=device)
    feats = wide[:, :131]
    idx = torch.randint(0, P, (n,), device=device)
    points = torch.randn(n, 5, device=device)
    fcl = torch.randn(n, 3, device=device)
    c = 5 + 175
    dims = [3, 16, 32, c]
    layers = []
    for i in range(3):
        layers.append((torch.randn(dims[i + 1], dims[i], device=device) / dims[i] ** 0.5,
                       torch.rand(dims[i + 1], device=device) + 0.5, torch.randn(dims[i + 1], device=device) * 0.1))
    norm = [20.0, 20.0, 4.0]
    mat = torch.cat([logits[idx], votes[idx], feats[idx]], 1).contiguous()
    want = ops.sir_input(points, mat, fcl, norm, (*layers, 1e-3), act, 10.0)
    got = ops.sir_input(points, [logits, votes, feats], fcl, norm, (*layers, 1e-3), act, 10.0, feats_index=idx)
    assert torch.equal(got, want)
    got2 = ops.sir_input(points, [mat[:, :11].contiguous(), mat[:, 11:]], fcl, norm, (*layers, 1e-3), act, 10.0)  # parts, no index
    assert torch.equal(got2, want)


def test_column_stats_and_batch_norm_edge_cases(ops, device):
    """K23 with no rows: sums / statistics / gradients are zeros, nothing is launched on empty inputs; a single row."""
    x0 = torch.empty(0, 12, device=device)
    assert torch.equal(ops.column_sum(x0), torch.zeros(12, device=device))
    m, v = ops.column_mean_var(x0)
    assert torch.equal(m, torch.zeros(12, device=device)) and torch.equal(v, torch.zeros(12, device=device))
    one = torch.ones(12, device=device)
    gx, dg, db = ops.batch_norm_act_backward(x0, x0, one, one, one, one, True)
    assert gx.shape == (0, 12) and not dg.any() and not db.any()
    assert ops.batch_norm_act_forward(x0, one, one, True).shape == (0, 12)
    x1 = torch.randn(1, 12, device=device)
    m, v = ops.column_mean_var(x1)
    assert torch.equal(m, x1[0]) and not v.any()
    assert torch.equal(ops.column_sum(x1), x1[0])


@pytest.mark.parametrize("n", [1024, 4000, 12000])
def test_nms_bev_binned_pairs_equal_all_pairs(ops, device, n):
    """n >= 1024 builds the pair bits through the BEV cell grid.  Axis-aligned IoU is a handful of fp32 operations that
    torch reproduces bit for bit, so the greedy result from torch's all-pairs matrix must be matched exactly — with
    clustered boxes, boxes too large for a cell (the brute-force list), centres far outside the grid (clamped border
    cells), exact duplicates and NaN boxes."""
    rng = np.random.default_rng(n)
    nclu = n // 8
    ctr = rng.uniform(-50, 50, (nclu, 2))[rng.integers(0, nclu, n)] + rng.normal(0, 0.8, (n, 2))
    wl = np.stack([rng.uniform(0.5, 2.5, n), rng.uniform(0.5, 5.5, n)], 1)
    big = rng.choice(n, n // 50, replace=False)
    wl[big] *= rng.uniform(3, 15, (big.size, 1))                    # r up to ~45 m: the big list
    far = rng.choice(n, n // 40, replace=False)
    ctr[far] += rng.choice([-1, 1], (far.size, 2)) * rng.uniform(900, 3000, (far.size, 2))  # clamped border cells
    dup = rng.choice(n, n // 60, replace=False)
    ctr[dup], wl[dup] = ctr[(dup + 1) % n], wl[(dup + 1) % n]
    boxes = np.concatenate([ctr - wl / 2, ctr + wl / 2, np.zeros((n, 1))], 1).astype(np.float32)
    boxes[rng.choice(n, 5, replace=False)] = np.nan
    b = torch.from_numpy(boxes).to(device)
    thresh = 0.3
    keep = ops.nms_bev(b, thresh, False).cpu().numpy()
    # all-pairs reference in torch fp32, the kernel's own formula (rect_overlap_normal / iou_bev)
    # (fmaxf / fminf return the non-NaN operand: torch.fmax / fmin, not maximum / minimum)
    zero, tiny = b.new_zeros(()), b.new_full((), 1e-8)
    l = torch.fmax(b[:, None, 0], b[None, :, 0]); r = torch.fmin(b[:, None, 2], b[None, :, 2])
    t = torch.fmax(b[:, None, 1], b[None, :, 1]); d = torch.fmin(b[:, None, 3], b[None, :, 3])
    ov = torch.fmax(r - l, zero) * torch.fmax(d - t, zero)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    hit = (ov / torch.fmax(area[:, None] + area[None, :] - ov, tiny) > thresh).cpu().numpy()
    alive, want = np.ones(n, bool), []
    for i in range(n):
        if alive[i]:
            want.append(i)
            alive[i + 1:] &= ~hit[i, i + 1:]
    np.testing.assert_array_equal(keep, np.asarray(want))
    assert np.array_equal(keep, ops.nms_bev(b, thresh, False).cpu().numpy())


@pytest.mark.parametrize("n,m,k,c,norm,act,grouped,long_seg", [
    (300000, 9000, 180, 128, "ln", "gelu", False, 120000), (300000, 9000, 128, 128, "ln", "gelu", True, 120000),
    (100000, 250, 136, 128, "ln", "gelu", False, 30000), (70001, 70001, 64, 64, "ln", "relu", True, 0),
    (50000, 3000, 128, 36, "ln", "gelu", False, 0), (129, 5, 128, 128, "ln", "gelu", True, 0), (17, 17, 40, 128, "ln", "relu", False, 0),
    (1, 1, 128, 128, "ln", "gelu", False, 0), (200000, 1, 128, 128, "ln", "gelu", True, 0)])
def test_linear_norm_act_segmax_equals_the_two_pass_form(ops, device, n, m, k, c, norm, act, grouped, long_seg):
    """K22s (`fsf_linear_norm_act_segmax`): rows sorted by segment -> the rows of fsf_linear_norm_act[_grouped] bit for bit, and the
    segment maxima of fsf_segment_reduce(mode max) over them bit for bit (max is exact, whatever the combination order: closed runs
    inside a 16-row group, the LDS slots of a 128-row block, the carry across a workgroup's blocks, the atomic max across
    workgroups); segment lengths from 1 to > 1e5 rows (a segment that spans many workgroups), one segment only, n = 1; output into a
    column slice of a wider buffer; `want_rows=False` writes no rows."""
    torch.manual_seed(n + m + k)
    if m == n:
        ids = torch.arange(n, device=device)
    else:
        ids = torch.randint(0, m, (n,), device=device)
        if long_seg:
            ids[:long_seg] = min(17, m - 1)
        ids[:m] = torch.maximum(ids[:m], torch.zeros_like(ids[:m]))
        ids = torch.cat([torch.arange(m, device=device), ids])[:n] if n >= m else ids  # every id occurs (unique -> no empty segment)
    ids = torch.sort(ids)[0]
    uniq, inv = torch.unique(ids, return_inverse=True)
    m_eff = uniq.numel()
    plan = ops.segment_plan_from_inverse(inv, m_eff)
    assert torch.equal(plan.order.long(), torch.arange(n, device=device))  # sorted input: the plan's order is the identity
    x = torch.randn(n, k, device=device) * torch.exp(torch.randn(n, 1, device=device))
    w = torch.randn(c, k, device=device) / k ** 0.5
    gam, bet = torch.rand(c, device=device) + 0.5, torch.randn(c, device=device) * 0.1
    planes = ops.linear_prepare_weight(w)
    kw = dict(norm=norm, gamma=gam, beta=bet, eps=1e-3, act=act)
    if grouped:
        table = torch.randn(m_eff, c, device=device)
        kw.update(row_add=table, row_add_index=inv)
    rows = ops.linear_norm_act(x, planes, c, **kw)
    want = ops.segment_reduce(rows, plan, "max")
    wide = torch.full((m_eff, 2 * c + 4), float("-inf"), device=device)
    seg_out = wide[:, c:2 * c]
    got_rows = ops.linear_norm_act_segmax(x, planes, c, inv, seg_out, **kw)
    assert torch.equal(got_rows, rows)
    assert torch.equal(seg_out, want)
    assert bool(torch.isinf(wide[:, :c]).all()) and bool(torch.isinf(wide[:, 2 * c:]).all())  # nothing written beside the slice
    seg2 = torch.full((m_eff, c), float("-inf"), device=device)
    assert ops.linear_norm_act_segmax(x, planes, c, inv, seg2, want_rows=False, **kw) is None
    assert torch.equal(seg2, want)
    # an independent yardstick for the maxima (torch's scatter-reduce over the same rows)
    ref = torch.full((m_eff, c), float("-inf"), device=device).scatter_reduce(0, inv[:, None].expand(n, c), rows, "amax", include_self=True)
    assert torch.equal(want, ref)


@pytest.mark.parametrize("n,m_target,ng,bsz,minp,empty_group", [(200000, 30000, 6, 1, 2, None), (50000, 49000, 6, 2, 2, 3), (1000, 10, 3, 1, 5, 1),
                                                               (5, 5, 6, 1, 2, None), (300000, 2000, 6, 1, 2, None)])
def test_cluster_key_survival_equals_the_torch_expression(ops, device, n, m_target, ng, bsz, minp, empty_group):
    """K25 against the plugin's own ATen chain (itself exact against the oracle's per-group loop in the full-size tests): surviving keys /
    pairs in ascending order and the pair -> surviving-key map; a group none of whose keys is dense enough keeps all of them
    (single_stage_fsd.py:953-954)."""
    torch.manual_seed(n + m_target)
    g = torch.randint(0, ng * bsz, (n,), device=device)
    vox = torch.randint(0, max(1, m_target // (ng * bsz)), (n, 1), device=device)
    keys = torch.cat([g[:, None], vox, vox % 7, vox % 3], 1)
    if empty_group is not None:  # every key of this group is a singleton: none reaches min_points -> the whole group survives
        sel = (g // bsz) == empty_group
        keys[sel, 1] = torch.arange(int(sel.sum()), device=device) + 10 ** 6
    new_keys, plan = ops.unique_rows(keys)
    inv, cnt = plan.inv, plan.cnt
    k_idx, k_group, v_idx, vox_inv = ops.cluster_key_survival(new_keys, cnt, inv, bsz, minp, ng)
    key_ok = cnt >= minp
    key_group = torch.div(new_keys[:, 0], bsz, rounding_mode="floor")
    has_valid = torch.zeros(ng, dtype=torch.bool, device=device)
    has_valid[key_group[key_ok]] = True
    key_keep = key_ok | ~has_valid[key_group]
    want_k = key_keep.nonzero().squeeze(1)
    want_v = key_keep[inv].nonzero().squeeze(1)
    remap = key_keep.long().cumsum(0) - 1
    assert torch.equal(k_idx, want_k) and torch.equal(v_idx, want_v) and torch.equal(vox_inv, remap[inv[want_v]])
    assert torch.equal(k_group.long(), key_group[want_k])
    # fsf_cluster_point_ids: labels numbered over all voxels -> from 0 inside each group, mapped to the surviving pairs
    labels = torch.cumsum(torch.randint(0, 2, (want_k.numel(),), device=device), 0).int()  # nondecreasing, like first-member numbering
    gp = (torch.arange(want_v.numel(), device=device) % ng).long()
    bp = (torch.arange(want_v.numel(), device=device) % bsz).long()
    got = ops.cluster_point_ids(labels, k_group, vox_inv, gp, bp, ng)
    vg = key_group[want_k]
    first = torch.searchsorted(vg, torch.arange(ng, device=device))
    base = labels.long()[first.clamp(max=max(labels.numel() - 1, 0))]
    want = torch.stack([gp, bp, (labels.long() - base[vg])[vox_inv]], 1)
    assert torch.equal(got, want)
    if empty_group is not None:
        assert bool(key_keep[key_group == empty_group].all()) and int((key_group == empty_group).sum()) > 0


@pytest.mark.parametrize("kind,cin,cout", [("subm", 128, 128), ("subm", 256, 128), ("strided", 128, 256), ("inverse", 256, 128)])
def test_spconv_backward_weight_on_strided_and_inverse_tables_equals_float64(ops, device, kind, cin, cout):
    """K10p against float64 on submanifold / strided / inverse tables (offsets without pairs, more than one channel tile)."""
    rng = np.random.default_rng(cin + cout)
    shape = (16, 200, 200)
    idx = torch.from_numpy(surface_sites(rng, 1, shape, 30011)).to(device)
    if kind == "subm":
        table, m_in, m_out = ops.rulebook_subm(idx, 1, shape), idx.size(0), idx.size(0)
    else:
        _, nbr, nbr_inv, _ = ops.rulebook_strided(idx, 1, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        if kind == "strided":
            table, m_in, m_out = nbr, idx.size(0), nbr.size(0)
        else:  # the inverse convolution on the same key: fine rows are the outputs
            table, m_in, m_out = nbr_inv, nbr.size(0), idx.size(0)
    ip, num = ops.rulebook_to_pairs(table)
    feat = torch.from_numpy(rng.standard_normal((m_in, cin)).astype(np.float32)).to(device)
    gout = torch.from_numpy(rng.standard_normal((m_out, cout)).astype(np.float32)).to(device)
    got = ops.spconv_backward_weight(feat, gout, ip, num)
    assert torch.equal(got, ops.spconv_backward_weight(feat, gout, ip, num))  # deterministic
    tb = table.long()
    for k in range(table.size(1)):
        sel = (tb[:, k] >= 0).nonzero().squeeze(1)
        want = feat[tb[sel, k]].double().t() @ gout[sel].double()
        scale = max(1.0, float(want.abs().max()))
        assert float((got[k].double() - want).abs().max()) <= 1e-5 * scale, (kind, k)


# ------------------------------------------------------------------------------------------------ K22h (f16 x 3 planes)
def _decode_row_planes(rp):
    """RowPlanes -> float64 [n, c]: (hi + lo) / s_row."""
    n, c = rp.n, rp.c
    raw = rp.data[:n * c * 4].view(torch.float16).view(n, c // 8, 2, 8).double()
    return (raw[:, :, 0, :] + raw[:, :, 1, :]).reshape(n, c) * rp.inv_scales[:n, None].double()


@pytest.mark.parametrize("n,c,norm,act", [(1000, 1024, "none", "none"), (2049, 768, "ln", "gelu"), (37, 128, "ln", "relu"),
                                          (1, 8, "none", "gelu"), (513, 2048, "ln", "gelu"), (300, 896, "none", "none")])
def test_rows_to_planes_split_and_norm(ops, device, n, c, norm, act):
    """fsf_rows_to_planes: hi + lo reproduces x s_row to 2^-22 of every element that matters at the row's scale (rows spanning 12
    orders of magnitude, an all-zero row), the fp32 rows it can emit equal act(LayerNorm(x)) within fp32 rounding, strided input."""
    import torch.nn.functional as F

    torch.manual_seed(n + c)
    buf = torch.randn(n, c + 8, device=device) * torch.exp(torch.randn(n, 1, device=device) * 4.0)
    x = buf[:, :c]
    if n > 2:
        x[1] = 0.0
    g, b = torch.rand(c, device=device) + 0.5, torch.randn(c, device=device) * 0.1
    rp, rows = ops.rows_to_planes(x, norm, g if norm == "ln" else None, b if norm == "ln" else None, 1e-3, act, want_rows=True)
    y = x.double()
    if norm == "ln":
        y = F.layer_norm(y, (c,), g.double(), b.double(), 1e-3)
    y = F.gelu(y) if act == "gelu" else F.relu(y) if act == "relu" else y
    scale = y.abs().amax(1, keepdim=True).clamp_min(1e-30)
    assert float(((rows.double() - y).abs() / scale).max()) <= 4e-6
    dec = _decode_row_planes(rp)
    err = (dec - rows.double()).abs()
    # |x s - hi - lo| <= max(2^-22 |x s|, 2^-25): relative to the element, or 2^-38 of the row maximum (which sits at 2^13 .. 2^14)
    bound = torch.maximum(rows.double().abs() * 2.0 ** -21.9, rows.double().abs().amax(1, keepdim=True) * 2.0 ** -37.9)
    assert bool((err <= bound).all()), float((err / bound.clamp_min(1e-300)).max())
    rp2 = ops.rows_to_planes(x, norm, g if norm == "ln" else None, b if norm == "ln" else None, 1e-3, act)
    assert torch.equal(rp2.data, rp.data) and torch.equal(rp2.inv_scales, rp.inv_scales)


@pytest.mark.parametrize("n,k,c,slice_c,norm,act,bias", [
    (10641, 1024, 1024, 128, "none", "none", True),    # shared_mlp / out_proj of the query heads at the frame's query count
    (10397, 768, 1024, 128, "none", "none", False),   # combine_fsd_feat_mlp / bbox_head.shared_mlp[0]
    (3000, 896, 1024, 128, "none", "none", True),
    (10641, 1024, 640, 128, "ln", "gelu", True),      # first layer of FSDSeparateHead's five branches
    (1500, 256, 256, 128, "none", "gelu", True),
    (129, 512, 272, 68, "ln", "relu", False),         # four stacked 68-wide layers, a partial row block
    (1, 32, 128, 128, "affine", "none", True),
])
def test_linear_planes_f16x3_vs_float64(ops, device, n, k, c, slice_c, norm, act, bias):
    """K22h: act(norm(x W^T + b)) with both operands as f16 hi | lo planes (three MFMA passes) against float64: <= 1e-5 of the output
    scale (what the sparse-conv plane kernels are held to), no farther from float64 than a few fp32 GEMM errors on well-scaled data,
    rows of very different magnitude in one launch, deterministic."""
    import torch.nn.functional as F

    torch.manual_seed(n + k + c)
    x = torch.randn(n, k, device=device) * torch.exp(torch.randn(n, 1, device=device) * 2.0)
    w = torch.randn(c, k, device=device) / k ** 0.5
    w[::7] *= 1e-3  # output channels with weights far below the layer's maximum
    b = torch.randn(c, device=device) if bias else None
    g, be = torch.rand(c, device=device) + 0.5, torch.randn(c, device=device) * 0.1
    assert ops.linear_planes_supported(k, c, slice_c)
    wp = ops.linear_prepare_weight_f16(w, slice_c)
    xp = ops.rows_to_planes(x)
    kw = dict(bias=b, norm=norm, gamma=g if norm != "none" else None, beta=be if norm != "none" else None, eps=1e-3, act=act)
    out = ops.linear_planes_norm_act(xp, wp, c, slice_c, **kw)

    def tail(y):
        if norm == "ln":
            parts = [F.layer_norm(y[:, s:s + slice_c], (min(slice_c, c - s),), g[s:s + slice_c].to(y.dtype), be[s:s + slice_c].to(y.dtype), 1e-3)
                     for s in range(0, c, slice_c)]
            y = torch.cat(parts, 1)
        elif norm == "affine":
            y = y * g.to(y.dtype) + be.to(y.dtype)
        return F.gelu(y) if act == "gelu" else F.relu(y) if act == "relu" else y

    pre = F.linear(x.double(), w.double(), b.double() if bias else None)
    want = tail(pre)
    # per ROW: rows differ by orders of magnitude, each is judged at its own output scale
    row_scale = (pre.abs().amax(1, keepdim=True) if norm != "ln" else want.abs().amax(1, keepdim=True)).clamp_min(1e-30)
    err = float(((out.double() - want).abs() / row_scale).max())
    ref32 = tail(F.linear(x, w, b))
    err32 = float(((ref32.double() - want).abs() / row_scale).max())
    assert err <= 1e-5, (err, err32)
    assert err <= max(8.0 * err32, 2e-6), (err, err32)
    assert torch.equal(out, ops.linear_planes_norm_act(xp, wp, c, slice_c, **kw))


def test_wide_mlp_chain_on_planes_equals_the_k22_path(ops, device, monkeypatch):
    """`build_mlp(768, [1024, 1024])` + a five-branch FSDSeparateHead at the frame's query count: the K22h route (planes handed from
    layer to layer, the LayerNorm + GELU pass writing planes) against float64 and against the K22 route (FSF_K22H=0)."""
    import torch.nn.functional as F

    from fullysparsefusion_amd import switches
    from fullysparsefusion_amd.mmdet3d_plugin.models.dense_heads.cluster_heads import FSDSeparateHead
    from fullysparsefusion_amd.mmdet3d_plugin.ops import sst_ops

    torch.manual_seed(5)
    mlp = sst_ops.build_mlp(768, [1024, 1024], dict(type="LN", eps=1e-3), act="gelu").to(device).eval()
    head = FSDSeparateHead(1024, dict(center=(3, 2, 128), dim=(3, 2, 128), rot=(2, 2, 128), vel=(2, 2, 128), score=(10, 2, 128)),
                           norm_cfg=dict(type="LN"), act="gelu").to(device).eval()
    x = torch.randn(10397, 768, device=device).clamp_min(0) * 3  # (max-pooled GELU features are non-negative)
    with torch.no_grad():
        assert switches.K22H and head.accepts_planes(x.size(0))
        mid = mlp(x, planes_out=True)
        assert isinstance(mid, sst_ops.hip_ops.RowPlanes) and mid.c == 1024
        got = head(mid)
        got_rows = mlp(x)                       # the same chain ending in fp32 rows
        monkeypatch.setattr(switches, "K22H", False)
        head.__dict__.pop("_fsf_sliced", None)
        ref_rows = mlp(x)
        ref = head(ref_rows)
        # float64
        y = x.double()
        for blk in mlp:
            y = F.gelu(F.layer_norm(F.linear(y, blk[0].weight.double()), (1024,), blk[1].weight.double(), blk[1].bias.double(), blk[1].eps))
    scale = float(y.abs().max())
    assert float((got_rows.double() - y).abs().max()) <= 1e-5 * scale
    assert float((ref_rows.double() - y).abs().max()) <= 1e-5 * scale
    for name in head.attrs:
        a, b = got[name].double(), ref[name].double()
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), name


@pytest.mark.parametrize("m,cin,cout", [(3000, 64, 128), (3000, 128, 256), (1517, 512, 512), (1517, 1024, 512), (7441, 512, 256),
                                        (20000, 256, 128), (130, 32, 68)])
def test_spconv_forward_split_planes_f16x3_vs_float64(ops, device, m, cin, cout):
    """K9b-XP (`fsf_spconv_forward_split_planes`: K9b with both operands as f16 hi | lo planes, per-ROW input scales, the accumulators
    moved between the rows' units by power-of-two ratios) against float64 on sampled rows — <= 1e-5 of the output scale with rows
    spanning e^+-1 in magnitude and weights down to 1e-3 of the layer maximum —, against K9b (bf16 x 6), with the fused epilogue,
    over the k-split + XCD-aware layout of the deep levels (1517 rows) and the unsplit one (20 000 rows); deterministic."""
    rng = np.random.default_rng(m + cin + cout)
    shape = (16, 48, 48) if m <= 8000 else (40, 512, 512)
    idx = surface_sites(rng, 2 if m <= 8000 else 1, shape, m)
    n = idx.shape[0]
    feat = (rng.standard_normal((n, cin)) * np.exp(rng.standard_normal((n, 1)))).astype(np.float32)
    feat[3 % n] = 0.0  # an all-zero row (scale of an empty row)
    #  rows whose maximum is tiny but not zero among ordinary neighbours: their unit (2^113 / max) used to carry the
    # accumulators of the rows around them through a factor of ~2^100 — inf; now the unit is capped at 2^60
    feat[7 % n] *= 1e-30
    feat[(n // 2) % n] *= 1e-36
    feat[(n // 3) % n] = 0.0
    feat[(n // 3) % n, 0] = 1e-38
    w = (rng.standard_normal((27, cin, cout)) / np.sqrt(cin * 6)).astype(np.float32)
    w[:, :, ::5] *= 1e-3
    nbr = ops.rulebook_subm(torch.from_numpy(idx).to(device), 2 if m <= 8000 else 1, shape)
    f, wd = torch.from_numpy(feat).to(device), torch.from_numpy(w).to(device)
    assert ops.spconv_split_planes_supported(cin, cout)
    wp = ops.spconv_prepare_weight_split_f16(wd)
    xp = ops.rows_to_planes(f)
    out = ops.spconv_forward_split_planes(xp, wp, 27, cout, nbr)
    assert torch.isfinite(out).all()
    rows = torch.from_numpy(np.unique(np.concatenate([rng.choice(n, size=min(n, 512), replace=False), [7 % n, (n // 2) % n, (n // 3) % n]]))).to(device)
    nb = nbr.index_select(0, rows).long()
    gathered = torch.where((nb >= 0)[:, :, None], f.double()[nb.clamp(min=0)], torch.zeros((), dtype=torch.float64, device=device))
    want64 = torch.einsum("rkc,kcd->rd", gathered, wd.double())
    scale_ = max(1.0, float(want64.abs().max()))
    err = float((out.index_select(0, rows).double() - want64).abs().max())
    assert err <= 1e-5 * scale_, err / scale_
    ref = ops.spconv_forward_split(f, ops.spconv_prepare_weight_split(wd), 27, cout, nbr)
    assert float((out - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(device)
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(device)
    res = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(device)
    out2 = ops.spconv_forward_split_planes(xp, wp, 27, cout, nbr, scale=sc, shift=sh, residual=res, relu=True)
    want2 = torch.relu(out * sc + sh + res)
    assert float((out2 - want2).abs().max()) <= 1e-5 * max(1.0, float(want2.abs().max()))
    assert torch.equal(out2, ops.spconv_forward_split_planes(xp, wp, 27, cout, nbr, scale=sc, shift=sh, residual=res, relu=True))
