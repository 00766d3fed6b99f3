"""CPU: the oracle restatements against the golden vectors produced by the reference's own Python
(tests/golden/make_golden.py).  This is what pins the oracle (SURVEY.md §8 c)."""
import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden
from oracle import project as oproj
from oracle import scatter as oscatter
from oracle import voxelize as ovox


@pytest.mark.parametrize("case", sorted(golden_cases(load_golden("scatter_v2.npz"))))
def test_scatter_v2_matches_reference(case):
    g = golden_cases(load_golden("scatter_v2.npz"))[case]
    out = oscatter.scatter_v2(g["feat"], g["coors"], str(g["mode"]), min_points=int(g["min_points"]))
    np.testing.assert_array_equal(out[1].numpy(), g["new_coors"])  # lexicographic unique rows: bit-exact
    if "inv" in g:
        np.testing.assert_array_equal(out[2].numpy(), g["inv"])
    if str(g["mode"]) == "max":
        np.testing.assert_array_equal(out[0].numpy(), g["new_feat"])
    else:
        np.testing.assert_allclose(out[0].numpy(), g["new_feat"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", ["nusc_small", "nusc_mid", "av2_small"])
def test_projection_matches_reference(tag):
    g = golden_cases(load_golden("project.npz"))[tag]
    ids, p2d = oproj.points_in_mask(g["points"], g["mask"], g["lidar2img"])
    np.testing.assert_array_equal(p2d, g["pts_2d"])  # fp32 bit-exact (fma chain in k order)
    np.testing.assert_array_equal(ids, g["obj_id"])
    assert (g["obj_id"] > 0).sum() > 50  # the fixture actually hits masks
    if "score" in g:
        cam_ids, score = oproj.cam_select_score(ids, g["mask_anno"])
        np.testing.assert_array_equal(cam_ids, g["cam_ids"])
        np.testing.assert_array_equal(score, g["score"])


@pytest.mark.parametrize("tag", ["v01", "v03", "v005", "v02"])
def test_divfloor_matches_torch(tag):
    g = golden_cases(load_golden("divfloor.npz"))[tag]
    c = ovox.divfloor_coors(g["points"], g["voxel"], g["min"], order="xyz")
    np.testing.assert_array_equal(c, g["coors_xyz"])


def test_two_floor_formulas_disagree_on_boundaries():
    """floor((x-min)/v) and torch.div(...,'floor') are different functions."""
    g = golden_cases(load_golden("divfloor.npz"))["v02"]
    pts = g["points"]
    a = ovox.dynamic_voxelize(pts, g["voxel"], [-51.2, -51.2, -5, 51.2, 51.2, 3])
    b = g["coors_xyz"]
    ok = (a >= 0).all(1)
    differ = (a[ok][:, 2] != b[ok][:, 0]).sum()
    assert differ > 0


def test_dynamic_voxelize_oob_slots():
    pts = np.array([[60.0, 0, 0], [0, 60.0, 0], [0, 0, 10.0], [0, 0, 0], [-51.2, -51.2, -5.0], [51.19, 51.19, 2.99]], dtype=np.float32)
    c = ovox.dynamic_voxelize(pts, (0.2, 0.2, 0.2), [-51.2, -51.2, -5, 51.2, 51.2, 3])
    assert c[0].tolist() == [-1, 0, 0]
    assert c[1].tolist() == [-1, -1, 0]
    assert c[2].tolist() == [-1, -1, -1]
    assert c[3].tolist() == [25, 256, 256]
    assert c[4].tolist() == [0, 0, 0]
    assert c[5].tolist() == [39, 511, 511]
    assert ovox.grid_size((0.2, 0.2, 0.2), [-51.2, -51.2, -5, 51.2, 51.2, 3]) == [512, 512, 40]


def test_ingroup_rank_contract():
    g = torch.randint(0, 40, (1000,))
    r = oscatter.ingroup_rank(g)
    for v in g.unique():
        rr = r[g == v]
        assert sorted(rr.tolist()) == list(range(rr.numel()))


def test_sparse_connected_components_equal_the_dense_reference_call():
    """oracle.modules.connected_components_xy switches to a sparse adjacency beyond 4096 centres; the labels must be those
    of the reference's dense call (single_stage_fsd.py:69-82) — checked on clustered centres with chains."""
    from scipy.sparse.csgraph import connected_components

    from oracle import modules as omod

    rng = np.random.default_rng(3)
    n = 5000
    ctr = rng.uniform(-20, 20, (300, 2))[rng.integers(0, 300, n)] + rng.normal(0, 0.25, (n, 2))
    pts = torch.from_numpy(np.concatenate([ctr, rng.uniform(-1, 1, (n, 1))], 1).astype(np.float32))
    for dist in (0.2, 0.6):
        got = omod.connected_components_xy(pts, dist)
        p = pts[:, :2]
        d = ((p[:, None, :] - p[None, :, :]) ** 2).sum(2) ** 0.5
        want = connected_components((d < dist).numpy(), directed=False)[1]
        np.testing.assert_array_equal(got.numpy(), want)
        assert 10 < want.max() < n - 10


# ------------------------------------------------------------- the oracle's own fast paths against its plain ones
def test_oracle_rotated_overlap_batch_equals_scalar():
    """`rotated_overlap_batch` / the vectorised `nms_lazy` (what makes the 10 k-box-per-class oracle NMS of the 10-sweep frame
    affordable) against the scalar intersect-and-sort polygon and the all-pairs greedy NMS they restate."""
    from oracle import refine as R

    rng = np.random.default_rng(0)
    n = 300
    c, wl, yaw = rng.uniform(-8, 8, (n, 2)), rng.uniform(0.5, 5, (n, 2)), rng.uniform(-4, 4, n)
    b = np.stack([c[:, 0] - wl[:, 0] / 2, c[:, 1] - wl[:, 1] / 2, c[:, 0] + wl[:, 0] / 2, c[:, 1] + wl[:, 1] / 2, yaw], 1)
    b[1] = b[0]                       # identical boxes
    b[2, 4] = b[3, 4] = 0.0           # axis-aligned
    b[4] = b[2]
    b[4, [0, 2]] += 0.5               # shifted copy: collinear edges
    b[5] = b[2]
    b[5, [0, 2]] += b[2, 2] - b[2, 0]  # touching along an edge
    for i in range(12):
        got = R.rotated_overlap_batch(b[i], b)
        want = np.array([R.rotated_overlap(b[i], b[j]) for j in range(n)])
        np.testing.assert_allclose(got, want, atol=1e-11)
    assert R.rotated_overlap_batch(b[0], np.zeros((0, 5))).shape == (0,)
    keep, margin, close = R.nms_lazy(b, 0.2, near_tol=1e-2)
    assert np.array_equal(keep, R.nms_from_iou(R.iou_bev_matrix(b), 0.2))
    assert margin <= close[:, 2].min() + 1e-15 if len(close) else margin >= 1e-2
    keep_a, _ = R.nms_lazy(b, 0.3, rotated=False)
    assert np.array_equal(keep_a, R.nms_from_iou(R.iou_bev_matrix(b, rotated=False), 0.3))


def test_oracle_point_pool_stop_at_cap_is_the_plain_result():
    from oracle import refine as R

    rng = np.random.default_rng(1)
    pts = rng.uniform(-10, 10, (4000, 3)).astype(np.float32)
    rois = np.concatenate([rng.uniform(-8, 8, (40, 2)), rng.uniform(-3, 0, (40, 1)), rng.uniform(1, 4, (40, 3)),
                           rng.uniform(-3, 3, (40, 1))], 1).astype(np.float32)
    for cap in (50, 400, 100000):
        a = R.dynamic_point_pool(rois, pts, [1.0, 1.0, 1.0], 64, cap)
        b = R.dynamic_point_pool(rois, pts, [1.0, 1.0, 1.0], 64, cap, stop_at_cap=True)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    full = R.dynamic_point_pool(rois, pts, [1.0, 1.0, 1.0], 64, 100000, return_margin=True)[3]
    near = R.dynamic_point_pool(rois, pts, [1.0, 1.0, 1.0], 64, 100000, return_margin=True, near_tol=0.05)[3]
    np.testing.assert_array_equal(near, full[full[:, 2] < 0.05])
