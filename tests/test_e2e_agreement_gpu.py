"""End-to-end agreement of an UN-RESTARTED GPU `FSF.simple_test` with an UN-RESTARTED oracle chain (VERDICT r3 "missing" 2).

Every other full-size test restarts the oracle from device intermediates at each discontinuity so that a difference is
attributable to ONE kernel.  This one does not: input frame -> final boxes on both sides, each stage fed by its own
side's previous stage, and then asks how far the two ends sit from each other — the only proxy this project has for
"detection mAP within 0.1 of the reference" (no dataset, no checkpoint: BASELINE.md).

Reported (and written to gpurun_out/e2e_agreement_<frame>.json): boxes on each side, greedy one-to-one matches by label +
BEV IoU (float64 polygon oracle), matched at IoU >= 0.99 with |dscore| <= 1e-3, unmatched count, worst matched IoU /
score difference; identity of the integer query structure (camera-query keys, LiDAR cluster keys); 99.9th-percentile and
maximum deviation of the two SIR stacks' group features on the common keys, relative to the feature scale.

Arbitration (round 5): the camera stack's SIR group features are the one quantity whose GPU / oracle distance is far above the
per-kernel bounds (LayerNorm(eps=1e-3) rows of the position MLP on `f_cluster ~ 0` amplify the centroid's fp32 rounding).
The oracle chain therefore runs a SECOND time in float64 — same integer structure: the voxel keys and the projected mask ids
depend on the fp32 input points only — through the segmentor, the fusion, the segmentation head and the camera stack, and the
test asserts that the GPU features are no farther from that float64 chain than a small multiple of the fp32 ORACLE's own
distance to it: the deviation is fp32 conditioning of the reference arithmetic, not a property of the HIP kernels.

The thresholds asserted below were read off the measurement on MI355X ("end-to-end agreement") and
then frozen.
"""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, build_av2_fsf, build_test_fsf
from oracle import modules as omod
from oracle import refine as orefine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fsf_pair(device):
    model = build_test_fsf()
    cpu = copy.deepcopy(model)
    return model.to(device), cpu


def _xyxyr(b):
    b = np.asarray(b, dtype=np.float64)
    return np.stack([b[:, 0] - b[:, 3] / 2, b[:, 1] - b[:, 4] / 2, b[:, 0] + b[:, 3] / 2, b[:, 1] + b[:, 4] / 2, b[:, 6]], 1)


def _pair_iou(a, b):
    """Rotated BEV IoU of a[i] with b[i] (float64 polygon intersection, oracle/refine.py)."""
    a, b = _xyxyr(a), _xyxyr(b)
    out = np.zeros(a.shape[0])
    for i in range(a.shape[0]):
        ov = float(orefine.rotated_overlap_batch(a[i], b[i:i + 1])[0])
        ua = (a[i, 2] - a[i, 0]) * (a[i, 3] - a[i, 1]) + (b[i, 2] - b[i, 0]) * (b[i, 3] - b[i, 1]) - ov
        out[i] = ov / max(ua, 1e-12)
    return out


def match_boxes(gb, gs, gl, ob, os_, ol):
    """Greedy one-to-one matching in descending GPU score: a GPU box takes the nearest (BEV centre) unmatched oracle box
    of its label within 0.5 m; returns (index pairs, IoU per pair, |dscore| per pair)."""
    gb, ob = np.asarray(gb, np.float64), np.asarray(ob, np.float64)
    taken = np.zeros(ob.shape[0], dtype=bool)
    pairs = []
    for i in np.argsort(-np.asarray(gs), kind="stable"):
        cand = np.nonzero((np.asarray(ol) == gl[i]) & ~taken)[0]
        if cand.size == 0:
            continue
        d = np.hypot(ob[cand, 0] - gb[i, 0], ob[cand, 1] - gb[i, 1])
        j = cand[int(np.argmin(d))]
        if d.min() <= 0.5:
            taken[j] = True
            pairs.append((int(i), int(j)))
    pairs = np.array(pairs, dtype=np.int64).reshape(-1, 2)
    iou = _pair_iou(gb[pairs[:, 0]], ob[pairs[:, 1]]) if len(pairs) else np.zeros(0)
    ds = np.abs(np.asarray(gs, np.float64)[pairs[:, 0]] - np.asarray(os_, np.float64)[pairs[:, 1]]) if len(pairs) else np.zeros(0)
    return pairs, iou, ds


def _feature_deviation(g_feats, g_keys, o_feats, o_keys):
    """Deviation of group features on the keys both sides hold, relative to the oracle's feature scale."""
    gk = {tuple(r): i for i, r in enumerate(np.asarray(g_keys).tolist())}
    rows = [(gk[tuple(r)], j) for j, r in enumerate(np.asarray(o_keys).tolist()) if tuple(r) in gk]
    if not rows:
        return dict(common=0, p999=float("nan"), max=float("nan"), scale=float("nan"))
    rows = np.array(rows)
    g, o = np.asarray(g_feats, np.float64)[rows[:, 0]], np.asarray(o_feats, np.float64)[rows[:, 1]]
    scale = max(1.0, float(np.abs(o).max()))
    d = np.abs(g - o).max(1) / scale
    return dict(common=int(len(rows)), p999=float(np.quantile(d, 0.999)), max=float(d.max()), median=float(np.median(d)),
                scale=scale)


@pytest.fixture(scope="module")
def av2_pair(device):
    model = build_av2_fsf(perturb_image_branch=True)
    cpu = copy.deepcopy(model)
    return model.to(device), cpu


@pytest.mark.parametrize("which", ["nuscenes_1sweep", "nuscenes_10sweep", "av2_150k"])
def test_unrestarted_gpu_frame_vs_unrestarted_oracle(which, request, device, monkeypatch):
    from fullysparsefusion_amd import synthetic

    if which == "av2_150k":
        model, cpu = request.getfixturevalue("av2_pair")
        frame = synthetic.make_frame_av2(seed=0)
    else:
        model, cpu = request.getfixturevalue("fsf_pair")
        frame = synthetic.make_frame(num_sweeps=10 if which == "nuscenes_10sweep" else 1, seed=0)
    pts8, mask, anno, L = (torch.from_numpy(frame[k]) for k in ("points", "mask_data", "mask_anno", "lidar2img"))
    cap = {}
    orig = model.combine_frustum_and_fsd

    def tap(*a):
        cap["combine_in"] = a
        return orig(*a)

    monkeypatch.setattr(model, "combine_frustum_and_fsd", tap)
    with torch.no_grad():
        res = model.simple_test([pts8.to(device)], [dict(lidar2img=L.to(device))], mask.to(device)[None], anno.to(device)[None])
        o = omod.simple_test(cpu, pts8, mask, anno, L)
        # the float64 yardstick: stage 1 (segmentor + fusion + segmentation head) and the camera stack in float64 from the same input
        cpu64 = copy.deepcopy(cpu).double()
        s1_64 = omod.fsf_stage1(cpu64, pts8, mask, anno, L, dtype=torch.float64)
        s2_64 = omod.fsf_stage2(cpu64, s1_64, anno, tuple(mask.shape[-2:]))
        # ... and the LiDAR stack (round 6): stage 3's arithmetic in float64 on the integer structure the fp32 chain decided
        # (pre-voxel cells, foreground masks, density filter, components: oracle/modules.py::fsf_stage3 `replay`)
        s3_64 = omod.fsf_stage3(cpu64, s1_64, replay=o["s3_decisions"])
        del cpu64
    monkeypatch.undo()
    gb, gs, gl = (res[0][k] for k in ("boxes_3d", "scores_3d", "labels_3d"))
    gb = gb.tensor.cpu().numpy()
    gs, gl = gs.cpu().numpy(), gl.cpu().numpy()
    ob, os_, ol = o["boxes"].numpy(), o["scores"].numpy(), o["labels"].numpy()
    pairs, iou, ds = match_boxes(gb, gs, gl, ob, os_, ol)
    good = (iou >= 0.99) & (ds <= 1e-3)
    c = lambda t: t.detach().cpu().numpy()  # noqa: E731
    f_centers, f_coors, _, f_feats, _, l_centers, l_coors, _, l_feats = cap["combine_in"]
    cam = _feature_deviation(c(f_feats)[:, :768], c(f_coors), o["s2"]["obj_feat"][:, :768].numpy(), o["s2"]["obj_coors"].numpy())
    lid = _feature_deviation(c(l_feats), c(l_coors), o["s3"]["cluster_feats"].numpy(), o["s3"]["cluster_inds"].numpy())
    assert np.array_equal(s2_64["obj_coors"].numpy(), o["s2"]["obj_coors"].numpy()), "the float64 chain changed the camera-query keys"
    cam_gpu64 = _feature_deviation(c(f_feats)[:, :768], c(f_coors), s2_64["obj_feat"][:, :768].numpy(), s2_64["obj_coors"].numpy())
    cam_o32_64 = _feature_deviation(o["s2"]["obj_feat"][:, :768].numpy(), o["s2"]["obj_coors"].numpy(),
                                    s2_64["obj_feat"][:, :768].numpy(), s2_64["obj_coors"].numpy())
    assert np.array_equal(s3_64["cluster_inds"].numpy(), o["s3"]["cluster_inds"].numpy()), "the float64 chain changed the LiDAR-query keys"
    lid_gpu64 = _feature_deviation(c(l_feats), c(l_coors), s3_64["cluster_feats"].numpy(), s3_64["cluster_inds"].numpy())
    lid_o32_64 = _feature_deviation(o["s3"]["cluster_feats"].numpy(), o["s3"]["cluster_inds"].numpy(),
                                    s3_64["cluster_feats"].numpy(), s3_64["cluster_inds"].numpy())
    report = dict(
        frame=which, points=int(pts8.shape[0]), gpu_boxes=int(gb.shape[0]), oracle_boxes=int(ob.shape[0]),
        matched=int(len(pairs)), matched_iou99_dscore1e3=int(good.sum()),
        unmatched_gpu=int(gb.shape[0] - len(pairs)), unmatched_oracle=int(ob.shape[0] - len(pairs)),
        min_matched_iou=float(iou.min()) if len(iou) else None, max_dscore=float(ds.max()) if len(ds) else None,
        median_one_minus_iou=float(np.median(1 - iou)) if len(iou) else None,
        camera_queries=dict(gpu=int(f_coors.shape[0]), oracle=int(o["s2"]["obj_coors"].shape[0]),
                            keys_identical=bool(np.array_equal(c(f_coors), o["s2"]["obj_coors"].numpy())), sir_feature_dev=cam,
                            sir_feature_dev_gpu_vs_float64=cam_gpu64, sir_feature_dev_oracle32_vs_float64=cam_o32_64),
        lidar_queries=dict(gpu=int(l_coors.shape[0]), oracle=int(o["s3"]["cluster_inds"].shape[0]),
                           keys_identical=bool(np.array_equal(c(l_coors), o["s3"]["cluster_inds"].numpy())), sir_feature_dev=lid,
                           sir_feature_dev_gpu_vs_float64=lid_gpu64, sir_feature_dev_oracle32_vs_float64=lid_o32_64),
        oracle_nms_margin=float(o["margin"]))
    print("\nE2E agreement:", json.dumps(report, indent=1))
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"e2e_agreement_{which}.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    # frozen from the MI355X measurement: every box the device returns has an oracle twin
    n = max(gb.shape[0], ob.shape[0])
    assert gb.shape[0] == ob.shape[0] > 0
    assert good.sum() >= E2E_MIN_MATCHED_FRACTION * n, report
    assert report["unmatched_gpu"] <= E2E_MAX_UNMATCHED and report["unmatched_oracle"] <= E2E_MAX_UNMATCHED, report
    assert report["camera_queries"]["keys_identical"] and report["lidar_queries"]["keys_identical"], report
    assert cam["p999"] <= E2E_MAX_CAMERA_SIR_DEV_P999 and lid["p999"] <= E2E_MAX_LIDAR_SIR_DEV_P999, report
    # arbitration against the float64 chain: the device is no farther from it than the fp32 oracle is (x 3, with a floor at the
    # per-kernel bound) — at the 99.9th percentile, at the median and at the maximum
    for q in ("median", "p999", "max"):
        assert cam_gpu64[q] <= E2E_F64_RATIO * cam_o32_64[q] + E2E_F64_FLOOR, (q, cam_gpu64, cam_o32_64)
        assert lid_gpu64[q] <= E2E_F64_RATIO * lid_o32_64[q] + E2E_F64_FLOOR, (q, lid_gpu64, lid_o32_64)
    assert cam_gpu64["p999"] <= E2E_MAX_CAMERA_SIR_DEV_VS_F64_P999, report
    assert lid_gpu64["p999"] <= E2E_MAX_LIDAR_SIR_DEV_VS_F64_P999, report
    # north_star's "fp features within 1e-4", stated for what it can mean end to end: the device's distance to exact arithmetic is at
    # most 1e-4 of the feature scale beyond the conditioning term — the distance an fp32 evaluation of the REFERENCE's own arithmetic
    # (the oracle chain) shows to the same float64 chain on the same frame
    assert cam_gpu64["p999"] <= E2E_CONTRACT + cam_o32_64["p999"], (cam_gpu64, cam_o32_64)
    assert lid_gpu64["p999"] <= E2E_CONTRACT + lid_o32_64["p999"], (lid_gpu64, lid_o32_64)


# Thresholds: measured first (round 4, MI355X, gpurun_out/e2e_agreement_*.json -> DESIGN.md section 3), then frozen with margin.
# Measured: 500 / 500 boxes matched on all three frames, worst matched IoU 0.99957, worst |dscore| 2.1e-5; query keys identical;
# SIR group features 99.9th percentile 5.6e-3 (camera stack: rel_mlp's LayerNorms amplify the 1e-5 m centroid rounding) and
# 3.4e-4 (LiDAR stack) of the feature scale.
E2E_MIN_MATCHED_FRACTION = 0.99      # of the 500 returned boxes, matched at BEV IoU >= 0.99 with |dscore| <= 1e-3
E2E_MAX_UNMATCHED = 5
E2E_MAX_CAMERA_SIR_DEV_P999 = 2e-2   # vs the fp32 ORACLE chain — whose own distance to the float64 chain is 5.7e-3 (10 sweeps), 3.5e-3 (AV2):
#                                      this bound limits the oracle's conditioning, the next three limit the device
# Round 5, measured on MI355X (profiles/r5_e2e_agreement_*.json): camera-stack SIR group features, 99.9th percentile of the row
# maximum relative to the feature scale — device vs the float64 chain 1.8e-4 / 9.0e-5 / 5.5e-5 (1 sweep / 10 sweeps / AV2), the fp32
# oracle vs the float64 chain 1.8e-4 / 5.7e-3 / 3.5e-3: the device is as close to float64 as the fp32 oracle on the small frame and
# 60 x closer on the large ones (its centroids and LayerNorm statistics are accumulated in blocked / pairwise order).
E2E_F64_RATIO = 3.0                  # |gpu - float64 chain| <= ratio x |fp32 oracle - float64 chain| + floor
E2E_F64_FLOOR = 1e-5
E2E_MAX_CAMERA_SIR_DEV_VS_F64_P999 = 3.6e-4  # = 2 x the largest measured (1.8e-4: the 1-sweep frame, where the fp32 ORACLE sits the same 1.8e-4
#                                             from float64 — groups of one to three points put f_cluster at ~0 in front of three LayerNorm(eps=1e-3))
# Round 6: the LiDAR stack arbitrated the same way (stage 3 in float64 on the fp32 chain's integer structure).  Measured on MI355X
# (profiles/r6_e2e_agreement_*.json; 1 sweep / 10 sweeps / AV2), 99.9th percentile of the row maximum relative to the feature scale:
#   device vs the fp32 oracle chain   3.2e-4 / 3.4e-4 / 2.5e-5
#   device vs the float64 chain       3.3e-4 / 3.8e-4 / 2.3e-4
#   fp32 ORACLE vs the float64 chain  3.4e-4 / 5.1e-4 / 2.3e-4   <- the frame's conditioning term: what ANY fp32 evaluation of the reference's
# arithmetic shows on these clusters (centroid rounding amplified by the position MLP's LayerNorms).  The device is never farther from
# float64 than the fp32 oracle is; the 1e-4 contract is asserted beyond that term.
E2E_MAX_LIDAR_SIR_DEV_P999 = 7e-4            # vs the fp32 oracle chain (was 2e-3): <= 2 x the measured 3.4e-4
E2E_MAX_LIDAR_SIR_DEV_VS_F64_P999 = 7.5e-4   # vs the float64 chain: <= 2 x the measured 3.8e-4
E2E_CONTRACT = 1e-4                          # north_star's feature tolerance, beyond the frame's fp32 conditioning term
