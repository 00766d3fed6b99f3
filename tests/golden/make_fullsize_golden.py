#!/usr/bin/env python
"""Generates tests/golden/stage1_10sweep.npz: the CPU ORACLE's stage-1 output (voxelize -> DynamicScatterVFE ->
SimpleSparseUNet -> neck -> projection + mask gather + image fusion -> segmentation head;
FSF.simple_test, projects/mmdet3d_plugin/models/detectors/FSF.py:1123-1130) on the full-size synthetic 10-sweep frame
bench.py times (BASELINE.json config 3: 310 615 points, u8[6,10,900,1600] masks), with the detector of
`tests/conftest.py::build_test_fsf`.

The oracle takes minutes at this size, so it is run ONCE here (build container, CPU) and a row-sampled fixture is
committed: data only (sampled rows of the expected tensors, full-tensor scales and integer checksums, a parameter
checksum of the model so that a drifted random init is detected instead of silently compared).  The GPU test
(tests/test_fullsize_gpu.py) runs the HIP path on the same frame and compares the same rows.

    python tests/golden/make_fullsize_golden.py            # stage1_10sweep.npz
    python tests/golden/make_fullsize_golden.py av2        # av2_segmentor_150k.npz: BASELINE config 5's segmentor (VoteSegmentor.extract_feat,
                                                           # single_stage_fsd.py:228-245, configs/Argoverse2/FSF_AV2_config.py:84-94 U-Net) on the
                                                           # 150 k-point +-200 m frame `bench.py --dataset av2` times
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import build_av2_fsf, build_test_fsf, param_checksum  # noqa: E402
from fullysparsefusion_amd import synthetic  # noqa: E402
from oracle import modules as omod  # noqa: E402

N_ROWS = 2048


def sample_rows(n, k, seed):
    return np.sort(np.random.default_rng(seed).choice(n, size=min(k, n), replace=False)).astype(np.int64)


def main():
    model = build_test_fsf()
    f = synthetic.make_frame(num_sweeps=10, seed=0)
    pts8 = torch.from_numpy(f["points"])
    t0 = time.perf_counter()
    with torch.no_grad():
        s1 = omod.fsf_stage1(model, pts8, torch.from_numpy(f["mask_data"]), torch.from_numpy(f["mask_anno"]),
                             torch.from_numpy(f["lidar2img"]))
    print(f"oracle stage 1 on {pts8.shape[0]} points: {time.perf_counter() - t0:.1f} s", flush=True)
    ex = s1["ex"]
    n, m = pts8.shape[0], ex["voxel_coors"].shape[0]
    prow, vrow = sample_rows(n, N_ROWS, 1), sample_rows(m, N_ROWS, 2)
    out = dict(
        param_checksum=np.float64(param_checksum(model)),
        num_points=np.int64(n), num_voxels=np.int64(m),
        point_rows=prow, voxel_rows=vrow,
        # integer outputs: full checksums + sampled rows
        voxel_coors_rows=ex["voxel_coors"].numpy()[vrow],
        voxel_coors_colsum=ex["voxel_coors"].numpy().astype(np.int64).sum(0),
        inv_rows=ex["inv"].numpy()[prow],
        inv_sum=np.int64(ex["inv"].numpy().astype(np.int64).sum()),
        obj_id_rows=s1["obj_id"].numpy()[prow],
        obj_id_sum=np.int64(s1["obj_id"].numpy().sum()),
        obj_id_nonzero=np.int64((s1["obj_id"].numpy() > 0).sum()),
    )
    for name, t, rows in [("voxel_feats", ex["voxel_feats"], vrow), ("unet", ex["unet"], vrow), ("neck", ex["neck"], prow),
                          ("seg_feats", s1["seg_feats"], prow), ("seg_logits", s1["seg_logits"], prow),
                          ("seg_vote_preds", s1["seg_vote_preds"], prow), ("offsets", s1["offsets"], prow)]:
        a = t.numpy()
        out[name + "_rows"] = a[rows]
        out[name + "_scale"] = np.float32(np.abs(a).max())
        out[name + "_abs_mean"] = np.float64(np.abs(a).astype(np.float64).mean())
    path = os.path.join(HERE, "stage1_10sweep.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main_av2():
    model = build_av2_fsf()
    f = synthetic.make_frame_av2(seed=0)
    pts = torch.from_numpy(f["points"][:, :4].copy())
    t0 = time.perf_counter()
    with torch.no_grad():
        ex = omod.segmentor_extract_feat(model.segmentor, [pts])
    print(f"oracle AV2 segmentor on {pts.shape[0]} points: {time.perf_counter() - t0:.1f} s", flush=True)
    n, m = pts.shape[0], ex["voxel_coors"].shape[0]
    prow, vrow = sample_rows(n, N_ROWS, 3), sample_rows(m, N_ROWS, 4)
    out = dict(
        param_checksum=np.float64(param_checksum(model.segmentor)),
        num_points=np.int64(n), num_voxels=np.int64(m), point_rows=prow, voxel_rows=vrow,
        coors_rows=ex["coors"].numpy()[prow], coors_colsum=ex["coors"].numpy().astype(np.int64).sum(0),
        voxel_coors_rows=ex["voxel_coors"].numpy()[vrow], voxel_coors_colsum=ex["voxel_coors"].numpy().astype(np.int64).sum(0),
        inv_rows=ex["inv"].numpy()[prow], inv_sum=np.int64(ex["inv"].numpy().astype(np.int64).sum()),
    )
    for name, t, rows in [("voxel_feats", ex["voxel_feats"], vrow), ("unet", ex["unet"], vrow), ("neck", ex["neck"], prow)]:
        a = t.numpy()
        out[name + "_rows"] = a[rows]
        out[name + "_scale"] = np.float32(np.abs(a).max())
        out[name + "_abs_mean"] = np.float64(np.abs(a).astype(np.float64).mean())
    path = os.path.join(HERE, "av2_segmentor_150k.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main_av2() if "av2" in sys.argv[1:] else main()
