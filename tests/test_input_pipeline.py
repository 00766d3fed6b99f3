"""CPU: the test-time input pipeline (on-disk .bin sweeps, PNG id planes, anno.json -> the tensors FSF.simple_test
takes) against vectors produced by the reference's own pipeline classes (tests/golden/make_golden.py::gen_input_pipeline)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from fullysparsefusion_amd.mmdet3d_plugin import datasets as D
from fullysparsefusion_amd.mmdet3d_plugin.registry import PIPELINES

CLASSES = ["car", "truck", "trailer", "bus", "construction_vehicle", "bicycle", "motorcycle", "pedestrian", "traffic_cone", "barrier"]
PC_RANGE = [-51.2, -51.2, -5, 51.2, 51.2, 3]


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    """Re-creates the synthetic sample directory from the golden's file contents."""
    from PIL import Image

    g = load_golden("input_pipeline.npz")
    root = tmp_path_factory.mktemp("frame")
    sdir = root / "masks" / "sample0"
    sdir.mkdir(parents=True)
    for cam in range(6):
        for ci, name in enumerate(CLASSES):
            Image.fromarray(g["planes"][cam, ci]).save(sdir / f"{cam}_{name}.png")
    (sdir / "anno.json").write_bytes(bytes(g["anno_json"]))
    g["key"].tofile(root / "key.bin")
    meta = json.loads(bytes(g["sweep_meta_json"]).decode())
    for k, m in enumerate(meta):
        g["sweeps"][k].tofile(root / f"sweep{k}.bin")
        m["data_path"] = str(root / f"sweep{k}.bin")
    return g, root, meta


def test_point_loading_multisweep_noaug_normalize(sample):
    g, root, meta = sample
    r = D.LoadPointsFromFile(coord_type="LIDAR", load_dim=5, use_dim=[0, 1, 2, 3, 4])(dict(pts_filename=str(root / "key.bin")))
    np.testing.assert_array_equal(r["points"].tensor.numpy(), g["loaded"])
    r.update(timestamp=1.5e9, sweeps=meta)
    r = D.LoadPointsFromMultiSweeps(sweeps_num=9, use_dim=[0, 1, 2, 3, 4], pad_empty_sweeps=True, remove_close=True, test_mode=True)(r)
    np.testing.assert_array_equal(r["points"].tensor.numpy(), g["multi"])  # sweep transform, time lag, close-point removal
    r = D.SaveNoAugPoints()(r)
    np.testing.assert_array_equal(r["points"].tensor.numpy(), g["saved"])
    r = D.NormalizePoints()(r)
    np.testing.assert_array_equal(r["points"].tensor.numpy(), g["normed"])
    # no sweeps on record: the key frame is repeated (close points removed from the copies)
    r2 = D.LoadPointsFromFile(coord_type="LIDAR", load_dim=5, use_dim=[0, 1, 2, 3, 4])(dict(pts_filename=str(root / "key.bin")))
    r2.update(timestamp=1.5e9, sweeps=[])
    r2 = D.LoadPointsFromMultiSweeps(sweeps_num=2, use_dim=[0, 1, 2, 3, 4], pad_empty_sweeps=True, remove_close=True)(r2)
    np.testing.assert_array_equal(r2["points"].tensor.numpy(), g["padded"])


def test_mask_loading_nuscenes(sample):
    g, root, _ = sample
    r = D.LoadMaskFromFiles(data_path=str(root / "masks"), class_names=CLASSES)(dict(sample_idx="sample0"))
    assert r["mask_data"].dtype == torch.uint8  # stays an integer plane
    np.testing.assert_array_equal(r["mask_data"].numpy(), g["mask_data"])
    np.testing.assert_array_equal(r["mask_anno"].numpy(), g["mask_anno"])  # rows sorted by obj id, padded to 250, valid flag
    loader = D.LoadMaskFromFiles(data_path="unused")
    single = loader.reorg_anno_single_cls([[dict(bbox=[1.0, 2.0, 3.0, 4.0], score=0.5, category=3, cam_id=0, obj_id=7)], [],
                                           [dict(bbox=[5.0, 6.0, 7.0, 8.0], score=0.25, category=1, cam_id=2, obj_id=2)]])
    np.testing.assert_array_equal(single.numpy(), g["single_anno"])


def test_argoverse_loader_resizes_the_front_camera(tmp_path):
    from PIL import Image

    sdir = tmp_path / "uuid0"
    sdir.mkdir()
    rng = np.random.default_rng(0)
    small = rng.integers(0, 600, (31, 41)).astype(np.uint16)  # ids beyond 255: 16-bit planes
    Image.fromarray(small).save(sdir / "0.png")
    full = rng.integers(0, 600, (1550, 2048)).astype(np.uint16)
    for i in range(1, 7):
        Image.fromarray(full).save(sdir / f"{i}.png")
    anno = [[dict(bbox=[4.0, 3.0, 20.0, 12.0], score=0.9, category=2, cam_id=0, obj_id=1)]] + [[] for _ in range(6)]
    (sdir / "anno.json").write_text(json.dumps(anno))
    l2i = [np.eye(4, dtype=np.float32) * (i + 1) for i in range(7)]
    r = D.LoadMaskFromFiles(data_path=str(tmp_path), class_names=["x"], is_argo=True)(dict(img_info=dict(uuid="uuid0"), lidar2img=l2i))
    assert r["mask_data"].shape == (7, 1, 1550, 2048) and r["mask_data"].dtype == torch.int32
    hf, wf = 1550 / 31, 2048 / 41
    ys = np.minimum(np.floor(np.arange(1550) * (31 / 1550)).astype(int), 30)
    xs = np.minimum(np.floor(np.arange(2048) * (41 / 2048)).astype(int), 40)
    np.testing.assert_array_equal(r["mask_data"][0, 0].numpy(), small[ys][:, xs].astype(np.int32))
    np.testing.assert_array_equal(r["mask_data"][3, 0].numpy(), full.astype(np.int32))
    np.testing.assert_allclose(r["mask_anno"][0, :4].numpy(), [4 * wf, 3 * hf, 20 * wf, 12 * hf], rtol=1e-6)
    np.testing.assert_allclose(r["lidar2img"][0][0, 0], wf, rtol=1e-6)
    np.testing.assert_allclose(r["lidar2img"][0][1, 1], hf, rtol=1e-6)
    assert r["lidar2img"][1][0, 0] == 2.0


def test_reference_test_pipeline_config_runs_end_to_end(sample):
    """The `test_pipeline` list of projects/configs/_base_/datasets/nuscenes_dataloader.py:97-136, restated, through
    Compose / the PIPELINES registry, then the one host->device step."""
    g, root, meta = sample
    pipeline = [
        dict(type="LoadPointsFromFile", coord_type="LIDAR", load_dim=5, use_dim=[0, 1, 2, 3, 4]),
        dict(type="LoadPointsFromMultiSweeps", sweeps_num=9, use_dim=[0, 1, 2, 3, 4], pad_empty_sweeps=True, remove_close=True),
        dict(type="SaveNoAugPoints"),
        dict(type="LoadMaskFromFiles", data_path=str(root / "masks"), class_names=CLASSES),
        dict(type="MultiScaleFlipAug3D", img_scale=(1333, 800), pts_scale_ratio=1, flip=False, transforms=[
            dict(type="GlobalRotScaleTrans", rot_range=[0, 0], scale_ratio_range=[1.0, 1.0], translation_std=[0, 0, 0]),
            dict(type="RandomFlip3D"),
            dict(type="PointsRangeFilter", point_cloud_range=PC_RANGE),
            dict(type="NormalizePoints"),
            dict(type="DefaultFormatBundle3D", class_names=CLASSES, with_label=False),
            dict(type="Collect3D", keys=["points", "mask_data", "mask_anno"])]),
    ]
    data = D.Compose(pipeline)(dict(pts_filename=str(root / "key.bin"), timestamp=1.5e9, sweeps=meta, sample_idx="sample0",
                                    lidar2img=[np.eye(4, dtype=np.float32)] * 6))
    pts = data["points"][0]
    want = g["normed"]
    keep = ((want[:, 0] > PC_RANGE[0]) & (want[:, 1] > PC_RANGE[1]) & (want[:, 2] > PC_RANGE[2]) & (want[:, 0] < PC_RANGE[3]) &
            (want[:, 1] < PC_RANGE[4]) & (want[:, 2] < PC_RANGE[5]))
    np.testing.assert_array_equal(pts.numpy(), want[keep])
    assert 0 < keep.sum() < len(keep) and pts.shape[1] == 8
    points, metas, mask, anno = D.frame_to_device(data, torch.device("cpu"))
    assert points[0].shape == pts.shape and mask.shape == (1, 6, 10, 45, 80) and mask.dtype == torch.uint8
    assert anno.shape == (1, 250, 9) and metas[0]["lidar2img"].shape == (6, 4, 4)
    for name in ("ObjectSample", "PointShuffle"):  # train-time steps stay placeholders that refuse to run
        with pytest.raises(NotImplementedError):
            PIPELINES.build(dict(type=name))(dict())
