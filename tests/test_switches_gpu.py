"""Every dispatch switch of fullysparsefusion_amd/switches.py ("<= 12 switches, each with a test"): the whole
`FSF.simple_test` forward of one frame with the switch OFF against the same forward on the defaults.  A switch that only re-schedules
or re-routes identical arithmetic (side streams, row order, fused box tail, sorted SIR stacks, direct refine groups) must return the
very same boxes; a switch that changes how a product is formed (f16 x 3 planes against bf16 x 6 / the fp32 kernel) must return the
same detections within the frame-level bound the un-restarted e2e tests use (box-to-box IoU ~1, |score| <= 2e-3).
(The two training switches — TRAIN_PLANES, SYNCBN_FUSED — have their off-settings in tests/test_hip_ops.py::test_spconv_autograd_*,
tests/test_distributed_cpu.py::test_fused_syncbn_relu_equals_upstream_formulation_world2_gloo and tests/test_distributed_gpu.py.)"""
import numpy as np
import pytest
import torch

import bench
from fullysparsefusion_amd import switches

pytestmark = pytest.mark.gpu

SCHEDULING = ["UNET_LATERAL_STREAM", "UNET_PLAN_STREAM", "UNET_MASK_ORDER", "BOX_TAIL_FUSED", "SIR_SORTED", "REFINE_DIRECT"]
ARITHMETIC = ["PLANES", "K22F", "K22H"]


def test_the_switch_list_is_the_one_this_file_covers():
    assert set(switches.ALL) == set(SCHEDULING + ARITHMETIC + ["TRAIN_PLANES", "SYNCBN_FUSED"]) and len(switches.ALL) <= 12
    src = open(switches.__file__).read()
    import re

    assert sorted(re.findall(r'_on\("FSF_([A-Z0-9_]+)"\)', src)) == sorted(switches.ALL)


@pytest.fixture(scope="module")
def frame(device):
    return bench.make_inputs(1, 3, device)[1]


@pytest.fixture(autouse=True)
def _side_streams_on_small_frames(monkeypatch):
    monkeypatch.setattr(switches, "UNET_LATERAL_MIN_ROWS", 0)  # (a tuning constant keeps the lateral stream off frames this small)


def _forward(device, frame):
    model = bench.build_model(device)  # (fresh: prepared weights are cached per module in the format the switches chose)
    with torch.no_grad():
        res = model.simple_test(frame["points"], frame["img_metas"], frame["mask_data"], frame["mask_anno"])[0]
    return res["boxes_3d"].tensor.numpy(), res["scores_3d"].numpy(), res["labels_3d"].numpy()


@pytest.fixture(scope="module")
def default_result(device, frame):
    assert all(getattr(switches, n) for n in switches.ALL), "the suite runs on the defaults"
    keep, switches.UNET_LATERAL_MIN_ROWS = switches.UNET_LATERAL_MIN_ROWS, 0
    try:
        return _forward(device, frame)
    finally:
        switches.UNET_LATERAL_MIN_ROWS = keep


@pytest.mark.parametrize("name", SCHEDULING + ARITHMETIC)
def test_switch_off_returns_the_default_detections(device, frame, default_result, monkeypatch, name):
    monkeypatch.setattr(switches, name, False)
    boxes, scores, labels = _forward(device, frame)
    b0, s0, l0 = default_result
    assert len(b0) > 20
    if name in SCHEDULING:
        np.testing.assert_array_equal(labels, l0)
        np.testing.assert_array_equal(scores, s0)
        np.testing.assert_array_equal(boxes, b0)
        return
    # another arithmetic for the same products: same detections up to rounding (a box at the score threshold or an NMS tie may flip)
    key0 = {(int(l), round(float(s), 3)) for l, s in zip(l0, s0)}
    key1 = {(int(l), round(float(s), 3)) for l, s in zip(labels, scores)}
    assert len(key0 ^ key1) <= max(2, len(b0) // 50), (name, len(key0 ^ key1), len(b0))
    n = min(len(b0), len(boxes))
    same = labels[:n] == l0[:n]
    assert same.mean() > 0.95
    assert np.abs(scores[:n][same] - s0[:n][same]).max() <= 2e-3
    assert np.median(np.abs(boxes[:n][same] - b0[:n][same]).max(1)) <= 1e-3
