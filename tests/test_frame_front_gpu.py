"""K32, the frame front issued under the previous frame's tail (`FSF.set_next_frame`): a loop that announces every next frame must
return, frame by frame, the very same boxes / scores / labels and the very same query features as the loop that announces nothing —
the front is the same kernels on the same inputs in the same order, only earlier and on a side stream.  Also: an announced frame
that does not come (another frame is passed instead) is dropped and the frame computes its own front; the first frame of a loop
(nothing prefetched) and a repeated frame object are handled; the hot-path-only entry point prefetches as well."""
import numpy as np
import pytest
import torch

import bench

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(device):
    return bench.build_model(device)


@pytest.fixture(scope="module")
def frames(device):
    # three distinct frames: one single-sweep (one host thread, tiny launches), two 3-sweep (two host threads / side streams)
    return [bench.make_inputs(1, 3, device)[1], bench.make_inputs(3, 5, device)[1], bench.make_inputs(3, 9, device)[1]]


def _boxes(res):
    r = res[0]
    return r["boxes_3d"].tensor.numpy().copy(), r["scores_3d"].numpy().copy(), r["labels_3d"].numpy().copy()


def _args(f):
    return f["points"], f["img_metas"], f["mask_data"], f["mask_anno"]


def _loop(model, frames, order, announce, wrong_announcement_at=()):
    out = []
    with torch.no_grad():
        for k, i in enumerate(order):
            if announce and k + 1 < len(order):
                nxt = frames[order[k + 1]]
                if k in wrong_announcement_at:  # announce a frame that will NOT be the next call
                    nxt = frames[(order[k + 1] + 1) % len(frames)]
                model.set_next_frame(*_args(nxt))
            out.append(_boxes(model.simple_test(*_args(frames[i]))))
    torch.cuda.synchronize()
    return out


ORDER = [0, 1, 2, 1, 1, 0, 2, 0]


@pytest.fixture(scope="module")
def plain(model, frames):
    res = _loop(model, frames, ORDER, announce=False)
    assert all(len(b) > 20 for b, _, _ in res)
    return res


def _same(a, b):
    assert len(a) == len(b)
    for (b0, s0, l0), (b1, s1, l1) in zip(a, b):
        np.testing.assert_array_equal(l1, l0)
        np.testing.assert_array_equal(s1, s0)
        np.testing.assert_array_equal(b1, b0)


def test_announced_loop_returns_the_plain_loop_bit_for_bit(model, frames, plain):
    _same(plain, _loop(model, frames, ORDER, announce=True))
    assert model._front_ready is None  # nothing was announced behind the last frame


def test_every_front_after_the_first_was_the_prefetched_one(model, frames):
    taken = []
    orig = model._frame_front

    def counting(*a, **k):
        taken.append(torch.cuda.current_stream() == getattr(model, "_front_stream", None))
        return orig(*a, **k)

    model._frame_front = counting
    try:
        _loop(model, frames, ORDER, announce=True)
    finally:
        del model.__dict__["_frame_front"]
    # one front per frame; the first on the caller's stream (nothing announced before it), every other one on the front stream
    assert taken == [False] + [True] * (len(ORDER) - 1)


def test_an_announced_frame_that_does_not_come_is_dropped(model, frames, plain):
    _same(plain, _loop(model, frames, ORDER, announce=True, wrong_announcement_at=(0, 3, 4)))


def test_announcing_and_then_stopping_leaves_nothing_behind(model, frames, plain):
    with torch.no_grad():
        model.set_next_frame(*_args(frames[2]))
        first = _boxes(model.simple_test(*_args(frames[0])))
        assert model._front_ready is not None          # frame 2's front exists ...
        again = _boxes(model.simple_test(*_args(frames[0])))  # ... but frame 0 comes again: dropped
        assert model._front_ready is None
    _same([plain[0], plain[0]], [first, again])


def test_hot_path_only_prefetches_too(model, frames):
    with torch.no_grad():
        ref = [model.forward_hot_path(*_args(frames[i])) for i in (1, 2, 1)]
        ref = [{k: v.clone() for k, v in r.items() if torch.is_tensor(v)} for r in ref]
        got = []
        for k, i in enumerate((1, 2, 1)):
            if k < 2:
                model.set_next_frame(*_args(frames[(2, 1)[k]]))
            got.append(model.forward_hot_path(*_args(frames[i])))
    torch.cuda.synchronize()
    for r, g in zip(ref, got):
        for k, v in r.items():
            assert torch.equal(v, g[k]), k


def test_training_mode_and_grad_mode_never_prefetch(model, frames):
    model.set_next_frame(*_args(frames[1]))
    with torch.enable_grad():
        model._prefetch_front()
    assert model._front_ready is None and "_next_frame" not in model.__dict__


def test_pre_voxelize_keys_formed_early_equal_the_ones_formed_in_place(model, frames, plain):
    """`FSF._pre_voxel_keys_early` (the 0.1 m keys + their unique on the front stream while the segmentor runs) against
    `pre_voxelize` forming them itself: the same boxes bit for bit, and the early form is the one the default path takes."""
    used = []
    orig = model.pre_voxelize

    def spying(d):
        used.append(model.__dict__.get("_pre_vox") is not None)
        return orig(d)

    model.pre_voxelize = spying
    try:
        _loop(model, frames, ORDER[:3], announce=False)
    finally:
        del model.__dict__["pre_voxelize"]
    assert used == [True] * 3
    model._pre_voxel_keys_early = lambda *a, **k: None
    try:
        _same(plain, _loop(model, frames, ORDER, announce=False))
        _same(plain, _loop(model, frames, ORDER, announce=True))
    finally:
        del model.__dict__["_pre_voxel_keys_early"]


def test_camera_rows_formed_early_equal_the_ones_formed_in_place(model, frames, plain):
    """`FSF._camera_rows_early` (the camera branch's row list + the unique of its keys, on the front stream while the segmentor runs)
    against `frustum_forward` forming them itself: the same boxes bit for bit; the early form is what the default path takes."""
    assert model.__dict__.get("_cam_rows_hold") is not None and model._cam_rows_hold["rows"] is not None
    # (frames below `concurrent_query_min_points` run both branches on the calling thread: the case the early form is on for)
    model._camera_rows_early = lambda *a, **k: None
    try:
        _same(plain, _loop(model, frames, ORDER, announce=False))
        _same(plain, _loop(model, frames, ORDER, announce=True))
    finally:
        del model.__dict__["_camera_rows_early"]


def test_announced_loop_without_the_plan_and_lateral_streams(device, frames, plain, monkeypatch):
    """`bench.py --serial`'s U-Net (no plan stream: the rulebooks are allocated on whatever stream `begin` runs on — the FRONT stream for
    an announced frame) under an announced loop.  The first version of K32 let those tables go back to the front stream's allocator when
    the forward's generator ended, the early key work re-used them while the convolutions that read them were still queued, and the
    process died with a memory access fault; they are now held like every other cross-stream tensor of the forward."""
    from fullysparsefusion_amd import switches

    monkeypatch.setattr(switches, "UNET_PLAN_STREAM", False)
    monkeypatch.setattr(switches, "UNET_LATERAL_STREAM", False)
    model = bench.build_model(device)
    for _ in range(3):  # (the hazard needs the allocator's pools warm: several passes over the loop)
        got = _loop(model, frames, ORDER, announce=True)
    _same(plain, got)
