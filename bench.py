#!/usr/bin/env python
"""bench.py — frames/sec of the FSF forward on synthetic nuScenes-shape 10-sweep frames (BASELINE.json metric), one
process per GPU.

    python bench.py                                   # 1 GPU, 20 steps, 5 warm-up
    python bench.py --gpus N --steps K --warmup W     # N > 1 without a launcher: re-executes itself under
                                                      # torch.distributed.run (one process per GPU, RCCL), like
                                                      # tools/dist_train.sh:8-9 of the reference
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full forward of the detector (FSF.simple_test: voxelize -> DynamicScatterVFE -> SimpleSparseUNet ->
neck -> projection + mask gather + image fusion + seg head -> camera-query grouping + SIR -> LiDAR-query pre-voxelize,
sampling, device CCL, SIR -> heads -> query combination -> RoI point pooling + refine SIR -> box decode + rotated BEV
NMS -> results on the host) over ONE frame whose inputs are already resident in HBM; the timed loop rotates over
`--frames` (default 4) DISTINCT synthetic frames so that no step re-runs the frame whose rulebook-shaped access pattern the
caches saw last.  `--hot-path-only` stops after the three query-generation stages; `--train` times fwd + bwd of a dummy
loss + gradient all-reduce + AdamW instead.  Frames are independent, so for inference N ranks are N replicas with no
data-path collective (weak scaling); the only collectives are the barrier and the max-over-ranks of the timed region.

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  * `roofline`: the dominant kernel (sparse-conv forward) timed with HIP events in an instrumented pass over the same
    frames, fp32-equivalent flops against the ceiling of the matrix pipe the kernel actually issues on, plus the contract's
    fp32-pipe peak, per-kernel split, `hbm` = GB/s and fraction of 8 TB/s of the scatter / gather / segmented-reduce /
    projection / fused-linear kernels (algorithmic bytes of SURVEY.md section 8d / HIP-event time), and `frame_roofline_ms`;
  * `cpu_baseline`: the CPU oracle (kind "port") on ONE FULL 10-sweep frame through stages 1-3, per-stage times.
"""
import argparse
import copy
import json
import os
import re
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3      # v_mfma_f32_* dense
MFMA_16BIT_PEAK_TFLOPS = 2500.0   # v_mfma_f32_16x16x32_{bf16,f16} dense
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sweeps", type=int, default=10, help="10 = BASELINE config 3 input; 1 = config 2 (parity case)")
    ap.add_argument("--frames", type=int, default=4, help="distinct synthetic frames the timed loop rotates over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-describe", action="store_true",
                    help="skip the extra untimed stages-1-3 pass that counts the queries (profiling runs: the per-frame "
                         "kernel table then holds exactly warmup + steps forwards)")
    ap.add_argument("--train", action="store_true",
                    help="time the training step instead (BASELINE configs 3/4): fwd + bwd of a dummy scalar loss over the "
                         "hot-path outputs + bucketed gradient all-reduce over RCCL + AdamW")
    ap.add_argument("--frames-per-gpu", type=int, default=1, help="batch size per rank (config 4 uses 2)")
    ap.add_argument("--dataset", choices=["nuscenes", "av2"], default="nuscenes",
                    help="nuscenes = BASELINE config 3 input (the headline); av2 = config 5 shape (+-204.8 m, 2048^2 x 32 grid, 7 cams, "
                         "int32 id planes, 26 classes)")
    ap.add_argument("--trained-like", action="store_true",
                    help="time the trained-like variant INSTEAD of the headline workload: 40 small 2-D instances per frame and "
                         "segmentation / classification biases calibrated so that ~5 %% of the points are foreground and a few hundred "
                         "boxes reach the NMS (what FSF_nuScenes_config.py:22-30,377-384 thresholds leave with trained weights); "
                         "the default run reports it BESIDE the headline (`trained_like`)")
    ap.add_argument("--no-trained-like", action="store_true", help="skip the trained-like side measurement of the default run")
    ap.add_argument("--serial", action="store_true",
                    help="profiling runs: the two query branches back to back on one stream, the U-Net's lateral / plan streams off — every "
                         "kernel then runs alone on the device, so a rocprofv3 trace of this setting holds IN-SITU kernel durations "
                         "(profiles/*_kernel_stats_full_forward_serial.txt, which `roofline.hbm[*].in_situ_*` is computed from)")
    ap.add_argument("--no-frame-front", action="store_true",
                    help="do not announce the next frame to the detector (FSF.set_next_frame): every frame's front is issued inside "
                         "its own frame, as in rounds 1-5")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-buffer (PCIe-inclusive) side measurement of the default run")
    ap.add_argument("--no-train-block", action="store_true", help="skip the short training-step side measurement of the default run")
    ap.add_argument("--hot-path-only", action="store_true",
                    help="time stages 1-3 only (segmentor + fusion, camera queries, LiDAR queries), no heads / refine / NMS")
    return ap.parse_args()


def build_model(device, dataset="nuscenes"):
    from fullysparsefusion_amd import mmdet3d_plugin as plugin
    from fullysparsefusion_amd.compat import Config

    torch.manual_seed(0)
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "fsf_nuscenes.py" if dataset == "nuscenes" else "fsf_av2.py"))
    model = plugin.build_model(cfg.model).eval()
    # random-init weights of the reference architecture (no checkpoint on the box); un-zero the image branch's last
    # layer (zero-init upstream, FSF.py:142-143) so the fusion arithmetic is not trivially zero
    torch.nn.init.normal_(model.segmentor_updated_mlp[-1].weight, std=0.02)
    return model.to(device)


def make_inputs(sweeps, seed, device, frames=1, dataset="nuscenes", trained_like=False):
    """One batch (`frames` frames, normally 1) resident on `device`; returns (first host frame, device batch)."""
    from fullysparsefusion_amd import synthetic

    if dataset == "av2":
        fs = [synthetic.make_frame_av2(seed=seed * 97 + i) for i in range(frames)]
    elif trained_like:
        fs = [synthetic.make_frame(num_sweeps=sweeps, seed=seed * 97 + i, mask_instances=40, mask_max_area=0.02) for i in range(frames)]
    else:
        fs = [synthetic.make_frame(num_sweeps=sweeps, seed=seed * 97 + i) for i in range(frames)]
    dev = dict(
        points=[torch.from_numpy(f["points"]).to(device) for f in fs],
        mask_data=torch.stack([torch.from_numpy(f["mask_data"]) for f in fs]).to(device),
        mask_anno=torch.stack([torch.from_numpy(f["mask_anno"]) for f in fs]).to(device),
        img_metas=[dict(lidar2img=torch.from_numpy(f["lidar2img"]).to(device)) for f in fs],
    )
    return fs[0], dev


def h2d_inclusive(model, sweeps, seeds, device, steps, warmup, ms_resident, announce=True):
    """The boundary hands over HOST buffers (`datasets/pipelines.py::frame_to_device`: points f32 [n, 8], the u8 id planes
    [1, 6, 10, 900, 1600] exactly as LoadMaskFromFiles leaves them, mask_anno, lidar2img — datasets/pipelines/loading.py:213-234,
    :301-339, :781-877): the same forward with every frame starting in PINNED host memory.  Two device slots; frame i + 1's copies are
    issued on a copy stream before frame i's forward is (a slot is re-filled only after the forward that read it has ended), so the
    transfer rides behind the previous frame.  Reports the frames/s of that loop, the exposed transfer time per frame (this loop's
    ms/frame - the HBM-resident loop's, same run) and the transfer alone."""
    from fullysparsefusion_amd import synthetic

    frames = [synthetic.make_frame(num_sweeps=sweeps, seed=sd * 97) for sd in seeds]
    host = [dict(points=torch.from_numpy(f["points"]).pin_memory(), mask_data=torch.from_numpy(f["mask_data"])[None].pin_memory(),
                 mask_anno=torch.from_numpy(f["mask_anno"])[None].pin_memory(), lidar2img=torch.from_numpy(f["lidar2img"]).pin_memory())
            for f in frames]
    nmax = max(h["points"].shape[0] for h in host)
    amax = max(h["mask_anno"].shape[1] for h in host)
    slots = [dict(points=torch.empty((nmax, host[0]["points"].shape[1]), dtype=torch.float32, device=device),
                  mask_data=torch.empty_like(host[0]["mask_data"], device=device),
                  mask_anno=torch.empty((1, amax) + tuple(host[0]["mask_anno"].shape[2:]), dtype=host[0]["mask_anno"].dtype, device=device),
                  lidar2img=torch.empty_like(host[0]["lidar2img"], device=device)) for _ in range(2)]
    copy_stream = torch.cuda.Stream()
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]
    main = torch.cuda.current_stream()
    for e in free:
        e.record(main)

    def upload(i):
        h, sl = host[i % len(host)], slots[i % 2]
        n, a = h["points"].shape[0], h["mask_anno"].shape[1]
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(free[i % 2])
            sl["points"][:n].copy_(h["points"], non_blocking=True)
            sl["mask_data"].copy_(h["mask_data"], non_blocking=True)
            sl["mask_anno"][:, :a].copy_(h["mask_anno"], non_blocking=True)
            sl["lidar2img"].copy_(h["lidar2img"], non_blocking=True)
            ready[i % 2].record(copy_stream)
        return dict(points=[sl["points"][:n]], mask_data=sl["mask_data"], mask_anno=sl["mask_anno"][:, :a],
                    img_metas=[dict(lidar2img=sl["lidar2img"])], ready=ready[i % 2])

    def loop(k):
        nxt = upload(0)
        for i in range(k):
            cur = nxt
            if i + 1 < k:
                nxt = upload(i + 1)  # (behind frame i: issued before its forward, on the copy stream)
            main.wait_event(ready[i % 2])
            # (frame i + 1 is announced with its upload's event: the detector's front stream waits for the copy, not the host)
            step(model, cur, nxt=nxt if announce and i + 1 < k else None)
            free[i % 2].record(main)

    loop(max(warmup, 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # the transfer alone (no forward beside it)
    t0 = time.perf_counter()
    for i in range(4):
        upload(i)
        free[i % 2].record(copy_stream)
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / 4 * 1e3
    mb = sum(v.numel() * v.element_size() for v in host[0].values()) / 1e6
    return dict(value_incl_h2d=round(1e3 / ms, 3), unit="frames/s", ms_per_step_incl_h2d=round(ms, 3), steps=steps,
                exposed_h2d_ms=round(ms - ms_resident, 3), h2d_alone_ms=round(alone, 3), h2d_mb_per_frame=round(mb, 1),
                h2d_gb_per_s_alone=round(mb / alone, 1),
                note="frames start in pinned host memory; two device slots, frame i + 1 copied on a copy stream while frame i runs; "
                     "nothing announced to the detector in this loop; `exposed_h2d_ms` = this loop's ms/frame - the UNANNOUNCED "
                     "HBM-resident loop of the same run (`frame_front.ms_per_step_unannounced`; the noise of either is ~0.1 ms); the "
                     "headline `value` stays the HBM-resident rate")


def dtype_text(train):
    """The arithmetic the timed path computes in, from the switches it ran with (the text used to say "exact splits ...
    bf16 x 6 in the fused Linear kernels" while K22f / K22h run f16 x 3)."""
    from fullysparsefusion_amd import switches

    conv = "f16 x 3 passes (row-scaled 22-bit hi | lo planes) in the sparse convolutions" if switches.PLANES else \
        "bf16 x 6 passes (exact 3-way split) in the sparse convolutions"
    lin = []
    if switches.K22F:
        lin.append("f16 x 3 (x split per row in the kernel, 22 bits relative to the row maximum) in the <= 128-channel-slice Linear "
                   "kernels K22 / K22s")
    else:
        lin.append("bf16 x 6 (exact split) in the <= 128-channel-slice Linear kernels K22 / K22s")
    if switches.K22H:
        lin.append("f16 x 3 planes in the >= 256-wide head Linears (K22h)")
    lin.append("bf16 x 6 (exact split) in the sliced head branches" + (" and the weight gradients (K10p)" if train else ""))
    return ("f32 (I/O and accumulation fp32; matrix products on the 16-bit matrix cores as splits of the fp32 operands: " + conv + "; "
            + "; ".join(lin) + "; each kernel held by tests to <= 2e-6 ... 1e-5 of the output scale against float64 — not narrower "
            "than an fp32 GEMM in effect; the position MLP of K21 and the 1-layer stem on the fp32 matrix pipe)")


def peer_access_matrix():
    """hipDeviceCanAccessPeer over the devices this process sees (the first multi-GPU record explains itself)."""
    n = torch.cuda.device_count()
    return [[int(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(n)] for i in range(n)]


def step(model, inp, hot_path_only=False, nxt=None):
    """One forward of the detector over one batch: FSF.simple_test (segmentation + image fusion, camera queries, LiDAR
    queries, query refinement, box decoding + NMS, results to the host) — or only its three query-generation stages.
    `nxt`: the NEXT step's batch, announced like a test loop's data loader would (FSF.set_next_frame): its host-bound front
    (voxelization, VFE, first index plan) is issued on a side stream while this step's host thread waits for its results."""
    with torch.no_grad():
        if nxt is not None:
            model.set_next_frame(nxt["points"], nxt["img_metas"], nxt["mask_data"], nxt["mask_anno"], ready=nxt.get("ready"))
        if hot_path_only:
            return model.forward_hot_path(inp["points"], inp["img_metas"], inp["mask_data"], inp["mask_anno"])
        return model.simple_test(inp["points"], inp["img_metas"], inp["mask_data"], inp["mask_anno"])


def count_launch_sources(model, inp, hot_path_only=False):
    """How many things one frame asks the device / the host for, counted in ONE extra untimed frame: C-ABI calls
    of libfsf_hip (each 1-13 kernel launches; the stage drivers K30 / K31 more), ATen ops that do device work (non-view ops seen by a TorchDispatchMode, ~1 launch each)
    and host synchronisations the Python side can see (`.item()` / `bool()` / `int()` of a device tensor, `nonzero`, boolean-mask
    indexing, `torch.cuda.synchronize`, and the C-ABI calls that read a count back: fsf_unique_rows, fsf_rulebook_strided, fsf_cluster_key_survival).  The exact
    kernel-launch count needs a trace: `kernel_launches_per_frame_rocprof` is read from the newest committed
    profiles/*_kernel_stats_full_forward.txt."""
    import collections
    import re

    from torch.utils._python_dispatch import TorchDispatchMode

    from fullysparsefusion_amd import _lib, hip_ops

    counts = collections.Counter()
    view = {"view", "slice", "select", "unsqueeze", "squeeze", "expand", "t", "transpose", "permute", "alias", "detach", "as_strided",
            "_unsafe_view", "reshape", "empty", "empty_strided", "size", "stride", "lift_fresh", "unbind", "split", "narrow", "new_empty",
            "empty_like", "unfold", "sym_size", "sym_stride", "sym_numel", "is_pinned", "split_with_sizes", "diagonal", "resize_",
            "record_stream"}

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func).replace("aten.", "").split(".")[0]
            if name not in view:
                counts["aten"] += 1
            a0 = args[0] if args else None
            if name in ("nonzero", "masked_select") or (name == "_local_scalar_dense" and torch.is_tensor(a0) and a0.is_cuda):
                counts["sync"] += 1  # (an .item() on a HOST tensor — the box tail reads its one read-back that way — waits for nothing)
            elif name in ("_to_copy", "copy_"):  # a device -> host copy waits for the stream
                src = args[1] if name == "copy_" else a0
                dst = a0.device if name == "copy_" else (kwargs or {}).get("device", None)
                if torch.is_tensor(src) and src.is_cuda and dst is not None and torch.device(dst).type == "cpu":
                    counts["sync"] += 1
            return func(*args, **(kwargs or {}))

    orig_check, orig_sync = _lib.check, torch.cuda.synchronize
    HOST_WAITS = 2  # FSF_OPT_HOST_WAITS: the library's own count of its stream waits (every count / flag read-back)
    waits0 = int(_lib.lib().fsf_get_option(HOST_WAITS))

    def check(status, what):
        counts["cabi"] += 1
        return orig_check(status, what)

    def sync(*a, **k):
        counts["sync"] += 1
        return orig_sync(*a, **k)

    _lib.check = hip_ops.check = check
    torch.cuda.synchronize = sync
    was = model.test_cfg.get("concurrent_query_branches", None)
    model.test_cfg["concurrent_query_branches"] = False  # (a dispatch mode is per thread: count on one)
    try:
        with Spy():
            step(model, inp, hot_path_only)
    finally:
        _lib.check = hip_ops.check = orig_check
        torch.cuda.synchronize = orig_sync
        if was is None:
            model.test_cfg.pop("concurrent_query_branches", None)
        else:
            model.test_cfg["concurrent_query_branches"] = was
    counts["sync"] += int(_lib.lib().fsf_get_option(HOST_WAITS)) - waits0
    orig_sync()
    traced = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir) if os.path.isdir(pdir) else [], reverse=True):
        if name.endswith("_kernel_stats_full_forward.txt"):
            m = re.search(r"(\d+) launches/frame", open(os.path.join(pdir, name)).read(1500))
            if m:
                traced = dict(value=int(m.group(1)), source=f"profiles/{name}")
                break
    return dict(c_abi_calls_per_frame=counts["cabi"], aten_device_ops_per_frame=counts["aten"], host_syncs_per_frame=counts["sync"],
                kernel_launches_per_frame_rocprof=traced,
                note="counted in one extra untimed frame with both query branches on one host thread; a C-ABI call is 1-13 launches (a stage driver — K30, K31 — more)")


def describe_output(model, inp, out, args):
    """Query / box counts of the timed workload (one extra untimed pass when the timed output does not carry them)."""
    if args.no_describe:
        return {}
    if not (args.train or args.hot_path_only):
        hot = step(model, inp, hot_path_only=True)
        extra = {"boxes_out": int(sum(len(r["boxes_3d"]) for r in out))}
    else:
        hot, extra = out, {}
    return dict(camera_queries=int(hot["frustum_obj_feats"].shape[0]), lidar_queries=int(hot["fsd_obj_feats"].shape[0]), **extra)


def calibrate_trained_like(model, inp, fg_fraction=0.05, nms_candidates=600):
    """Random-init weights make 91 % of the points foreground and every query a box candidate in every class — conservative, but
    not the stage mix of a deployed model.  This shifts TWO biases of the (otherwise unchanged) random-init model, calibrated on
    one frame, so that the reference's own thresholds cut like they do with trained weights:
      * the background logit of the segmentation head (`VoteSegHead.conv_seg`), until the share of points whose class-group score
        passes `score_thresh` (FSF_nuScenes_config.py:22-30: 0.1 per group) is `fg_fraction`;
      * the class-score branch of the refined head (`FSDSeparateHead.score`), until `nms_candidates` (query, class) pairs pass
        `score_thr` (0.01, FSF_nuScenes_config.py:377-384) and enter the per-class NMS.
    The foreground points are still scattered (random logits have no spatial structure), so the LiDAR-query clustering sees many
    small clusters; the camera branch gets its realism from the frame (40 instances instead of 250).  Returns what was achieved."""
    cfg = model.cfg
    names = list(cfg["class_names"])
    groups = [[names.index(n) for n in g] for g in cfg["group_names"]]
    thr = cfg["score_thresh"]
    head = model.segmentor.segmentation_head
    with torch.no_grad():
        lg = model.forward_hot_path(inp["points"], inp["img_metas"], inp["mask_data"], inp["mask_anno"])["seg"]["seg_logits"].float()

        def frac(delta):
            z = lg.clone()
            z[:, -1] += delta
            p = z.softmax(1)
            fg = torch.zeros(lg.size(0), dtype=torch.bool, device=lg.device)
            for gi, idx in enumerate(groups):
                fg |= p[:, idx].sum(1) > thr[gi]
            return float(fg.float().mean())

        lo, hi = 0.0, 40.0
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            lo, hi = (mid, hi) if frac(mid) > fg_fraction else (lo, mid)
        head.conv_seg.bias[-1] += hi
        got_fg = frac(hi)
        # refined head: capture its class logits on this frame
        rh = model.frustum_refined_head[0]
        cap = {}
        orig = rh.forward

        def tap(*a, **k):
            out = orig(*a, **k)
            cap["cls"] = out["cls_logits"][0]
            return out

        rh.forward = tap
        try:
            model.simple_test(inp["points"], inp["img_metas"], inp["mask_data"], inp["mask_anno"])
        finally:
            rh.forward = orig
        cls = cap["cls"].float().flatten()
        k = min(nms_candidates, cls.numel())
        kth = float(torch.topk(cls, k).values[-1])
        logit_thr = float(np.log(rh.test_cfg["score_thr"] / (1.0 - rh.test_cfg["score_thr"])))
        score_mlp = rh.task_heads[0].score
        last = [m for m in score_mlp.modules() if isinstance(m, torch.nn.Linear)][-1]
        last.bias += (logit_thr - kth)
    return dict(fg_fraction_target=fg_fraction, fg_fraction_points=round(got_fg, 4), seg_background_bias_shift=round(hi, 3),
                nms_candidates_target=int(k), refined_score_bias_shift=round(logit_thr - kth, 3), queries=int(cap["cls"].shape[0]))


def head_outputs(out):
    """Every tensor the reference's losses consume, in graph order: segmentation logits + vote predictions, then per query head
    (camera `frustum_obj_head`, LiDAR `bbox_head`, every refine stage's `frustum_refined_head`) the per-task class logits and box
    regressions (`FSF.forward_train`, FSF.py:806-903, :905-959)."""
    seg = out["seg"]
    ts = [seg["seg_logits"], seg["seg_vote_preds"]]
    results = [out.get("frustum_obj_result"), out.get("fsd_obj_result")] + list(out.get("stage_results", []))
    for res in results:
        if res is None:
            continue
        for key in ("cls_logits", "reg_preds", "iou_logits"):
            ts.extend(res.get(key, []))
    return ts


def dummy_loss(out):
    """'fwd+bwd with a dummy scalar loss (sum of head outputs)' — the loss / target path is not built, so
    every tensor the heads hand to a loss contributes.  An `out` of `forward_hot_path` (no heads run) falls back to the query features."""
    if out.get("frustum_obj_result") is None:
        seg = out["seg"]
        return (seg["seg_logits"].sum() + seg["seg_vote_preds"].sum() + out["frustum_obj_feats"].sum()
                + out["fsd_obj_feats"].sum()) * 1e-6
    return sum(t.sum() for t in head_outputs(out)) * 1e-6


class TrainStep:
    """fwd of `FSF.forward_train_graph` (training-mode norms; every module `FSF.forward_train` runs) + bwd of the sum of all head
    outputs + gradient all-reduce (FrameDataParallel buckets, overlapped with the backward pass) + AdamW."""

    def __init__(self, model, hot_path_only=False):
        from fullysparsefusion_amd.data_parallel import FrameDataParallel

        self.hot_path_only = hot_path_only  # (round <= 4's graph: the loss taken at the query features, heads / refine stage not run)
        self.model = model.train()
        self.dp = FrameDataParallel(model)
        params = [p for p in model.parameters() if p.requires_grad]
        try:  # one multi-tensor kernel per parameter chunk instead of the ~40 foreach launches (2.5 -> 0.5 ms per step)
            self.opt = torch.optim.AdamW(params, lr=1e-6, weight_decay=0.01, fused=all(p.is_cuda for p in params))
        except (RuntimeError, TypeError):
            self.opt = torch.optim.AdamW(params, lr=1e-6, weight_decay=0.01)

    def __call__(self, inp):
        self.dp.zero_grad()
        fwd = self.model.forward_hot_path if self.hot_path_only else self.model.forward_train_graph
        out = fwd(inp["points"], inp["img_metas"], inp["mask_data"], inp["mask_anno"])
        self.dp.backward(dummy_loss(out))
        self.opt.step()
        return out


# ------------------------------------------------------------------------------------------------ instrumented pass
class _Probe:
    """Wraps entry points of `hip_ops`.  mode "events": HIP events around every call, recorded on the calling thread's
    current stream (the stream the C ABI launches on) from a pre-created pool — right for kernels of 100+ us issued back to
    back while the device is busy (the sparse convolutions of the U-Net).  mode "capture": only remembers (function,
    arguments); `replay()` then times every remembered call in isolation, REP launches back to back between one event pair,
    so that the host's launch latency (which an event pair around a single short launch on a drained stream would
    measure instead of the kernel) is hidden behind the previous launch."""

    REP = 4

    def __init__(self, mode):
        self.mode = mode
        self.records = []
        self.lock = threading.Lock()
        self.saved = {}
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(2048)] if mode == "events" else []

    def wrap(self, mod, name, account):
        orig = getattr(mod, name)
        self.saved[(mod, name)] = orig

        def wrapped(*a, **k):
            if self.mode == "capture":
                out = orig(*a, **k)
                with self.lock:
                    self.records.append((name, orig, account, a, k, out))
                return out
            with self.lock:
                e0 = self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)
                e1 = self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **k)
            e1.record()
            with self.lock:
                self.records.append((name, e0, e1, account, a, k, out))
            return out

        setattr(mod, name, wrapped)

    def restore(self):
        for (mod, name), orig in self.saved.items():
            setattr(mod, name, orig)

    def table(self):
        """{key: dict(calls, ms, bytes, flops)} — after a device sync."""
        agg = {}
        if self.mode == "capture":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for name, orig, account, a, k, out in self.records:
                key, byts, flops = account(a, k, out)
                orig(*a, **k)  # warm (allocator, workspaces)
                e0.record()
                for _ in range(self.REP):
                    orig(*a, **k)
                e1.record()
                e1.synchronize()
                d = agg.setdefault(key, dict(calls=0, ms=0.0, bytes=0.0, flops=0.0))
                d["calls"] += 1
                d["ms"] += e0.elapsed_time(e1) / self.REP
                d["bytes"] += byts
                d["flops"] += flops
            return agg
        for name, e0, e1, account, a, k, out in self.records:
            key, byts, flops = account(a, k, out)
            d = agg.setdefault(key, dict(calls=0, ms=0.0, bytes=0.0, flops=0.0))
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["bytes"] += byts
            d["flops"] += flops
        return agg


def _n(t):
    return 0 if t is None else int(t.size(0))


def _acc_spconv(kind):
    def account(a, k, out):
        if kind == "fp32":
            feat, weight_t, nbr = a[0], a[1], a[2]
            cin, cout = weight_t.size(2), weight_t.size(1)
        elif kind == "planes":  # (sources, wplanes, kvol, cout, nbr)
            cout, nbr = int(a[3]), a[4]
            cin = sum(p.c for p in a[0])
        else:  # K9b: (feat, planes, kvol, cout, nbr)
            feat, cout, nbr = a[0], int(a[3]), a[4]
            cin = feat.size(1)
        pairs = float((nbr >= 0).sum())
        return ("spconv_" + kind, pairs * (cin + cout) * 4 + nbr.size(1) * cin * cout * 4 + 8 * pairs, 2.0 * pairs * cin * cout)
    return account


def _acc_bwd_weight(a, k, out):
    feat, grad_out, num = a[0], a[1], a[3]   # hip_ops.spconv_backward_weight(feat, grad_out, pairs, num)
    pairs = float(num.sum())
    cin, cout = feat.size(1), grad_out.size(1)
    # (the library runs 128 x 128 tiles on K10p — bf16 MFMA x 6 — and 64-wide ones on K10 — fp32 MFMA —: csrc/spconv_bwd.hip::bwd_plan)
    split = cin > 64 and cout > 64 and os.environ.get("FSF_BWD_SPLIT", "1") != "0"
    return ("spconv_bwd_weight_bf16x6" if split else "spconv_bwd_weight"), pairs * (cin + cout) * 4 + 8 * pairs + num.numel() * cin * cout * 4, \
        2.0 * pairs * cin * cout


def train_extras(train_step, pool, steps, world, dist, device):
    """What a `--train` line says beyond frames/s (every rank takes part: the collectives are real).
      * `roofline`: the step's dominant kernel — K10, `fsf::spconv_bwd_weight_kernel`, on the fp32 matrix pipe — timed with HIP
        events in situ over `n` instrumented steps: flops 2 P Cin Cout of the pair lists it received / time, against 157.3 TFLOP/s;
        the K9 forward / data-gradient launches of the same steps beside it;
      * `allreduce`: the step timed again under `no_sync()` (no collective at all) — the difference to the timed step is the all-reduce
        time the backward pass did NOT hide; the buckets' all-reduces timed in isolation (back to back, nothing else running) are the
        collective's full cost; overlap = 1 - exposed / isolated."""
    from fullysparsefusion_amd import hip_ops

    nframes = len(pool)
    n = max(2, min(steps, 4))
    p = _Probe("events")
    p.wrap(hip_ops, "spconv_backward_weight", _acc_bwd_weight)
    p.wrap(hip_ops, "spconv_forward_planes", _acc_spconv("planes"))
    p.wrap(hip_ops, "spconv_forward_split", _acc_spconv("split"))
    p.wrap(hip_ops, "spconv_forward", _acc_spconv("fp32"))
    try:
        for i in range(n):
            train_step(pool[i % nframes])
        torch.cuda.synchronize()
    finally:
        p.restore()
    t = p.table()
    kern = {}
    peaks = {"spconv_bwd_weight": MFMA_F32_PEAK_TFLOPS, "spconv_bwd_weight_bf16x6": MFMA_16BIT_PEAK_TFLOPS / 6,
             "spconv_planes": MFMA_16BIT_PEAK_TFLOPS / 3, "spconv_split": MFMA_16BIT_PEAK_TFLOPS / 6, "spconv_fp32": MFMA_F32_PEAK_TFLOPS}
    for key, d in t.items():
        tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        kern[key] = dict(launches_per_step=round(d["calls"] / n, 1), ms_per_step=round(d["ms"] / n, 3), tflops_fp32_equivalent=round(tf, 2),
                         pipe_peak_tflops_fp32_equivalent=round(peaks[key], 1), frac_of_pipe_peak=round(tf / peaks[key], 4),
                         algorithmic_gflop_per_step=round(d["flops"] / n / 1e9, 1))
    dom_key = "spconv_bwd_weight_bf16x6" if kern.get("spconv_bwd_weight_bf16x6", {}).get("ms_per_step", 0.0) >= \
        kern.get("spconv_bwd_weight", {}).get("ms_per_step", 0.0) and "spconv_bwd_weight_bf16x6" in kern else "spconv_bwd_weight"
    k10 = kern.get(dom_key, dict(tflops_fp32_equivalent=0.0, frac_of_pipe_peak=0.0))
    title = ("fsf::spconv_bwd_weight_split_kernel (K10p: weight gradient over the spconv-v1 pair lists, exact 3-way bf16 split of both operands, "
             "v_mfma_f32_16x16x32_bf16 x 6 per fp32-equivalent product)" if dom_key.endswith("bf16x6") else
             "fsf::spconv_bwd_weight_kernel (K10: weight gradient over the spconv-v1 pair lists, fp32 matrix pipe)")
    roof = dict(bound="mfma", kernel=title,
                achieved=k10["tflops_fp32_equivalent"], peak=round(peaks[dom_key], 1), unit="TFLOP/s", frac=k10["frac_of_pipe_peak"], traffic=None,
                ms_per_step_in_kernel=k10.get("ms_per_step"), launches_per_step=k10.get("launches_per_step"),
                kernels=kern, note="HIP events in situ on the launching stream over %d instrumented training steps; forward / data-gradient "
                                   "launches (K9c/K9d, K9b, fp32 kernel) of the same steps listed beside the dominant kernel" % n)
    # --- the collective: exposed vs isolated
    dp = train_step.dp

    def timed(k, sync):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(k):
            if sync:
                train_step(pool[i % nframes])
            else:
                with dp.no_sync():
                    train_step(pool[i % nframes])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3

    k = max(2, min(steps, 4))
    t_sync, t_local = timed(k, True), timed(k, False)
    iso = 0.0
    if dist is not None and world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            for b in dp.buckets:
                dist.all_reduce(b.flat, group=dp.group)
        torch.cuda.synchronize()
        iso = (time.perf_counter() - t0) / 3 * 1e3
        dp.zero_grad()
    # --- naiveSyncBN1d's statistics collectives (C3): 2 x (number of synced norms) latency-bound [2C] all-reduces per step, on the
    # critical path of both passes.  Counted from the module's own log; exposed = the step with them skipped on every rank alike
    # (statistics per rank: a timing experiment) against the full step; isolated = the same message sizes back to back.
    from fullysparsefusion_amd.mmdet3d_plugin.ops import norm as norm_mod

    sync_sizes, t_nobn, bn_iso = [], t_sync, 0.0
    if dist is not None and world > 1:
        norm_mod.SYNC_LOG = log = []
        train_step(pool[0])
        torch.cuda.synchronize()
        norm_mod.SYNC_LOG = None
        sync_sizes = [numel for _, numel in log]
        norm_mod.SYNC_COLLECTIVES = False
        try:
            t_nobn = timed(k, True)
        finally:
            norm_mod.SYNC_COLLECTIVES = True
        msgs = [torch.zeros(numel, dtype=torch.float32, device=device) for numel in sync_sizes]
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            for m in msgs:
                dist.all_reduce(m)
        torch.cuda.synchronize()
        bn_iso = (time.perf_counter() - t0) / 3 * 1e3
    vals = torch.tensor([t_sync, t_local, iso, t_nobn, bn_iso], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
    t_sync, t_local, iso, t_nobn, bn_iso = [float(v) for v in vals]
    exposed = max(0.0, t_sync - t_local)
    nbytes = sum(b.flat.numel() for b in dp.buckets) * 4
    allreduce = dict(world_size=world, gradient_bytes=nbytes, buckets=len(dp.buckets), step_ms=round(t_sync, 3), step_ms_no_sync=round(t_local, 3),
                     exposed_ms=round(exposed, 3), isolated_ms=round(iso, 3),
                     overlap_fraction=(round(1.0 - min(1.0, exposed / iso), 3) if iso > 0 else None),
                     bus_gb_per_s_isolated=(round(2.0 * (world - 1) / world * nbytes / (iso * 1e-3) / 1e9, 1) if iso > 0 else None),
                     note="max over ranks; world size 1 issues no collective (exposed = run-to-run noise)",
                     syncbn=dict(collectives_per_step=len(sync_sizes), floats_per_step=int(sum(sync_sizes)),
                                 step_ms_without_them=round(t_nobn, 3), exposed_ms=round(max(0.0, t_sync - t_nobn), 3),
                                 isolated_ms=round(bn_iso, 3),
                                 note="naiveSyncBN1d [2C] statistics all-reduces (forward + backward of every synced norm): exposed = "
                                      "full step - the step with these collectives skipped on every rank; isolated = the same messages "
                                      "back to back; 0 at world size 1"))
    return roof, allreduce


def _acc_seg_reduce(a, k, out):
    feat, plan = a[0], a[1]
    n, c = feat.shape
    return "seg_reduce", 4.0 * c * n + 8.0 * n + 4.0 * c * plan.m, 0.0       # 4C B/row + 8 B/row + 4C B/segment


def _acc_seg_reduce_short(a, k, out):
    feats, plan = a[0], a[1]
    n = feats[0].size(0)
    c = sum(int(f.size(1)) for f in feats)
    return "seg_reduce", 4.0 * c * n + 8.0 * n + 4.0 * c * plan.m, 0.0      # (the short-segment form: same bytes, up to 8 tensors per launch)


def _acc_gather_rows(a, k, out):
    src, idx = a[0], a[1]
    return "gather_rows", idx.numel() * (8.0 + 8.0 * src.size(1)), 0.0         # 8 B index + 4C read + 4C written per row


def _acc_voxel2point(a, k, out):
    points, voxel_feats = a[0], a[2]
    n, c = points.size(0), voxel_feats.size(1)
    return "voxel2point", n * (12.0 + 8.0 + 32.0 + 4.0 * c + 4.0 * (c + 3) + 1.0), 0.0


def _acc_project(a, k, out):
    xyz, mask = a[0], a[2]
    n, (ncam, ncls) = xyz.size(0), mask.shape[:2]
    # 12 B/pt of xyz + one mask element per (cam, class) + the int64 id tensor (552 B/pt at 6 x 10, u8 planes)
    return "project_gather", n * (12.0 + ncam * ncls * (mask.element_size() + 8.0)), 0.0


def _acc_cam_select(a, k, out):
    obj = a[0]
    n, ncam, ncls = obj.shape
    return "cam_select_score", n * (8.0 * ncam * ncls + 4.0 * ncls + (8.0 * ncls if k.get("return_ids") else 0.0)), 0.0


def _acc_project_score(a, k, out):
    xyz, mask = a[0], a[2]
    n, (ncam, ncls) = xyz.size(0), mask.shape[:2]
    # 12 B/pt of xyz + one mask element per (cam, class) read; the f32 score row (+ i64 ids when asked for) + the flag written
    ids = 8.0 * ncls if k.get("return_ids") else 0.0
    return "project_score", n * (12.0 + ncam * ncls * mask.element_size() + 4.0 * ncls + ids + 1.0), 0.0


def _acc_sir_input(a, k, out):
    points, feats, f_cluster = a[0], a[1], a[2]
    extra = k.get("extra", a[7] if len(a) > 7 else None)
    n = points.size(0)
    fcols = sum(t.size(1) for t in feats) if isinstance(feats, (list, tuple)) else feats.size(1)  # (parts read through an index)
    c = points.size(1) + fcols + (extra.size(1) if extra is not None else 0)
    return "sir_input", n * 4.0 * (c + f_cluster.size(1)) + n * 4.0 * c, 0.0   # 4(P+Cf+Ce+R) read + 4C written per row


def _acc_linear(a, k, out):
    x, c = a[0], int(a[2])
    n, kk = x.shape
    grouped = k.get("row_add") is not None
    return ("linear_norm_act", n * 4.0 * (kk + c) + (n * (8.0 + 4.0 * c) if grouped else 0.0), 2.0 * n * kk * c)


def _acc_linear_segmax(a, k, out):
    """K22s: the rows of K22 (x in; rows out unless `want_rows=False`) + the segmented max it fuses (8 B of segment id per row,
    4 c B per segment written) — SURVEY 8(d)'s per-unit figures of the two kernels it replaces, minus the re-read of the rows."""
    x, c, seg_out = a[0], int(a[2]), a[4]
    n, kk = x.shape
    m = seg_out.size(0)
    grouped = k.get("row_add") is not None
    rows_out = 4.0 * c if k.get("want_rows", True) else 0.0
    return ("linear_norm_act_segmax", n * (4.0 * kk + rows_out + 8.0) + m * 4.0 * c + (n * 4.0 * c if grouped else 0.0), 2.0 * n * kk * c)


def _acc_linear_sliced(a, k, out):
    x, kk, nslice, slice_c = a[0], int(a[1]), int(a[4]), int(a[5])
    n = x.size(0)
    reads = kk * (nslice if int(a[2]) > 0 else 1)  # distinct input columns (a shared input is read once from HBM)
    return ("linear_norm_act", n * 4.0 * (reads + nslice * slice_c), 2.0 * n * kk * nslice * slice_c)


def _acc_linear_planes(a, k, out):
    """K22h: x planes in (4 K B/row + 4 B scale), f32 rows out — the K22 family's bytes and fp32-equivalent flops."""
    xp, c = a[0], int(a[2])
    return ("linear_norm_act", xp.n * (4.0 * xp.c + 4.0 + 4.0 * c), 2.0 * xp.n * xp.c * c)


def _acc_rows_to_planes(a, k, out):
    x = a[0]
    n, c = x.shape
    return "rows_to_planes", n * (8.0 * c + 4.0 + (4.0 * c if k.get("want_rows") else 0.0)), 0.0  # 4C read, 4C planes (+ 4C rows) written


def instrumented_pass(model, pool, steps, hot_path_only):
    """Two passes over the same frames: the sparse-conv forward kernels with HIP events in situ (`steps` frames); the
    scatter / gather / segmented-reduce / projection / fused-linear kernels captured on ONE frame and replayed in isolation.
    Returns ({key: totals}, frames the conv totals cover, frames the hbm totals cover)."""
    from fullysparsefusion_amd import hip_ops

    p = _Probe("events")
    p.wrap(hip_ops, "spconv_forward", _acc_spconv("fp32"))
    p.wrap(hip_ops, "spconv_forward_split", _acc_spconv("split"))
    if hasattr(hip_ops, "spconv_forward_planes"):
        p.wrap(hip_ops, "spconv_forward_planes", _acc_spconv("planes"))
    # The conv launches are timed with the U-Net's side stream off (FSF_UNET_LATERAL_STREAM=0): with it, the fine lateral blocks run
    # beside the small deep levels and an event pair on one stream brackets time its kernel shares with the other stream's —
    # the frame gets shorter while every overlapped launch looks longer.  (profiles/*_kernel_stats_full_forward_serial_unet.txt is
    # the rocprofv3 summary of the same setting.)
    from fullysparsefusion_amd import switches

    prev, switches.UNET_LATERAL_STREAM = switches.UNET_LATERAL_STREAM, False
    try:
        for i in range(steps):
            step(model, pool[i % len(pool)], hot_path_only)
        torch.cuda.synchronize()
    finally:
        p.restore()
        switches.UNET_LATERAL_STREAM = prev
    conv = p.table()
    q = _Probe("capture")
    q.wrap(hip_ops, "segment_reduce", _acc_seg_reduce)
    q.wrap(hip_ops, "segment_reduce_short", _acc_seg_reduce_short)
    q.wrap(hip_ops, "gather_rows", _acc_gather_rows)
    q.wrap(hip_ops, "voxel2point", _acc_voxel2point)
    q.wrap(hip_ops, "project_gather_mask", _acc_project)
    q.wrap(hip_ops, "cam_select_score", _acc_cam_select)
    q.wrap(hip_ops, "sir_input", _acc_sir_input)
    q.wrap(hip_ops, "linear_norm_act", _acc_linear)
    q.wrap(hip_ops, "linear_norm_act_sliced", _acc_linear_sliced)
    q.wrap(hip_ops, "linear_norm_act_segmax", _acc_linear_segmax)
    q.wrap(hip_ops, "linear_planes_norm_act", _acc_linear_planes)
    q.wrap(hip_ops, "rows_to_planes", _acc_rows_to_planes)
    if hasattr(hip_ops, "project_score"):
        q.wrap(hip_ops, "project_score", _acc_project_score)
    # (the capture counts the calls of hip_ops' wrappers: for this one untimed frame the SIR stacks run kernel by kernel from Python
    # instead of through fsf_sir_stack_forward, K31 — the same kernels with the same arguments, tests/test_sir_stack_gpu.py)
    stacks = [m for m in model.modules() if getattr(m, "native_stack", None) is True or type(m).__name__ == "FullySparseBboxHead"]
    for m in stacks:
        m.native_stack = False
    try:
        step(model, pool[0], hot_path_only)
        torch.cuda.synchronize()
    finally:
        q.restore()
        for m in stacks:
            m.native_stack = True
    hbm = q.table()
    return conv, hbm


SPCONV_KERNELS = {
    "spconv_fp32": ("fsf::spconv_fwd_dma_kernel (fp32 MFMA, compacting, output-stationary in LDS)", MFMA_F32_PEAK_TFLOPS,
                    "v_mfma_f32_16x16x4_f32"),
    "spconv_split": ("fsf::spconv_fwd_split_kernel (bf16 MFMA x6 = exact 3-way split, row-stationary in registers)",
                     MFMA_16BIT_PEAK_TFLOPS / 6, "v_mfma_f32_16x16x32_bf16, 6 per fp32-equivalent product"),
    "spconv_planes": ("fsf::spconv_fwd_pipe_kernel (K9d: f16 MFMA x3 = row-scaled 2-way f16 split, channel-stationary waves, cell skipping, "
                      "chunk-granular pipeline; layers it does not take run fsf::spconv_fwd_planes_kernel, K9c)",
                      MFMA_16BIT_PEAK_TFLOPS / 3, "v_mfma_f32_16x16x32_f16, 3 per fp32-equivalent product"),
}


def shape_ceiling(achieved_tflops, dom_key):
    """What the dominant kernel's SHAPE reaches with the sparse-convolution specifics removed (tools/profiling/k9d_shape_probe.hip: the
    same per-iteration loads, LDS round trip and MFMAs at the same occupancy, every cell live) — from the newest committed probe run.
    `frac` stays priced against the pipe's peak; this says how much of the gap is the shape's and how much the kernel's."""
    if dom_key != "spconv_planes":
        return None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir) if os.path.isdir(pdir) else [], reverse=True):
        if name.endswith("_k9d_shape_ceiling_probe.txt"):
            m = re.search(r"^full shape.*?=\s*([0-9.]+) fp32-eq = ([0-9.]+) of 833", open(os.path.join(pdir, name)).read(), flags=re.M)
            if m:
                tf = float(m.group(1))
                return dict(tflops_fp32_equivalent=tf, frac_of_pipe_peak=float(m.group(2)), kernel_frac_of_shape=round(achieved_tflops / tf, 4),
                            source=f"profiles/{name}",
                            note="gather -> LDS -> MFMA pipeline of K9d, three workgroups per CU, all cells live, no table / scales / epilogue")
    return None


def roofline_blocks(conv, hbm_table, steps, ms_per_step, traffic_of):
    """`roofline` (dominant kernel = the sparse-conv forward kernel with the most time), `hbm` and `frame_roofline_ms`."""
    kernels = {}
    for k, v in conv.items():
        title, peak, pipe = SPCONV_KERNELS[k]
        tf = v["flops"] / max(v["ms"], 1e-9) / 1e9
        kernels[title] = dict(launches_per_step=v["calls"] // steps, ms_per_step=round(v["ms"] / steps, 3),
                              tflops_fp32_equivalent=round(tf, 2), pipe=pipe, pipe_peak_tflops_fp32_equivalent=round(peak, 1),
                              frac_of_pipe_peak=round(tf / peak, 4),
                              algorithmic_gflop_per_step=round(v["flops"] / steps / 1e9, 1),
                              algorithmic_mb_per_launch=round(v["bytes"] / max(v["calls"], 1) / 1e6, 1))
    dom_key = max(conv, key=lambda k: conv[k]["ms"])
    dom = conv[dom_key]
    title, peak, pipe = SPCONV_KERNELS[dom_key]
    traffic = traffic_of(title.split(" ")[0]) if traffic_of else {}  # (the kernel's C++ name up to its template arguments)
    achieved = dom["flops"] / max(dom["ms"], 1e-9) / 1e9
    all_flops, all_ms = sum(v["flops"] for v in conv.values()), sum(v["ms"] for v in conv.values())
    hbm = {}
    hbm_floor_ms = 0.0
    for k, v in sorted(hbm_table.items()):  # one frame, every call replayed in isolation
        gbs = v["bytes"] / max(v["ms"], 1e-9) / 1e6
        hbm[k] = dict(calls_per_step=v["calls"], ms_per_step=round(v["ms"], 3),
                      algorithmic_mb_per_step=round(v["bytes"] / 1e6, 1), gb_per_s=round(gbs, 1),
                      frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 4))
        if k in ("linear_norm_act", "linear_norm_act_segmax"):
            hbm[k]["tflops_fp32_equivalent"] = round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)
        hbm_floor_ms += v["bytes"] / (HBM_PEAK_GBS * 1e6)
    # the same algorithmic bytes against the kernels' IN-SITU time (rocprofv3, everything serialised) where a committed trace has it:
    # the replay above runs each call four times back to back on warm caches and flatters
    situ, situ_src = committed_in_situ()
    merged = dict(hbm_table)
    if "linear_norm_act_segmax" in merged:  # one kernel family in the trace
        a, b = merged.pop("linear_norm_act_segmax"), merged.get("linear_norm_act", dict(bytes=0.0, ms=0.0, calls=0, flops=0.0))
        merged["linear_norm_act"] = {kk: a[kk] + b[kk] for kk in ("bytes", "ms", "calls", "flops")}
    in_situ = {}
    for k, v in sorted(merged.items()):
        if k in situ and situ[k] > 0:
            gbs = v["bytes"] / situ[k] / 1e6
            in_situ[k] = dict(ms_per_step=round(situ[k], 3), gb_per_s=round(gbs, 1), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 4))
    conv_floor_ms = sum(v["flops"] / steps / (SPCONV_KERNELS[k][1] * 1e9) for k, v in conv.items())
    frame_floor = conv_floor_ms + hbm_floor_ms
    roof = dict(
        bound="mfma", kernel=title, achieved=round(achieved, 3), peak=round(peak, 1), unit="TFLOP/s",
        frac=round(achieved / peak, 4),
        peak_note=f"fp32-equivalent flops (2 * pairs * Cin * Cout) against the ceiling of the pipe the kernel issues on ({pipe})",
        launches_per_step=dom["calls"] // steps, avg_launch_us=round(dom["ms"] / max(dom["calls"], 1) * 1e3, 2),
        ms_per_step_in_kernel=round(dom["ms"] / steps, 3),
        sparse_conv_all_kernels=dict(ms_per_step=round(all_ms / steps, 3), launches_per_step=sum(v["calls"] for v in conv.values()) // steps,
                                     tflops_fp32_equivalent=round(all_flops / max(all_ms, 1e-9) / 1e9, 2),
                                     algorithmic_gflop_per_step=round(all_flops / steps / 1e9, 1)),
        kernels=kernels, traffic=traffic.get("value"), traffic_unit=traffic.get("unit"), traffic_source=traffic.get("source"),
        traffic_kernel=traffic.get("kernel"),
        shape_ceiling=shape_ceiling(achieved, dom_key),
        hbm_in_situ=in_situ,
        debug=dict(hbm_isolated_replay=hbm,
                   note="cache-warm isolated replay of one frame's calls (4 launches back to back per call): an upper bound of what each "
                        "kernel reaches alone, NOT the in-situ rate — quote `hbm_in_situ`"),
        hbm_in_situ_note=(f"algorithmic bytes of this run / per-frame kernel time of the family in {situ_src} (rocprofv3 --kernel-trace of "
                          "`bench.py --serial`: branches and U-Net streams serialised, every launch of the frame, cold caches as in the frame)"
                          if situ_src else None),
        frame_roofline_ms=round(frame_floor, 3),
        frame_roofline_note="sum over the instrumented kernels of (fp32-equivalent conv flops / the issuing pipe's ceiling) + "
                            "(algorithmic bytes / 8 TB/s); kernels that are not instrumented (sorts, rulebooks, CCL, pooling, NMS, "
                            "glue) add nothing, so this is a LOWER bound of the frame's roofline time",
        frame_roofline_frac=round(frame_floor / ms_per_step, 4),
        note="sparse-conv kernels: HIP-event timing in situ on the launching stream, in an instrumented pass over the same "
             "frames right after the timed region")
    return roof


# kernels behind each entry of `roofline.hbm` (substrings of the names rocprofv3 reports)
HBM_KERNELS = {
    "seg_reduce": ("seg_reduce_kernel", "seg_fixup_long_kernel", "seg_fixup_kernel", "seg_short_kernel"),
    "sir_input": ("sir_input_kernel",),
    "linear_norm_act": ("linear_norm_act_kernel",),
    "gather_rows": ("gather_rows_kernel",),
    "rows_to_planes": ("rows_to_planes_kernel",),
    "voxel2point": ("voxel2point_kernel",),
    "project_gather": ("project_gather_kernel",),
    "project_score": ("project_score_kernel",),
}


def committed_in_situ():
    """Per-frame kernel time of the HBM-bound kernel families IN SITU: from the newest committed rocprofv3 summary of the forward with
    everything serialised (`bench.py --serial`: one stream, no overlapping kernels), profiles/*_kernel_stats_full_forward_serial.txt.
    {hbm key: ms per frame}, source file."""
    prof = os.path.join(ROOT, "profiles")
    names = sorted(n for n in (os.listdir(prof) if os.path.isdir(prof) else []) if n.endswith("_kernel_stats_full_forward_serial.txt"))
    if not names:
        return {}, None
    table = {}
    with open(os.path.join(prof, names[-1])) as f:
        for line in f:
            parts = line.split(None, 6)
            if line.startswith("#") or len(parts) < 7 or not parts[0].replace(".", "").isdigit():
                continue
            for key, subs in HBM_KERNELS.items():
                if any(sub in parts[6] for sub in subs):
                    table[key] = table.get(key, 0.0) + float(parts[1])
    return table, "profiles/" + names[-1]


def committed_traffic(kernel_prefix=None):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_pmc_traffic.json, newest round
    last): the per-kernel table's entries whose name starts with `kernel_prefix` (all template variants, launch-weighted), else
    the average over every sparse-conv launch."""
    prof = os.path.join(ROOT, "profiles")
    out = {}
    for name in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        if name.endswith("_pmc_traffic.json"):
            with open(os.path.join(prof, name)) as f:
                t = json.load(f)
            unit = t.get("unit", "HBM bytes per launch (FETCH_SIZE x correction + WRITE_SIZE, rocprofv3 --pmc, separate passes)")
            rows = [k for k in t.get("kernels", []) if kernel_prefix and k.get("kernel", "").startswith(kernel_prefix)]
            launches = sum(k.get("launches_per_step", 0.0) for k in rows)
            if launches > 0:
                v = sum(k.get("hbm_mb_per_step", 0.0) for k in rows) * 1e6 / launches
                out = dict(value=round(v, 1), source=f"profiles/{name}", unit=unit, kernel=kernel_prefix)
                continue
            v = t.get("spconv_forward", {}).get("hbm_bytes_per_api_launch")
            if v is not None:
                out = dict(value=v, source=f"profiles/{name}", unit=unit, kernel="all sparse-conv launches (average)")
    return out


# ---------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(model_cpu, sweeps):
    """The CPU oracle (a port of the reference path: torch.unique + scatter_reduce + spconv-v1 restatement + scipy CCL;
    torch_scatter / spconv / mmcv are not installable) on ONE FULL frame of the timed workload,
    stages 1-3 (segmentor + fusion, camera queries, LiDAR queries), after an untimed warm-up pass on a 1-sweep frame."""
    from fullysparsefusion_amd import synthetic
    from oracle import modules as omod

    def run(frame):
        pts8 = torch.from_numpy(frame["points"])
        mask, anno, L = torch.from_numpy(frame["mask_data"]), torch.from_numpy(frame["mask_anno"]), torch.from_numpy(frame["lidar2img"])
        t = [time.perf_counter()]
        with torch.no_grad():
            s1 = omod.fsf_stage1(model_cpu, pts8, mask, anno, L)
            t.append(time.perf_counter())
            omod.fsf_stage2(model_cpu, s1, anno, (900, 1600))
            t.append(time.perf_counter())
            omod.fsf_stage3(model_cpu, s1)
            t.append(time.perf_counter())
        return pts8.shape[0], [t[i + 1] - t[i] for i in range(3)]

    run(synthetic.make_frame(num_sweeps=1, seed=1))  # warm-up: thread pools, allocator, scipy imports
    n, (t1, t2, t3) = run(synthetic.make_frame(num_sweeps=sweeps, seed=0))
    total = t1 + t2 + t3
    return dict(value=round(1.0 / total, 5), unit="frames/s", cores=torch.get_num_threads(), nproc=os.cpu_count(), kind="port",
                samples=1, timing="1 untimed warm-up pass (1-sweep frame) + 1 timed pass; SURVEY 8(d)'s 3 + 10 passes would cost ~10 min of "
                                  "the driver's run",
                stage_seconds=dict(segmentor_fusion_seg_head=round(t1, 2), camera_queries=round(t2, 2), lidar_queries=round(t3, 2)),
                sample=f"ONE full {sweeps}-sweep frame ({n} points, the timed workload's frame 0) through oracle stages 1-3 in "
                       f"{total:.1f} s after a 1-sweep warm-up pass; the refine stage and NMS are not in the CPU sample, so this "
                       f"over-states the CPU rate of the full forward")


# ------------------------------------------------------------------------------------------------------ launching
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: one process per GPU under torch.distributed.run (RCCL)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU (or let bench.py launch them)")
    if world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=device)  # RCCL; barrier + max-reduce of the timing only

    model = build_model(device, args.dataset)
    if args.serial:
        from fullysparsefusion_amd import switches

        switches.UNET_LATERAL_STREAM = switches.UNET_PLAN_STREAM = False
        # (one stream means one stream: nothing announced, the early key / row work in its place — the same kernels, in line)
        args.no_frame_front = True
        model._pre_voxel_keys_early = model._camera_rows_early = lambda *a, **k: None
        for m in model.modules():
            if isinstance(getattr(m, "test_cfg", None), dict) or hasattr(getattr(m, "test_cfg", None), "get"):
                try:
                    m.test_cfg["concurrent_query_branches"] = False
                except TypeError:
                    pass
    model_cpu = (None if args.no_cpu_baseline or rank != 0 or world != 1 or args.train or args.dataset != "nuscenes"
                 else copy.deepcopy(model).cpu())
    nframes = max(1, args.frames)
    pool = [make_inputs(args.sweeps, seed=rank * 131 + j, device=device, frames=args.frames_per_gpu, dataset=args.dataset,
                        trained_like=args.trained_like)[1] for j in range(nframes)]
    tl_info = calibrate_trained_like(model, pool[0]) if args.trained_like else None
    if args.train:
        train_step = TrainStep(model, hot_path_only=args.hot_path_only)
        run = lambda i: train_step(pool[i % nframes])  # noqa: E731
    else:
        announce = not args.no_frame_front
        run = lambda i: step(model, pool[i % nframes], args.hot_path_only, pool[(i + 1) % nframes] if announce else None)  # noqa: E731

    for i in range(args.warmup):
        run(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = run(args.warmup + i)
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = [args.frames_per_gpu * args.steps / local_elapsed]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        g = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(g, torch.tensor([per_rank[0]], dtype=torch.float64, device=device))
        per_rank = [float(x.item()) for x in g]

    result = None
    last = pool[(args.warmup + args.steps - 1) % nframes]
    if rank == 0:
        n_pts = [int(p["points"][0].shape[0]) for p in pool]
        result = {
            "metric": ((f"frames/sec fwd+bwd+allreduce+AdamW nuScenes {args.sweeps}-sweep FSF (dummy loss)" if args.train
                        else f"frames/sec fwd nuScenes {args.sweeps}-sweep FSF" + (" (query-generation stages only)" if args.hot_path_only else ""))
                       if args.dataset == "nuscenes" else
                       "frames/sec " + ("fwd+bwd+allreduce+AdamW" if args.train else "fwd") + " Argoverse-2-shape FSF"),
            "value": round(world * args.frames_per_gpu * args.steps / elapsed, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            **({"commit": os.environ["FSF_COMMIT"]} if os.environ.get("FSF_COMMIT") else {}),
            "dtype": dtype_text(args.train),
            "data": "synthetic",
            "config": {
                "workload": ((f"fsf_nuscenes_{args.sweeps}sweep_" if args.dataset == "nuscenes" else "fsf_av2_long_range_") + (
                    ("train_step: fwd of the query-generation stages only + bwd of a dummy scalar loss + gradient all-reduce + AdamW"
                     if args.hot_path_only else
                     "train_step: fwd of FSF.forward_train's graph (segmentor + fusion, camera / LiDAR queries, both query heads, query "
                     "combination, refine stage with RoI pooling + SIR + refined head) + bwd of the sum of all head outputs + gradient "
                     "all-reduce + AdamW")
                    if args.train else "hot_path_fwd: stages 1-3 of FSF.simple_test only" if args.hot_path_only else
                    "simple_test: full forward = segmentor + image fusion, camera queries, LiDAR queries, heads, query "
                    "refinement (RoI point pooling + SIR), box decode + rotated BEV NMS, results to host") +
                    (" (BASELINE config 3 input" if args.dataset == "nuscenes" else " (BASELINE config 5 shape") +
                    ", random-init weights of the reference architecture)"),
                "points_per_frame": n_pts[0],
                "distinct_frames_rotated": nframes,
                "points_per_frame_all": n_pts,
                "frames_per_gpu_per_step": args.frames_per_gpu,
                "mask_data": "u8[1,6,10,900,1600]" if args.dataset == "nuscenes" else "i32[1,7,1,1550,2048]",
                **describe_output(model, last, out, args),
                "parallelism": (f"dp{world}: frame-level data parallel, bucketed gradient all-reduce over RCCL" if args.train
                                else f"replicas x{world} (frames independent, no data-path collective)"),
                "rccl_world_size": world,
                "rccl_backend": (f"nccl (RCCL {'.'.join(str(v) for v in torch.cuda.nccl.version())})" if dist is not None else
                                 "not initialised (1 rank)"),
                "visible_devices": torch.cuda.device_count(),
                "peer_access": peer_access_matrix(),
                "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                "per_rank_frames_per_s": [round(v, 3) for v in per_rank],
                **({"trained_like": tl_info} if tl_info is not None else {}),
            },
        }
        if args.trained_like:
            result["metric"] += " (trained-like variant: calibrated foreground / candidate counts, 40 masks per frame)"
    ms_plain = None
    if not args.train:  # (all ranks, outside the timed region) the same loop WITHOUT announcing the next frame: rounds 1-5's loop
        ms_plain = result["ms_per_step"] if rank == 0 else None
        if not args.no_frame_front:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step(model, pool[(args.warmup + i) % nframes], args.hot_path_only)
            torch.cuda.synchronize()
            ms_plain = (time.perf_counter() - t1) / args.steps * 1e3
            t1 = time.perf_counter()  # (and the announced loop once more, behind it: a process's first loop runs 0.2-0.5 ms slower
            for i in range(args.steps):  # than its later ones whatever it is, so the pair to compare is this one and the unannounced)
                run(args.warmup + i)
            torch.cuda.synchronize()
            ms_again = (time.perf_counter() - t1) / args.steps * 1e3
        if rank == 0:
            result["frame_front"] = {
                "announced": not args.no_frame_front, "ms_per_step_unannounced": round(ms_plain, 3),
                **({"ms_per_step_announced_repeat": round(ms_again, 3)} if not args.no_frame_front else {}),
                "note": "the timed loop announces step i + 1's batch before step i (FSF.set_next_frame — what a test loop's data loader "
                        "knows): that frame's host-bound front (point split, image-branch projection + score MLP, voxelization, voxel "
                        "unique + read-back, DynamicScatterVFE, the U-Net's row order / first index plans / input planes / first two encoder levels) is issued on a "
                        "side stream while step i's host thread would idle in the box tail's read-back; same kernels, same inputs, "
                        "bit-identical boxes (tests/test_frame_front_gpu.py); every step still runs every kernel of its frame inside "
                        "the timed region (the first timed step's front falls in the last warm-up step, the last timed step issues "
                        "the front of a frame beyond the region); `ms_per_step_unannounced` = the same loop, same process, nothing "
                        "announced (`--no-frame-front` times that loop as the headline), `ms_per_step_announced_repeat` = the announced "
                        "loop again behind it (the loops of a process get faster as it warms: compare these two, not the headline, with "
                        "the unannounced figure; interleaved rounds: tools/profiling/frame_front_ab.py, profiles/r6_frame_front_ab.txt)"}
    if args.train and not args.no_roofline:  # (all ranks: the measurement runs real collectives)
        roof, allreduce = train_extras(train_step, pool, args.steps, world, dist, device)
        if rank == 0:
            result["roofline"], result["allreduce"] = roof, allreduce
    if rank == 0 and not args.no_roofline and not args.train:
        result["launches"] = count_launch_sources(model, last, args.hot_path_only)
        n = min(args.steps, 2 * nframes)
        conv_t, hbm_t = instrumented_pass(model, pool, n, args.hot_path_only)
        result["roofline"] = roofline_blocks(conv_t, hbm_t, n, result["ms_per_step"],
                                             committed_traffic if args.dataset == "nuscenes" else None)
        if not args.hot_path_only:  # where the frame time goes: the three query-generation stages vs the rest
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                step(model, pool[i % nframes], hot_path_only=True)
            torch.cuda.synchronize()
            result["stages"] = {"query_generation_ms": round((time.perf_counter() - t0) / n * 1e3, 3),
                                "full_forward_ms": result["ms_per_step"]}
    if (rank == 0 and world == 1 and not (args.train or args.hot_path_only or args.trained_like or args.no_h2d)
            and args.dataset == "nuscenes" and args.frames_per_gpu == 1):
        # BESIDE the headline: the same forward with every frame handed over as host buffers (PCIe-inclusive rate)
        result["h2d"] = h2d_inclusive(model, args.sweeps, [rank * 131 + j for j in range(nframes)], device, args.steps, args.warmup,
                                      ms_plain, announce=False)
        # (the host-buffer loop announces nothing: with the next frame announced it ran 13.45 ms against 13.11 unannounced on the same
        # box — the upload of frame i + 2, issued when frame i ends, then runs beside frame i + 1's front instead of beside its tail;
        # cause not established further.  Compared with the unannounced resident loop of the same run.)
    if (rank == 0 and world == 1 and not (args.train or args.hot_path_only or args.trained_like or args.no_trained_like)
            and args.dataset == "nuscenes" and args.frames_per_gpu == 1):
        # BESIDE the headline: the same forward on the trained-like variant (see calibrate_trained_like)
        m2 = build_model(device, args.dataset)  # (same seed -> the same random-init weights; the timed model stays untouched)
        pool2 = [make_inputs(args.sweeps, seed=977 + j, device=device, trained_like=True)[1] for j in range(2)]
        info = calibrate_trained_like(m2, pool2[0])
        for i in range(3):
            out2 = step(m2, pool2[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k2 = 10
        for i in range(k2):
            out2 = step(m2, pool2[i % 2], nxt=None if args.no_frame_front else pool2[(i + 1) % 2])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k2
        hot2 = step(m2, pool2[(k2 - 1) % 2], hot_path_only=True)
        result["trained_like"] = dict(
            value=round(1.0 / dt, 3), unit="frames/s", ms_per_step=round(dt * 1e3, 3), steps=k2,
            note="reported beside the headline, never instead of it: same model and kernels, 40 small 2-D instances per frame, two bias "
                 "shifts calibrated on one frame so that the config's own thresholds keep ~5 % of the points and a few hundred box "
                 "candidates (bench.py::calibrate_trained_like)",
            camera_queries=int(hot2["frustum_obj_feats"].shape[0]), lidar_queries=int(hot2["fsd_obj_feats"].shape[0]),
            boxes_out=int(sum(len(r["boxes_3d"]) for r in out2)), **info)
        del m2, pool2
    if (rank == 0 and world == 1 and not (args.train or args.hot_path_only or args.trained_like or args.no_train_block or args.serial)
            and args.dataset == "nuscenes" and args.frames_per_gpu == 1):
        # BESIDE the headline: BASELINE config 3 is "fwd+bwd" — the training step (`bench.py --train` times it as its own line) for a
        # few steps, so that the driver's default run sees it too
        torch.cuda.empty_cache()
        m3 = build_model(device, args.dataset)
        ts = TrainStep(m3)
        k3, w3 = 5, 2
        for i in range(w3):
            ts(pool[i % nframes])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k3):
            ts(pool[(w3 + i) % nframes])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k3
        roof3, _ = train_extras(ts, pool, 2, 1, None, device)
        result["train"] = dict(
            value=round(1.0 / dt, 3), unit="frames/s", ms_per_step=round(dt * 1e3, 3), steps=k3, warmup=w3, frames_per_gpu=1,
            workload="fwd of FSF.forward_train's graph in training mode (query heads, query combination and the refine stage included) + bwd "
                     "of the sum of all head outputs + (1-rank) gradient bucket pass + fused AdamW: the data-parallel mechanics of BASELINE "
                     "config 3; losses / target assignment are out of scope",
            roofline=dict(kernel=roof3.get("kernel"), bound=roof3.get("bound"), achieved=roof3.get("achieved"), peak=roof3.get("peak"),
                          unit=roof3.get("unit"), frac=roof3.get("frac"), ms_per_step_in_kernel=roof3.get("ms_per_step_in_kernel"),
                          launches_per_step=roof3.get("launches_per_step")))
        del m3, ts
        torch.cuda.empty_cache()
    if rank == 0 and model_cpu is not None:
        result["cpu_baseline"] = cpu_baseline(model_cpu, args.sweeps)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
