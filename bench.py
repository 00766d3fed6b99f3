#!/usr/bin/env python
"""bench.py — frames/sec of the FSF forward on synthetic nuScenes-shape 10-sweep frames (BASELINE.json metric), one
process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full forward of the detector (FSF.simple_test: voxelize -> DynamicScatterVFE -> SimpleSparseUNet ->
neck -> projection + mask gather + image fusion + seg head -> camera-query grouping + SIR -> LiDAR-query pre-voxelize,
sampling, device CCL, SIR -> heads -> query combination -> RoI point pooling + refine SIR -> box decode + rotated BEV
NMS -> results on the host) over one frame whose inputs are already resident in HBM.  `--hot-path-only` stops after the
three query-generation stages; `--train` times fwd + bwd of a dummy loss + gradient all-reduce + AdamW instead.
Frames are independent, so for inference N ranks are N replicas with no data-path collective (weak scaling); the only
collectives are the barrier and the max-over-ranks of the timed region.  Rank 0 prints ONE JSON line with `roofline`
(dominant kernel: the fused sparse-conv implicit GEMM on the fp32 MFMA, timed with HIP events in an instrumented pass
over the same frames) and `cpu_baseline` (the CPU oracle restatement timed on this box's host cores on a bounded
sample).
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_* dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sweeps", type=int, default=10, help="10 = BASELINE config 3 input; 1 = config 2 (parity case)")
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


def make_inputs(sweeps, seed, device, frames=1, dataset="nuscenes"):
    from fullysparsefusion_amd import synthetic

    if dataset == "av2":
        fs = [synthetic.make_frame_av2(seed=seed * 97 + i) for i in range(frames)]
    else:
        fs = [synthetic.make_frame(num_sweeps=sweeps, seed=seed * 97 + i) for i in range(frames)]
    dev = dict(
        points=[torch.from_numpy(f["points"]).to(device) for f in fs],
        mask_data=torch.stack([torch.from_numpy(f["mask_data"]) for f in fs]).to(device),
        mask_anno=torch.stack([torch.from_numpy(f["mask_anno"]) for f in fs]).to(device),
        img_metas=[dict(lidar2img=torch.from_numpy(f["lidar2img"]).to(device)) for f in fs],
    )
    return fs[0], dev


def step(model, inp, hot_path_only=False):
    """One forward of the detector over one batch: FSF.simple_test (segmentation + image fusion, camera queries, LiDAR
    queries, query refinement, box decoding + NMS, results to the host) — or only its three query-generation stages."""
    with torch.no_grad():
        if hot_path_only:
            return model.forward_hot_path(inp["points"], inp["img_metas"], inp["mask_data"], inp["mask_anno"])
        return model.simple_test(inp["points"], inp["img_metas"], inp["mask_data"], inp["mask_anno"])


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


def dummy_loss(out):
    """'fwd+bwd with a dummy scalar loss (sum of head outputs)' — the loss/target path is not
    built, so every tensor the heads would consume contributes."""
    seg = out["seg"]
    return (seg["seg_logits"].sum() + seg["seg_vote_preds"].sum() + out["frustum_obj_feats"].sum()
            + out["fsd_obj_feats"].sum()) * 1e-6


class TrainStep:
    """fwd (training-mode norms) + bwd + gradient all-reduce (FrameDataParallel buckets, overlapped with the backward
    pass) + AdamW."""

    def __init__(self, model):
        from fullysparsefusion_amd.data_parallel import FrameDataParallel

        self.model = model.train()
        self.dp = FrameDataParallel(model)
        self.opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-6, weight_decay=0.01)

    def __call__(self, inp):
        self.dp.zero_grad()
        out = self.model.forward_hot_path(inp["points"], inp["img_metas"], inp["mask_data"], inp["mask_anno"])
        self.dp.backward(dummy_loss(out))
        self.opt.step()
        return out


def spconv_roofline(model, inp, steps, hot_path_only=False):
    """Roofline entry for the dominant op, the sparse convolution (34 layers per frame): every launch of the two forward
    kernels is timed with HIP events in an instrumented pass (the stream is torch's current stream, which is the one the
    C ABI launches on), algorithmic flops = 2 * pairs * Cin * Cout, algorithmic bytes = pairs * (Cin + Cout) * 4 +
    kvol * Cin * Cout * 4 + 8 * pairs.  `achieved` / `frac` are fp32-equivalent flops against the
    fp32 matrix-pipe peak for the op as a whole; `kernels` splits them by kernel: the fp32-pipe kernel
    (v_mfma_f32_16x16x4_f32) and the row-stationary kernel that forms the same fp32-accurate product from an exact 3-way
    bf16 split on v_mfma_f32_16x16x32_bf16 (six MFMAs per fp32-equivalent one: its own hardware ceiling is 2.5 PF / 6)."""
    from fullysparsefusion_amd import hip_ops

    records = []
    originals = {"fp32": hip_ops.spconv_forward, "split": hip_ops.spconv_forward_split}

    def wrap_fp32(feat, weight_t, nbr, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = originals["fp32"](feat, weight_t, nbr, **kw)
        e1.record()
        records.append(("fp32", e0, e1, nbr, weight_t.size(2), weight_t.size(1)))
        return out

    def wrap_split(feat, planes, kvol, cout, nbr, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = originals["split"](feat, planes, kvol, cout, nbr, **kw)
        e1.record()
        records.append(("split", e0, e1, nbr, feat.size(1), cout))
        return out

    hip_ops.spconv_forward, hip_ops.spconv_forward_split = wrap_fp32, wrap_split
    try:
        for _ in range(steps):
            step(model, inp, hot_path_only)
        torch.cuda.synchronize()
    finally:
        hip_ops.spconv_forward, hip_ops.spconv_forward_split = originals["fp32"], originals["split"]
    per = {"fp32": [0.0, 0.0, 0], "split": [0.0, 0.0, 0]}  # flops, ms, launches
    byts = 0.0
    for kind, e0, e1, nbr, cin, cout in records:
        pairs = float((nbr >= 0).sum())
        per[kind][0] += 2.0 * pairs * cin * cout
        per[kind][1] += e0.elapsed_time(e1)
        per[kind][2] += 1
        byts += pairs * (cin + cout) * 4 + nbr.size(1) * cin * cout * 4 + 8 * pairs
    flops = per["fp32"][0] + per["split"][0]
    ms = per["fp32"][1] + per["split"][1]
    launches = len(records)
    achieved = flops / (ms * 1e-3) / 1e12
    traffic, source = None, None
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for name in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        if name.endswith("_pmc_traffic.json"):
            with open(os.path.join(prof, name)) as f:
                t = json.load(f)
            traffic, source = t.get("spconv_forward", {}).get("hbm_bytes_per_api_launch"), f"profiles/{name}"
    kernels = {
        "fsf::spconv_fwd_dma_kernel (fp32 MFMA, compacting, output-stationary in LDS)": dict(
            launches_per_step=per["fp32"][2] // steps, ms_per_step=round(per["fp32"][1] / steps, 3),
            tflops=round(per["fp32"][0] / max(per["fp32"][1], 1e-9) / 1e9, 2), peak_tflops=157.3),
        "fsf::spconv_fwd_split_kernel (bf16 MFMA x6 = exact 3-way split, row-stationary in registers)": dict(
            launches_per_step=per["split"][2] // steps, ms_per_step=round(per["split"][1] / steps, 3),
            tflops_fp32_equivalent=round(per["split"][0] / max(per["split"][1], 1e-9) / 1e9, 2),
            peak_tflops_fp32_equivalent=round(2500.0 / 6, 1)),
    }
    return dict(bound="mfma", kernel="sparse convolution forward (fsf::spconv_fwd_split_kernel + fsf::spconv_fwd_dma_kernel)",
                achieved=round(achieved, 3), peak=157.3, unit="TFLOP/s", frac=round(achieved / 157.3, 4),
                peak_note="fp32 matrix-pipe peak; the split kernel reaches fp32 accuracy on the bf16 pipe, see `kernels`",
                kernels=kernels, traffic=traffic,
                traffic_unit="HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc)", traffic_source=source,
                launches_per_step=launches // steps, avg_launch_us=round(ms / launches * 1e3, 2),
                algorithmic_gflop_per_step=round(flops / steps / 1e9, 1), algorithmic_mb_per_step=round(byts / steps / 1e6, 1),
                ms_per_step_in_kernel=round(ms / steps, 3),
                note="HIP-event timing in an instrumented pass over the same frames, right after the timed region")


def cpu_baseline(model_cpu):
    """The CPU oracle (a port of the reference path: torch.unique + scatter_reduce + spconv-v1 restatement) on a
    bounded sample: ONE synthetic sweep (1/10 of a 10-sweep frame's points), stages 1-3."""
    from fullysparsefusion_amd import synthetic
    from oracle import modules as omod

    f = synthetic.make_frame(num_sweeps=1, seed=0)
    full_n = synthetic.make_points(10, 0).shape[0]
    pts8 = torch.from_numpy(f["points"])
    mask, anno, L = torch.from_numpy(f["mask_data"]), torch.from_numpy(f["mask_anno"]), torch.from_numpy(f["lidar2img"])
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    with torch.no_grad():
        s1 = omod.fsf_stage1(model_cpu, pts8, mask, anno, L)
        omod.fsf_stage2(model_cpu, s1, anno, (900, 1600))
        omod.fsf_stage3(model_cpu, s1)
    dt = time.perf_counter() - t0
    frac = pts8.shape[0] / full_n
    return dict(value=round(frac / dt, 5), unit="frames/s", cores=cores, kind="port",
                sample=f"1 of 10 sweeps ({pts8.shape[0]} of {full_n} points) through oracle stages 1-3 (segmentor + fusion, camera "
                       f"queries, LiDAR queries; the refine stage and NMS are not in the CPU sample, so this over-states the "
                       f"CPU rate of the full forward) in {dt:.1f} s; value = frame fraction / time")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=device)  # RCCL; barrier + max-reduce of the timing only
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    model = build_model(device, args.dataset)
    model_cpu = (None if args.no_cpu_baseline or rank != 0 or world != 1 or args.train or args.dataset != "nuscenes"
                 else copy.deepcopy(model).cpu())
    frame, inp = make_inputs(args.sweeps, seed=rank, device=device, frames=args.frames_per_gpu, dataset=args.dataset)
    if args.train:
        train_step = TrainStep(model)
        run = lambda: train_step(inp)
    else:
        run = lambda: step(model, inp, args.hot_path_only)

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        n_pts = int(inp["points"][0].shape[0])
        result = {
            "metric": (("frames/sec fwd+bwd+allreduce+AdamW nuScenes 10-sweep FSF (dummy loss)" if args.train
                        else "frames/sec fwd nuScenes 10-sweep FSF" + (" (query-generation stages only)" if args.hot_path_only else ""))
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
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ((f"fsf_nuscenes_{args.sweeps}sweep_" if args.dataset == "nuscenes" else "fsf_av2_long_range_") + (
                    "train_step: fwd of the query-generation stages + bwd of a dummy scalar loss + gradient all-reduce + AdamW"
                    if args.train else "hot_path_fwd: stages 1-3 of FSF.simple_test only" if args.hot_path_only else
                    "simple_test: full forward = segmentor + image fusion, camera queries, LiDAR queries, heads, query "
                    "refinement (RoI point pooling + SIR), box decode + rotated BEV NMS, results to host") +
                    (" (BASELINE config 3 input" if args.dataset == "nuscenes" else " (BASELINE config 5 shape") +
                    ", random-init weights of the reference architecture)"),
                "points_per_frame": n_pts,
                "frames_per_gpu_per_step": args.frames_per_gpu,
                "mask_data": "u8[1,6,10,900,1600]" if args.dataset == "nuscenes" else "i32[1,7,1,1550,2048]",
                **describe_output(model, inp, out, args),
                "parallelism": (f"dp{world}: frame-level data parallel, bucketed gradient all-reduce over RCCL" if args.train
                                else f"replicas x{world} (frames independent, no data-path collective)"),
            },
        }
    if rank == 0 and not args.no_roofline and not args.train:
        result["roofline"] = spconv_roofline(model, inp, min(args.steps, 5), args.hot_path_only)
        if args.dataset != "nuscenes":  # the committed PMC passes were taken on the nuScenes-shape workload
            result["roofline"].update(traffic=None, traffic_source=None)
        if not args.hot_path_only:  # where the frame time goes: the three query-generation stages vs the rest
            n = min(args.steps, 5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                step(model, inp, hot_path_only=True)
            torch.cuda.synchronize()
            result["stages"] = {"query_generation_ms": round((time.perf_counter() - t0) / n * 1e3, 3),
                                "full_forward_ms": result["ms_per_step"]}
    if rank == 0 and model_cpu is not None:
        result["cpu_baseline"] = cpu_baseline(model_cpu)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
