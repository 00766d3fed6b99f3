"""Same-process interleaved A/B of the announced loop (FSF.set_next_frame, K32) against the plain loop: usage
python tools/profiling/frame_front_ab.py [sweeps] [rounds] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
dataset = os.environ.get("AB_DATASET", "nuscenes")
tl = os.environ.get("AB_TRAINED_LIKE") == "1"
model = bench.build_model(dev, dataset)
pool = [bench.make_inputs(sweeps, seed=j, device=dev, dataset=dataset, trained_like=tl)[1] for j in range(4)]
if tl:
    bench.calibrate_trained_like(model, pool[0])


def loop(announce, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        bench.step(model, pool[i % 4], False, pool[(i + 1) % 4] if announce else None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


for name in os.environ.get("AB_DISABLE", "").split(","):  # e.g. AB_DISABLE=_camera_rows_early,_pre_voxel_keys_early
    if name:
        setattr(model, name, lambda *a, **k: None)
loop(False, 5)
loop(True, 5)
res = {True: [], False: []}
for r in range(rounds):
    for a in (True, False) if r % 2 == 0 else (False, True):
        res[a].append(loop(a, steps))
print(f"# {sweeps}-sweep frame, {rounds} interleaved rounds of {steps} frames, ms per frame")
print("announced  ", " ".join(f"{v:7.3f}" for v in res[True]), f"  median {sorted(res[True])[rounds // 2]:.3f}")
print("unannounced", " ".join(f"{v:7.3f}" for v in res[False]), f"  median {sorted(res[False])[rounds // 2]:.3f}")
