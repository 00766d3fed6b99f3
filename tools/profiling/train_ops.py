"""GPU time of one training step by aten op and input shape (torch.profiler; ROCm kernels attributed to the op that launched them).
usage: train_ops.py [top]"""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
model = bench.build_model(dev)
_, inp = bench.make_inputs(10, 0, dev)
ts = bench.TrainStep(model)
for _ in range(3): ts(inp)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ts(inp)
    torch.cuda.synchronize()
top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None: t = getattr(e, "self_cuda_time_total", 0)
    if t > 0: rows.append((t, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print(f"# self device time by (op, shapes), one training step; total {sum(r[0] for r in rows) / 1e3:.1f} ms")
for t, c, k, sh in rows[:top]:
    print(f"{t / 1e3:8.3f} ms  x{c:<4d} {k[:48]:48s} {sh}")
