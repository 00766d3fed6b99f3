"""The ATen glue of one training step by WHO runs it: every op that launches device work, grouped by (top-level ancestor — the autograd
node in the backward, the module-level op in the forward; op name; input shapes), device time and launches.  (GPU box)"""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from torch.profiler import ProfilerActivity, profile
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); torch.manual_seed(0)
model = bench.build_model(dev)
_, inp = bench.make_inputs(10, 0, dev)
ts = bench.TrainStep(model)
for _ in range(3): ts(inp)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ts(inp); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
tot = [0, 0.0]
for e in prof.events():
    ks = getattr(e, "kernels", None)
    if not ks: continue
    if e.cpu_parent is not None and getattr(e.cpu_parent, "kernels", None): continue   # count at the outermost op that owns the kernels
    top = e
    while top.cpu_parent is not None: top = top.cpu_parent
    if e.name.startswith("hip") or "fsf" in e.name.lower(): continue                   # C-ABI launches: not glue
    shapes = str(getattr(e, "input_shapes", ""))[:60]
    key = (top.name[:60], e.name[:28], shapes)
    us = sum(k.duration for k in ks)
    agg[key][0] += len(ks); agg[key][1] += us; tot[0] += len(ks); tot[1] += us
print(f"ATen launches per training step: {tot[0]}, device time {tot[1] / 1e3:.2f} ms")
for (top, op, shapes), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{us:8.0f} us {n:4d}  {op:28s} {shapes:60s} <- {top}")
