"""Every segmented reduction of one 10-sweep frame: shape, mode, call site, time replayed in isolation."""
import os, sys, torch, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); model.test_cfg['concurrent_query_branches'] = False
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(2): bench.step(model, inp)
calls = []
orig = hip_ops.segment_reduce
def site():
    for fs in reversed(traceback.extract_stack()[:-2]):
        if 'fullysparsefusion_amd' in fs.filename and 'hip_ops' not in fs.filename and 'sst_ops.py' not in fs.filename:
            return f"{os.path.basename(fs.filename)}:{fs.lineno} {fs.name}"
    return '?'
def rec(*a, **k):
    out = orig(*a, **k); calls.append((a, k, site())); return out
hip_ops.segment_reduce = rec
with torch.no_grad(): bench.step(model, inp)
hip_ops.segment_reduce = orig
def t(f, it=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
rows = []
for a, k, s in calls:
    feat, plan, mode = a[0], a[1], a[2]
    us = t(lambda: orig(*a, **k))
    cnt = (plan.seg_offsets[1:] - plan.seg_offsets[:-1]).max().item()
    rows.append((us, feat.shape[0], feat.shape[1], plan.m, mode, feat.stride(0), cnt, s))
print(f"{len(rows)} calls, {sum(r[0] for r in rows):.0f} us")
for r in sorted(rows, reverse=True): print(f"{r[0]:8.1f} us  n={r[1]:7d} c={r[2]:4d} m={r[3]:6d} mode={r[4]:4s} stride={r[5]:4d} longest={r[6]:6d}  {r[7]}")
# how often are 16 / 32 / 128 consecutive rows one segment (what an in-epilogue segmented max could fold before its atomics)
seen = set()
for a, k, s in calls:
    plan = a[1]
    if a[2] != 'max' or id(plan) in seen: continue
    seen.add(id(plan))
    inv = plan.inv
    n = inv.numel()
    line = f"n={n:7d} m={plan.m:6d}"
    for g in (16, 32, 128):
        nb = n // g
        blk = inv[:nb * g].view(nb, g)
        uni = (blk == blk[:, :1]).all(1).float().mean().item()
        runs = (blk[:, 1:] != blk[:, :-1]).sum(1).float().mean().item() + 1
        line += f" | {g}: uniform {uni:.3f} runs/blk {runs:.2f}"
    print(line, s)
