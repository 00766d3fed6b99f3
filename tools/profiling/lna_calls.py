"""Every K22 (fsf_linear_norm_act) call of one 10-sweep frame with its shape and its time replayed in isolation."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(2): bench.step(model, inp)
calls = []
orig = hip_ops.linear_norm_act
def rec(*a, **k):
    out = orig(*a, **k); calls.append((a, k)); return out
hip_ops.linear_norm_act = rec
with torch.no_grad(): bench.step(model, inp)
hip_ops.linear_norm_act = orig
def t(f, it=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
rows = []
for a, k in calls:
    x, c = a[0], int(a[2])
    us = t(lambda: orig(*a, **k))
    rows.append((us, x.shape[0], x.shape[1], c, k.get('norm', 'none'), k.get('act', 'none'), k.get('row_add') is not None, x.stride(0)))
tot = sum(r[0] for r in rows)
print(f"{len(rows)} calls, {tot:.0f} us")
for r in sorted(rows, reverse=True)[:40]:
    n, kk, c = r[1], r[2], r[3]
    print(f"{r[0]:8.1f} us  n={n:7d} k={kk:5d} c={c:5d} norm={r[4]:6s} act={r[5]:5s} grouped={int(r[6])} xstride={r[7]}  {n*(kk+c)*4/r[0]/1e6:6.2f} TB/s  {2*n*kk*c/r[0]/1e6:6.1f} TF")
