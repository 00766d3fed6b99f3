"""Every K22 / K22s / K22h call of one 10-sweep frame with its shape and its time replayed in isolation (sorted by time)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(2): bench.step(model, inp)
calls = []
names = ["linear_norm_act", "linear_norm_act_segmax", "linear_norm_act_sliced", "linear_planes_norm_act", "rows_to_planes", "norm_act",
         "sir_input"]
orig = {n: getattr(hip_ops, n) for n in names}
def mk(n):
    def rec(*a, **k):
        out = orig[n](*a, **k); calls.append((n, a, k)); return out
    return rec
for n in names: setattr(hip_ops, n, mk(n))
model.test_cfg["concurrent_query_branches"] = False
with torch.no_grad(): bench.step(model, inp)
for n in names: setattr(hip_ops, n, orig[n])
def t(f, it=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
rows = []
for n, a, k in calls:
    x = a[0]
    if n == "linear_planes_norm_act": shape = (x.n, x.c, int(a[2]))
    elif n == "rows_to_planes" or n == "norm_act": shape = (x.shape[0], x.shape[1], x.shape[1])
    elif n == "linear_norm_act_sliced": shape = (x.shape[0], int(a[1]), int(a[4]) * int(a[5]))
    elif n == "sir_input": shape = (x.shape[0], 0, 0)
    else: shape = (x.shape[0], x.shape[1], int(a[2]))
    if n == "linear_norm_act_segmax":
        import functools
        so = a[4]
        def f(a=a, k=k, so=so):
            so.fill_(float("-inf")); orig[n](*a, **k)
        us = t(f) - t(lambda: so.fill_(float("-inf")))
    else:
        us = t(lambda: orig[n](*a, **k))
    rows.append((us, n, shape, k.get('norm', ''), k.get('act', ''), k.get('row_add') is not None, k.get('want_rows', None)))
tot = {}
for r in rows: tot[r[1]] = tot.get(r[1], 0.0) + r[0]
print(f"{len(rows)} calls; per family us:", {k: round(v) for k, v in tot.items()}, "total", round(sum(tot.values())))
for r in sorted(rows, reverse=True)[:60]:
    n, kk, c = r[2]
    tf = 2 * n * kk * c / r[0] / 1e6 if kk and c and r[1].startswith("linear") else 0.0
    print(f"{r[0]:8.1f} us  {r[1]:24s} n={n:7d} k={kk:5d} c={c:5d} norm={str(r[3]):6s} act={str(r[4]):5s} grouped={int(r[5])} rows={r[6]}  {tf:6.1f} TF")
