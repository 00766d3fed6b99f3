"""Which dense GEMMs run in a frame: shapes, time, TFLOP/s (run on the GPU box)."""
import os, sys, torch, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
import torch.nn.functional as F
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(2): bench.step(model, inp)
torch.cuda.synchronize()
rec = []
orig = F.linear
def timed(x, w, b=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(x, w, b); e1.record()
    rec.append((e0, e1, tuple(x.shape), tuple(w.shape), b is not None))
    return out
F.linear = timed
torch.nn.functional.linear = timed
omm = torch.matmul
bench.step(model, inp)
torch.cuda.synchronize()
F.linear = orig
agg = collections.OrderedDict()
tot = 0
for e0, e1, xs, ws, hb in rec:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    k = (xs, ws, hb)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += us
print(f"{'x':>22} {'w':>14} bias calls   us/call  TF/s   GB/s(x+out)")
for (xs, ws, hb), (n, us) in agg.items():
    rows = 1
    for d in xs[:-1]: rows *= d
    fl = 2.0 * rows * ws[0] * ws[1]
    by = 4.0 * rows * (ws[0] + ws[1])
    print(f"{str(xs):>22} {str(ws):>14} {int(hb):4d} {n:5d} {us/n:9.1f} {fl/(us/n)/1e6:6.2f} {by/(us/n)/1e3:8.1f}")
print('total F.linear us', tot, 'calls', len(rec))
