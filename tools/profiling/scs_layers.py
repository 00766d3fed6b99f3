"""Per-layer timing of the row-stationary split-bf16 conv (K9b) vs the fp32-pipe kernel on the frame's U-Net layers."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(2): bench.step(model, inp)
calls = []
orig = hip_ops.spconv_forward
def rec(feat, wt, nbr, **kw):
    calls.append((feat, wt, nbr, kw)); return orig(feat, wt, nbr, **kw)
hip_ops.spconv_forward = rec
with torch.no_grad(): bench.step(model, inp, hot_path_only=True)
hip_ops.spconv_forward = orig
def t(f, it=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
tot_a = tot_b = tot_best = 0
print(f"{'i':>3} {'m_out':>7} {'cin':>5} {'cout':>5} {'p/out':>6} {'fp32 us':>9} {'split us':>9} {'maxdiff':>9}")
for i, (feat, wt, nbr, kw) in enumerate(calls):
    kvol, cout, cin = wt.shape
    w = wt.permute(0, 2, 1).contiguous()       # [kvol, cin, cout]
    planes = hip_ops.spconv_prepare_weight_split(w)
    a = t(lambda: orig(feat, wt, nbr, **kw))
    b = t(lambda: hip_ops.spconv_forward_split(feat, planes, kvol, cout, nbr, **kw))
    d = float((orig(feat, wt, nbr, **kw) - hip_ops.spconv_forward_split(feat, planes, kvol, cout, nbr, **kw)).abs().max())
    p = float((nbr >= 0).sum()) / nbr.shape[0]
    tot_a += a; tot_b += b; tot_best += min(a, b)
    print(f"{i:3d} {nbr.shape[0]:7d} {cin:5d} {cout:5d} {p:6.2f} {a:9.1f} {b:9.1f} {d:9.2e}")
print('total fp32', tot_a, 'split', tot_b, 'best-of', tot_best)
