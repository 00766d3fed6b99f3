"""Per-layer timing of the K9b launches of one frame (rows arrive in the U-Net's neighbour-mask order)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(2): bench.step(model, inp)
calls = []
orig = hip_ops.spconv_forward_split
def rec(*a, **k):
    calls.append((a, k)); return orig(*a, **k)
hip_ops.spconv_forward_split = rec
with torch.no_grad(): bench.step(model, inp, hot_path_only=True)
hip_ops.spconv_forward_split = orig
def t(f, it=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
tot = 0
for i, (a, k) in enumerate(calls):
    feat, planes, kvol, cout, nbr = a[:5]
    us = t(lambda: orig(*a, **k)); tot += us
    live = (nbr >= 0)
    m = nbr.shape[0]; pad = (-m) % 128
    lw = torch.cat([live, torch.zeros(pad, kvol, dtype=torch.bool, device=dev)]).view(-1, 128, kvol).any(1).sum(1).float().mean().item()
    l32 = torch.cat([live, torch.zeros((-m) % 32, kvol, dtype=torch.bool, device=dev)]).view(-1, 32, kvol).any(1).sum(1).float().mean().item()
    print(f"{i:3d} m {m:7d} cin {feat.shape[1]:5d} cout {cout:5d} p/out {live.sum().item()/m:6.2f}  offsets/WG {lw:5.2f} /wave {l32:5.2f}  {us:8.1f} us")
print('total', tot)
