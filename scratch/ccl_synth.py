import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
n = 83734
def t(f):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 100
x = torch.arange(n, device=dev, dtype=torch.float32) * 2.0
pts = torch.stack([x, torch.zeros_like(x), torch.zeros_like(x)], 1).contiguous()
grp = torch.zeros(n, dtype=torch.int32, device=dev)
dt = torch.tensor([0.6], device=dev)
import inspect
print(inspect.signature(hip_ops.connected_components_grouped))
print('line, grouped:', t(lambda: hip_ops.connected_components_grouped(pts, grp, dt)))
print('line, plain  :', t(lambda: hip_ops.connected_components(pts, 0.6)))
pts2 = torch.rand(n, 3, device=dev) * 100
print('random 100m, grouped:', t(lambda: hip_ops.connected_components_grouped(pts2, grp, dt)))
