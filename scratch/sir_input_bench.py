"""Micro-benchmark of fsf_sir_input at the LiDAR-SIR shape (for rocprofv3 --pmc runs)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fullysparsefusion_amd import hip_ops as ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
n, p, cf, r = int(sys.argv[1]) if len(sys.argv) > 1 else 510652, 5, 175, 3
c = p + cf
points = torch.randn(n, 8, device=dev)[:, :p]
feats = torch.randn(n, cf, device=dev)
fcl = torch.randn(n, r, device=dev)
dims = [r, 16, 32, c]
layers = [(torch.randn(dims[i + 1], dims[i], device=dev) / dims[i] ** 0.5, torch.rand(dims[i + 1], device=dev) + 0.5,
           torch.randn(dims[i + 1], device=dev) * 0.1) for i in range(3)]
f = lambda: ops.sir_input(points, feats, fcl, [20.0, 20.0, 4.0], (*layers, 1e-3), 'gelu', 10.0)
for _ in range(3): f()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): out = f()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print(f'n={n} C={c}: {ms*1e3:.1f} us  {(n*(p+cf+r+c)*4)/ms/1e9:.2f} TB/s algorithmic')
