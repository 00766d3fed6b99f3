"""K22 fused Linear+LN+GELU vs the library GEMM + fsf_norm_act, at the SIR shapes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from fullysparsefusion_amd import hip_ops as ops
import torch.nn.functional as F
dev = torch.device('cuda:0')
def t(f, it=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
for n, k, c in [(510652, 256, 128), (510652, 180, 128), (510652, 136, 128), (250992, 256, 128), (310615, 128, 128), (310615, 128, 64), (50000, 256, 128), (10397, 128, 128), (10641, 1024, 1024), (10397, 768, 1024), (310615, 12, 64)]:
    kp = (k + 3) // 4 * 4
    x = torch.randn(n, kp, device=dev)[:, :k]
    xc = x.contiguous()
    w = torch.randn(c, k, device=dev) / k ** 0.5
    g = torch.rand(c, device=dev) + 0.5; b = torch.randn(c, device=dev) * 0.1
    planes = ops.linear_prepare_weight(w)
    t_f = t(lambda: ops.linear_norm_act(x, planes, c, norm='ln', gamma=g, beta=b, eps=1e-3, act='gelu')) if c <= 128 else t(lambda: F.gelu(F.layer_norm(ops.linear_norm_act(x, planes, c), (c,), g, b, 1e-3)))
    t_l = t(lambda: F.linear(xc, w))
    t_n = t(lambda: ops.norm_act(F.linear(xc, w), g, b, 1e-3, 'ln', 'gelu')) if c <= 512 else t(lambda: F.gelu(F.layer_norm(F.linear(xc, w), (c,), g, b, 1e-3)))
    fl = 2.0 * n * k * c
    print(f'n={n:7d} k={k:4d} c={c:4d}: fused {t_f:7.1f} us ({fl/t_f/1e6:6.1f} TF/s-equiv)   F.linear {t_l:7.1f} us   F.linear+norm_act {t_n:7.1f} us')
