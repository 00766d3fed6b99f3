import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from fullysparsefusion_amd import hip_ops as ops
dev = torch.device('cuda:0')
n, k, c = 510652, 256, 128
x = torch.randn(n, k, device=dev); w = torch.randn(c, k, device=dev) / 16
g = torch.rand(c, device=dev) + 0.5; b = torch.randn(c, device=dev) * 0.1
planes = ops.linear_prepare_weight(w)
for _ in range(5): ops.linear_norm_act(x, planes, c, norm='ln', gamma=g, beta=b, eps=1e-3, act='gelu')
torch.cuda.synchronize()
