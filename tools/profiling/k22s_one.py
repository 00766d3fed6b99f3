"""ONE K22s shape for counter passes: the LiDAR-query SIR stack's grouped 128 -> 128 layer (510 652 rows sorted by group, 10 397 groups,
one 1.2e5-row group), LayerNorm + GELU + segmented max, rows written.  `python tools/profiling/k22s_one.py [reps]`"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from fullysparsefusion_amd import hip_ops as ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
n, g = 510652, 10397
x = torch.randn(n, 128, device=dev)
w = torch.randn(128, 128, device=dev) / 128 ** 0.5
gam = torch.rand(128, device=dev) + 0.5; bet = torch.randn(128, device=dev) * 0.1
planes = ops.linear_prepare_weight(w)
ids = torch.randint(0, g, (n,), device=dev); ids[:120000] = 17
sid = torch.sort(ids)[0]
u, sinv = torch.unique(sid, return_inverse=True)
so = torch.full((u.numel(), 128), float("-inf"), device=dev)
tb = torch.randn(u.numel(), 128, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ops.linear_norm_act_segmax(x, planes, 128, sinv, so, norm='ln', gamma=gam, beta=bet, eps=1e-3, act='gelu', row_add=tb, row_add_index=sinv)
torch.cuda.synchronize()
