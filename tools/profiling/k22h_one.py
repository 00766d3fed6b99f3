"""ONE K22h shape for counter passes: 10 641 query rows x 1024 -> 1024 (the heads' shared MLP / out_proj layers), x in plane form.
`python tools/profiling/k22h_one.py [reps]`; FSF_K22H_ONE=k22 runs the same product on K22 (bf16 x 6) instead."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from fullysparsefusion_amd import hip_ops as ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
n, k, c = 10641, 1024, 1024
x = torch.randn(n, k, device=dev).clamp_min(0) * 2
w = torch.randn(c, k, device=dev) / k ** 0.5
b = torch.randn(c, device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if os.environ.get("FSF_K22H_ONE") == "k22":
    planes = ops.linear_prepare_weight(w, fmt="bf16x6")
    for _ in range(reps): ops.linear_norm_act(x, planes, c, bias=b)
else:
    wp = ops.linear_prepare_weight_f16(w, 128); xp = ops.rows_to_planes(x)
    for _ in range(reps): ops.linear_planes_norm_act(xp, wp, c, 128, bias=b)
torch.cuda.synchronize()
