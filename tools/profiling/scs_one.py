import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
calls = []
orig = hip_ops.spconv_forward_split
def rec(*a, **k):
    calls.append((a, k)); return orig(*a, **k)
hip_ops.spconv_forward_split = rec
with torch.no_grad(): bench.step(model, inp, hot_path_only=True)
hip_ops.spconv_forward_split = orig
which = int(sys.argv[1]) if len(sys.argv) > 1 else 2
a, k = calls[which]
print('layer', which, 'feat', tuple(a[0].shape), 'cout', a[3], 'rows', a[4].shape[0], 'pairs/out', float((a[4] >= 0).sum()) / a[4].shape[0])
for _ in range(5): orig(*a, **k)
torch.cuda.synchronize()
