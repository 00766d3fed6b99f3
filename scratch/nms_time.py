"""Final multi-class NMS of the bench frame, timed alone."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
orig = hip_ops.nms_bev_multiclass
cap = []
def rec(*a, **k):
    cap.append((a, k)); return orig(*a, **k)
hip_ops.nms_bev_multiclass = rec
with torch.no_grad(): bench.step(model, inp)
a, k = cap[-1]
for _ in range(3): orig(*a, **k)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): orig(*a, **k)
e1.record(); torch.cuda.synchronize()
print('nms_bev_multiclass', a[0].shape[0], 'boxes x', a[1].shape[0], 'classes:', round(e0.elapsed_time(e1) / 10 * 1e3, 1), 'us')
