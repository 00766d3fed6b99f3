"""RoI pooling (K17) in the benchmark frame: how many rows each RoI keeps and the launch's time.  usage: pool_stats.py [sweeps]  (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device("cuda:0")
model = bench.build_model(dev)
_, inp = bench.make_inputs(int(sys.argv[1]) if len(sys.argv) > 1 else 10, 0, dev)
for _ in range(2): bench.step(model, inp)
calls = []
orig = hip_ops.dynamic_point_pool
def spy(*a, **k):
    out = orig(*a, **k); calls.append((a, k, out)); return out
hip_ops.dynamic_point_pool = spy
bench.step(model, inp)
hip_ops.dynamic_point_pool = orig
for a, k, out in calls:
    rois, pts = a[0], a[1]
    roi_idx = out[1]
    cnt = torch.bincount(roi_idx.clamp(min=0), minlength=rois.size(0))
    q = torch.quantile(cnt.float(), torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev)).tolist()
    print(f"rois {rois.size(0)} points {pts.size(0)} rows {roi_idx.numel()} max_inbox {a[3]}: rows per RoI median {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} max {q[3]:.0f}; "
          f"RoIs at the cap {int((cnt >= a[3]).sum())}, empty {int((cnt == 0).sum())}")
    def t(f, it=10):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): f()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
    print(f"  whole call {t(lambda: orig(*a, **k)):.0f} us")
