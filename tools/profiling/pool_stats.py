"""RoI / pooled-point statistics of the bench frame's refine stage (what the pooling kernels see).  (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
calls = []
orig = hip_ops.dynamic_point_pool
def spy(*a, **k):
    calls.append((a, k)); return orig(*a, **k)
hip_ops.dynamic_point_pool = spy
bench.step(model, inp)
hip_ops.dynamic_point_pool = orig
for a, k in calls:
    rois, pts = a[0], a[1]
    box = rois[:, k.get('box_col', 0):][:, :7]
    print("rois", tuple(rois.shape), "pts", tuple(pts.shape), "args", a[2:], {kk: v for kk, v in k.items() if kk != 'pts_batch'})
    for name, col in (("w", 3), ("l", 4), ("h", 5)):
        v = box[:, col]
        print(f"  {name}: min {v.min():.2f} median {v.median():.2f} p90 {v.quantile(0.9):.2f} max {v.max():.2f}")
    gp, gr, gf = orig(rois, pts, a[2], 10 ** 6, 10 ** 8, **{kk: v for kk, v in k.items()})
    cnt = torch.bincount(gr, minlength=rois.size(0)).float()
    print(f"  uncapped hits per RoI: median {cnt.median():.0f} mean {cnt.mean():.0f} p90 {cnt.quantile(0.9):.0f} max {cnt.max():.0f}; first RoIs: {cnt[:12].tolist()}")
    c512 = cnt.clamp(max=a[3]).cumsum(0)
    print(f"  RoIs until max_all={a[4] if len(a) > 4 else k.get('max_all_pts')}: {(c512 < (a[4] if len(a) > 4 else 50000)).sum().item()}")
