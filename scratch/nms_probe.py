"""What the final NMS sees on the bench frame: box count, BEV radius quantiles, extent."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
orig = hip_ops.nms_bev_multiclass
def rec(boxes, rank, count, thr, rotated=False):
    b = boxes
    hx, hy = 0.5 * (b[:, 2] - b[:, 0]), 0.5 * (b[:, 3] - b[:, 1])
    r = (hx * hx + hy * hy).sqrt()
    cx, cy = 0.5 * (b[:, 0] + b[:, 2]), 0.5 * (b[:, 1] + b[:, 3])
    q = torch.tensor([0.1, 0.5, 0.9, 0.99, 1.0], device=dev)
    print('n', b.shape[0], 'classes', rank.shape[0], 'count', count.tolist(), 'rotated', rotated, 'thr', thr)
    print('radius quantiles', torch.quantile(r, q).tolist(), 'cx range', float(cx.min()), float(cx.max()), 'cy', float(cy.min()), float(cy.max()))
    for R in (2.0, 4.0, 8.0): print('r >', R, int((r > R).sum()))
    return orig(boxes, rank, count, thr, rotated=rotated)
hip_ops.nms_bev_multiclass = rec
with torch.no_grad(): bench.step(model, inp)
