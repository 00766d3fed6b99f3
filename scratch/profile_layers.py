"""Per-launch table of the sparse-conv kernel + sizes along the hot path (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
model = bench.build_model(dev)
frame, inp = bench.make_inputs(sweeps, 0, dev)
for _ in range(2): bench.step(model, inp)
torch.cuda.synchronize()
rec = []
orig = hip_ops.spconv_forward
def timed(feat, wt, nbr, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(feat, wt, nbr, **kw); e1.record()
    rec.append((e0, e1, feat.shape[0], nbr.shape[0], wt.shape[2], wt.shape[1], int((nbr >= 0).sum()), (nbr >= 0).sum(0).tolist()))
    return out
hip_ops.spconv_forward = timed
out = bench.step(model, inp, hot_path_only=True)
torch.cuda.synchronize()
hip_ops.spconv_forward = orig
print(f"{'i':>3} {'m_in':>7} {'m_out':>7} {'cin':>5} {'cout':>5} {'pairs':>9} {'p/out':>6} {'us':>8} {'TF/s':>7} {'tile_eff':>8}")
tot = 0
for i, (e0, e1, mi, mo, ci, co, p, perk) in enumerate(rec):
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    print(f"{i:3d} {mi:7d} {mo:7d} {ci:5d} {co:5d} {p:9d} {p/max(mo,1):6.2f} {us:8.1f} {2*p*ci*co/us/1e6:7.2f}")
print('total spconv us', tot)
seg = out['seg']
print('points', inp['points'][0].shape, 'seg feats', seg['seg_feats'].shape)
print('frustum queries', out['frustum_obj_feats'].shape, 'lidar queries', out['fsd_obj_feats'].shape)
# sizes inside the LiDAR-query branch
cap = {}
f0 = model.backbone.forward
def capf(points, features, coors, f_cluster=None):
    cap['n'] = points.shape[0]; return f0(points, features, coors, f_cluster)
model.backbone.forward = capf
f1 = model.frustum_sir.forward
def capg(points, features, coors, f_cluster=None):
    cap['nf'] = points.shape[0]; return f1(points, features, coors, f_cluster)
model.frustum_sir.forward = capg
bench.step(model, inp, hot_path_only=True)
print('LiDAR SIR points', cap.get('n'), 'frustum SIR points', cap.get('nf'))
import torch.nn.functional as F
sc = seg['seg_logits'].softmax(1)
print('fg prob mean', float(1 - sc[:, -1].mean()))
