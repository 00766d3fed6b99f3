import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
cap = {}
orig = hip_ops.connected_components_grouped
def rec(points, group_idx, dist_table, *a, **k):
    cap['args'] = (points.clone(), group_idx.clone(), dist_table if not torch.is_tensor(dist_table) else dist_table.clone()); return orig(points, group_idx, dist_table, *a, **k)
hip_ops.connected_components_grouped = rec
import fullysparsefusion_amd.mmdet3d_plugin.models.detectors.single_stage_fsd as m
for name in dir(m):
    pass
with torch.no_grad(): bench.step(model, inp)
pts, grp, dt = cap['args']
print('n', pts.shape, 'groups', grp.min().item(), grp.max().item(), 'dist table', dt)
n = pts.shape[0]; T = (n + 255) // 256
pad = T * 256 - n
x = torch.cat([pts[:, 0], pts[-1:, 0].expand(pad)]).view(T, 256); y = torch.cat([pts[:, 1], pts[-1:, 1].expand(pad)]).view(T, 256)
g = torch.cat([grp, grp[-1:].expand(pad)]).view(T, 256)
x0, x1, y0, y1, g0, g1 = x.min(1)[0], x.max(1)[0], y.min(1)[0], y.max(1)[0], g.min(1)[0], g.max(1)[0]
dtt = torch.as_tensor(dt, device=dev, dtype=torch.float32)
dmax = torch.stack([dtt[a:b + 1].max() for a, b in zip(g0.tolist(), g1.tolist())])
print('tiles', T, 'all pairs', T * (T + 1) // 2)
grp_ok = ~((g1[:, None] < g0[None, :]) | (g1[None, :] < g0[:, None]))
gap = torch.maximum(torch.maximum(x0[None, :] - x1[:, None], x0[:, None] - x1[None, :]), torch.maximum(y0[None, :] - y1[:, None], y0[:, None] - y1[None, :]))
box_ok = gap <= torch.minimum(dmax[:, None], dmax[None, :])
iu = torch.triu(torch.ones(T, T, dtype=torch.bool, device=dev))
print('group-range survivors', int((grp_ok & iu).sum()), 'with box', int((grp_ok & box_ok & iu).sum()))
print('tile box widths: x mean', float((x1 - x0).mean()), 'y mean', float((y1 - y0).mean()))
print('first rows', pts[:5], grp[:5])
