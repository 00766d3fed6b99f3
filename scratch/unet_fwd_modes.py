import os, sys, copy, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import synthetic
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bench.build_model(torch.device('cpu')).eval()
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.8, 1.2); m.bias.data.normal_(0, 0.1)
seg = model.segmentor.to(dev)
f = synthetic.make_frame(num_sweeps=1, seed=3)
pts = [torch.from_numpy(f['points'][:12000, :5].copy()).to(dev)]
with torch.no_grad():
    bp, coors = seg.voxelize(pts)
    vf, vc, inv = seg.voxel_encoder(bp, coors, return_inv=True)
feats = {}
def hook(name):
    def h(mod, inp, out):
        feats.setdefault(name, []).append(out.features.detach().clone() if hasattr(out, 'features') else out[0]['voxel_feats'].detach().clone())
    return h
bb = seg.backbone
for n_, m_ in bb.named_children():
    if n_ == 'encoder_layers':
        for k, mm in m_.named_children(): mm.register_forward_hook(hook(k))
    else:
        m_.register_forward_hook(hook(n_))
with torch.no_grad():
    o1 = bb(dict(voxel_feats=vf, voxel_coors=vc, batch_size=1))[0]['voxel_feats']
x = vf.clone().requires_grad_()
o2 = bb(dict(voxel_feats=x, voxel_coors=vc, batch_size=1))[0]['voxel_feats']
for k, (a, b) in feats.items():
    print(f'{k:20s} rows {a.shape[0]:6d} ch {a.shape[1]:4d}  max|fused - unfused| / max = {float((a - b).abs().max() / a.abs().max()):.2e}')
