import os, sys, copy, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import synthetic
from oracle import modules as omod
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bench.build_model(torch.device('cpu')).eval()
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.8, 1.2); m.bias.data.normal_(0, 0.1)
cpu = copy.deepcopy(model)
seg = model.segmentor.to(dev)
f = synthetic.make_frame(num_sweeps=1, seed=3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
pts = torch.from_numpy(f['points'][:n, :5].copy())
ex = omod.segmentor_extract_feat(cpu.segmentor, [pts])
vf, vc = ex['voxel_feats'].detach(), ex['voxel_coors']
probe = torch.from_numpy(np.random.default_rng(1).standard_normal((vc.shape[0], 128)).astype(np.float32))
def run_oracle(mod, dtype):
    mod.zero_grad(set_to_none=True)
    x = vf.to(dtype).clone().requires_grad_()
    out = omod.unet_forward(mod.segmentor.backbone, x, vc, 1, grad=True)
    (out * probe.to(dtype)).sum().backward()
    return out.detach(), {n_: p.grad.clone() for n_, p in mod.segmentor.backbone.named_parameters()}, x.grad
o32, g32, gx32 = run_oracle(copy.deepcopy(cpu), torch.float32)
o64, g64, gx64 = run_oracle(copy.deepcopy(cpu).double(), torch.float64)
x = vf.to(dev).clone().requires_grad_()
out = seg.backbone(dict(voxel_feats=x, voxel_coors=vc.to(dev), batch_size=1))[0]['voxel_feats']
(out * probe.to(dev)).sum().backward()
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
print('voxels', vc.shape[0], 'fwd ours', rel(out.detach(), o64), 'cpu32', rel(o32, o64), ' dx ours', rel(x.grad, gx64), 'cpu32', rel(gx32, gx64))
for n_, p in seg.backbone.named_parameters():
    print(f'{n_:50s} ours {rel(p.grad, g64[n_]):.2e} cpu32 {rel(g32[n_], g64[n_]):.2e}  shape {tuple(p.shape)}')
