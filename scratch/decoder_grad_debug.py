import os, sys, copy, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import bench
from oracle import modules as omod
from fullysparsefusion_amd.mmdet3d_plugin.ops.spconv import SparseConvTensor
from test_hip_ops import sparse_sites
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bench.build_model(torch.device('cpu')).eval()
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.8, 1.2); m.bias.data.normal_(0, 0.1)
bb_cpu = copy.deepcopy(model.segmentor.backbone)
bb = model.segmentor.backbone.to(dev)
rng = np.random.default_rng(0)
m_in = int(sys.argv[1]) if len(sys.argv) > 1 else 1286
shape = [20, 256, 256]
idx = sparse_sites(rng, 1, shape, m_in)
if len(sys.argv) > 2 and sys.argv[2] == 'real':
    from fullysparsefusion_amd import synthetic
    from oracle import voxelize as ovox, spconv as osp
    f = synthetic.make_frame(num_sweeps=1, seed=3)
    coors = ovox.dynamic_voxelize(f['points'][:12000, :5], synthetic.SEG_VOXEL, synthetic.PC_RANGE)
    vox = np.unique(coors[(coors >= 0).all(1)], axis=0)
    idx1 = np.concatenate([np.zeros((vox.shape[0], 1), np.int64), vox], 1).astype(np.int32)
    idx, _, shape = osp.build_rulebook(idx1, 1, [40, 512, 512], (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), False)
    shape = list(shape); m_in = idx.shape[0]
feat = torch.from_numpy(rng.standard_normal((m_in, 128)).astype(np.float32))
def run_oracle(mod, dtype):
    mod.zero_grad(set_to_none=True)
    x = feat.to(dtype).clone().requires_grad_()
    omod._GRAD[0] = True
    try:
        t = omod._SpT(x, idx, shape, 1, {})
        for block in mod.encoder_layers.encoder_layer3._modules.values():
            t = omod._convmodule(block, t)
        enc3 = t
        torch.manual_seed(2); xb = torch.randn(enc3.features.shape[0], 128, dtype=torch.float64).to(dtype).requires_grad_()
        lat = omod._basic_block(mod.lateral_layer3, enc3)
        cat = omod._SpT(torch.cat((xb, lat.features), dim=1), lat.indices, lat.shape, lat.batch_size, lat.rb)
        merged = omod._convmodule(mod.merge_layer3, cat)
        n_, c_out = merged.features.shape
        cat.features = merged.features + cat.features.view(n_, c_out, -1).sum(dim=2)
        out = omod._convmodule(mod.upsample_layer3, cat).features
    finally:
        omod._GRAD[0] = False
    torch.manual_seed(1); probe = torch.randn(out.shape, dtype=torch.float64)
    (out * probe.to(dtype)).sum().backward()
    return out.detach(), x.grad, xb.grad, {k: p.grad.clone() for k, p in mod.named_parameters() if p.grad is not None}, probe, xb.detach()
o32, gx32, gb32, g32, probe, xb0 = run_oracle(copy.deepcopy(bb_cpu), torch.float32)
o64, gx64, gb64, g64, _, _ = run_oracle(copy.deepcopy(bb_cpu).double(), torch.float64)
bb.zero_grad(set_to_none=True)
x = feat.to(dev).clone().requires_grad_()
t = SparseConvTensor(x, torch.from_numpy(idx).to(dev), shape, 1)
enc3 = bb.encoder_layers.encoder_layer3(t)
xb = xb0.float().to(dev).clone().requires_grad_()
out = bb.decoder_layer_forward(enc3, enc3._like(xb), bb.lateral_layer3, bb.merge_layer3, bb.upsample_layer3).features
(out * probe.float().to(dev)).sum().backward()
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
print('rows', m_in, enc3.features.shape[0], out.shape[0], 'fwd ours', f'{rel(out.detach(), o64):.2e}', 'cpu32', f'{rel(o32, o64):.2e}',
      'dx ours', f'{rel(x.grad, gx64):.2e}', 'cpu32', f'{rel(gx32, gx64):.2e}', 'dxb ours', f'{rel(xb.grad, gb64):.2e}', 'cpu32', f'{rel(gb32, gb64):.2e}')
params = dict(bb.named_parameters())
for k in g64:
    print(f'{k:40s} ours {rel(params[k].grad, g64[k]):.2e} cpu32 {rel(g32[k], g64[k]):.2e}')
