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
feat = torch.from_numpy(rng.standard_normal((m_in, 128)).astype(np.float32))
stage_name = sys.argv[2] if len(sys.argv) > 2 else 'encoder_layer3'
stage = getattr(bb.encoder_layers, stage_name)
stage_cpu = getattr(bb_cpu.encoder_layers, stage_name)
def run_oracle(mod, dtype):
    mod.zero_grad(set_to_none=True)
    x = feat.to(dtype).clone().requires_grad_()
    omod._GRAD[0] = True
    try:
        t = omod._SpT(x, idx, shape, 1, {})
        for block in mod._modules.values():
            t = omod._convmodule(block, t)
    finally:
        omod._GRAD[0] = False
    out = t.features
    torch.manual_seed(1); probe = torch.randn(out.shape, dtype=torch.float64)
    (out * probe.to(dtype)).sum().backward()
    return out.detach(), x.grad, {n_: p.grad.clone() for n_, p in mod.named_parameters()}, probe
o32, gx32, g32, probe = run_oracle(copy.deepcopy(stage_cpu), torch.float32)
o64, gx64, g64, _ = run_oracle(copy.deepcopy(stage_cpu).double(), torch.float64)
x = feat.to(dev).clone().requires_grad_()
t = SparseConvTensor(x, torch.from_numpy(idx).to(dev), shape, 1)
out = stage(t).features
(out * probe.float().to(dev)).sum().backward()
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
print('rows in', m_in, 'rows out', out.shape[0], 'fwd ours', f'{rel(out.detach(), o64):.2e}', 'cpu32', f'{rel(o32, o64):.2e}', 'dx ours', f'{rel(x.grad, gx64):.2e}', 'cpu32', f'{rel(gx32, gx64):.2e}')
for n_, p in stage.named_parameters():
    print(f'{n_:20s} ours {rel(p.grad, g64[n_]):.2e} cpu32 {rel(g32[n_], g64[n_]):.2e}')
