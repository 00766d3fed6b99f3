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
pts = torch.from_numpy(f['points'][:12000, :5].copy())
ex = omod.segmentor_extract_feat(cpu.segmentor, [pts])
vf, vc = ex['voxel_feats'].detach(), ex['voxel_coors']
probe = torch.from_numpy(np.random.default_rng(1).standard_normal((vc.shape[0], 128)).astype(np.float32))
# ---- oracle with recorded activations
rec_o = []
cm, bbk = omod._convmodule, omod._basic_block
def cm2(seq, x):
    t = cm(seq, x); t.features.retain_grad(); rec_o.append(t.features); return t
def bb2(blk, x):
    t = bbk(blk, x); t.features.retain_grad(); rec_o.append(t.features); return t
omod._convmodule, omod._basic_block = cm2, bb2
rec_oc = []
cv = omod._conv
def cv2(conv, x):
    t = cv(conv, x); t.features.retain_grad(); rec_oc.append(t.features); return t
omod._conv = cv2
mod64 = copy.deepcopy(cpu).double()
x64 = vf.double().clone().requires_grad_()
out64 = omod.unet_forward(mod64.segmentor.backbone, x64, vc, 1, grad=True)
(out64 * probe.double()).sum().backward()
omod._convmodule, omod._basic_block = cm, bbk
omod._conv = cv
# ---- GPU with hooks
rec_g = []
def hook(mod, inp, out):
    out.features.retain_grad(); rec_g.append(out.features)
bb = seg.backbone
hs = [bb.conv_input.register_forward_hook(hook)]
for stage in bb.encoder_layers._modules.values():
    for blk in stage._modules.values(): hs.append(blk.register_forward_hook(hook))
for i in range(bb.stage_num, 0, -1):
    for nm in (f'lateral_layer{i}', f'merge_layer{i}', f'upsample_layer{i}'): hs.append(getattr(bb, nm).register_forward_hook(hook))
from fullysparsefusion_amd.mmdet3d_plugin.ops.spconv import SparseConvolution
rec_gc, conv_names = [], []
def chook(name):
    def h(mod, inp, out):
        out.features.retain_grad(); rec_gc.append(out.features); conv_names.append(name)
    return h
for n_, m_ in bb.named_modules():
    if isinstance(m_, SparseConvolution): hs.append(m_.register_forward_hook(chook(n_)))
x = vf.to(dev).clone().requires_grad_()
out = bb(dict(voxel_feats=x, voxel_coors=vc.to(dev), batch_size=1))[0]['voxel_feats']
(out * probe.to(dev)).sum().backward()
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
names = ['conv_input'] + [f'enc{s+1}.{b}' for s, st in enumerate(bb.encoder_layers._modules.values()) for b in range(len(st._modules))]
for i in range(bb.stage_num, 0, -1): names += [f'lateral{i}', f'merge{i}', f'upsample{i}']
assert len(rec_o) == len(rec_g) == len(names), (len(rec_o), len(rec_g), len(names))
for n_, a, b in zip(names, rec_g, rec_o):
    print(f'{n_:12s} rows {a.shape[0]:5d} act err {rel(a.detach(), b.detach()):.2e}   grad err {rel(a.grad, b.grad):.2e}')

print('--- conv outputs (pre-norm)')
assert len(rec_oc) == len(rec_gc), (len(rec_oc), len(rec_gc))
for n_, a, b in zip(conv_names, rec_gc, rec_oc):
    ga, gb = a.grad.double().cpu(), b.grad.double()
    d = (ga - gb).abs()
    bad = int((d > 1e-4 * gb.abs().max()).sum())
    print(f'{n_:40s} rows {a.shape[0]:5d} act err {rel(a.detach(), b.detach()):.2e}   grad err {rel(a.grad, b.grad):.2e}  elements off by >1e-4: {bad} of {d.numel()}')
