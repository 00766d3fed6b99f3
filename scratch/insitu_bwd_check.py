import os, sys, copy, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import synthetic, hip_ops
from fullysparsefusion_amd.mmdet3d_plugin.ops import spconv as sp
from oracle import modules as omod
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = bench.build_model(torch.device('cpu')).eval()
cpu = copy.deepcopy(model)
seg = model.segmentor.to(dev)
f = synthetic.make_frame(num_sweeps=1, seed=3)
pts = torch.from_numpy(f['points'][:12000, :5].copy())
ex = omod.segmentor_extract_feat(cpu.segmentor, [pts])
vf, vc = ex['voxel_feats'].detach(), ex['voxel_coors']
orig_bwd = sp._SparseConvFn.backward
def ref_conv(src, w, table):  # out[o] = sum_k src[table[o,k]] @ w[k]   (w [27, cin', cout'] as weight_t [k][cout'][cin'] semantics: out = src @ w[k].T)
    out = torch.zeros((table.size(0), w.size(1)), dtype=torch.float64, device=src.device)
    for k in range(table.size(1)):
        idx = table[:, k].long(); ok = idx >= 0
        out[ok] += src[idx[ok]].double() @ w[k].double().t()
    return out
def checked(ctx, grad):
    feat, weight = ctx.saved_tensors
    rb, inverse = ctx.rb, ctx.inverse
    res = orig_bwd(ctx, grad)
    kvol = rb.nbr.size(1)
    w = weight.detach().reshape(kvol, weight.shape[-2], weight.shape[-1])
    table_t, flip = rb.table_transposed(inverse)
    ref = ref_conv(grad.contiguous(), w.flip(0) if flip else w, table_t)
    e = float((res[0].double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    print(f'kind {rb.kind:8s} inverse {inverse!s:5s} rows_in {feat.shape[0]:5d} rows_out {grad.shape[0]:5d} cin {w.shape[1]:4d} cout {w.shape[2]:4d}  dgrad err {e:.2e}  grad contiguous {grad.is_contiguous()} strides {tuple(grad.stride())}')
    return res
sp._SparseConvFn.backward = staticmethod(checked)
x = vf.to(dev).clone().requires_grad_()
out = seg.backbone(dict(voxel_feats=x, voxel_coors=vc.to(dev), batch_size=1))[0]['voxel_feats']
out.sum().backward()
