import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fullysparsefusion_amd import hip_ops as ops, synthetic
from oracle import spconv as osp, voxelize as ovox
dev = torch.device('cuda:0')
f = synthetic.make_frame(num_sweeps=1, seed=3)
pts = f['points'][:12000, :5]
coors = ovox.dynamic_voxelize(pts, synthetic.SEG_VOXEL, synthetic.PC_RANGE)
vox = np.unique(coors[(coors >= 0).all(1)], axis=0)
idx1 = np.concatenate([np.zeros((vox.shape[0], 1), np.int64), vox], 1).astype(np.int32)
shape1 = [40, 512, 512]
idx2, _, shape2 = osp.build_rulebook(idx1, 1, shape1, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), False)
print('level1', idx1.shape[0], 'level2', idx2.shape[0])
for (idx, shape, pad, cin, cout) in [(idx2, shape2, (1, 1, 1), 128, 128)]:
    out_idx, pairs, oshape = osp.build_rulebook(idx, 1, shape, (3, 3, 3), (2, 2, 2), pad, (1, 1, 1), False)
    _, nbr, nbr_inv, _ = ops.rulebook_strided(torch.from_numpy(idx).to(dev), 1, shape, (3, 3, 3), (2, 2, 2), pad)
    print('level3', out_idx.shape[0], 'pairs', sum(len(p[0]) for p in pairs))
    for kind in ['strided', 'inverse']:
        rng = np.random.default_rng(5)
        if kind == 'strided':
            table, table_t, m_in, m_out, ci, co = nbr, nbr_inv, idx.shape[0], out_idx.shape[0], cin, cout
        else:
            table, table_t, m_in, m_out, ci, co = nbr_inv, nbr, out_idx.shape[0], idx.shape[0], cout, cin
        feat = torch.from_numpy(rng.standard_normal((m_in, ci)).astype(np.float32)).requires_grad_()
        w = torch.from_numpy((rng.standard_normal((27, ci, co)) / np.sqrt(ci * 6)).astype(np.float32)).requires_grad_()
        gout = torch.from_numpy(rng.standard_normal((m_out, co)).astype(np.float32))
        want = osp.indice_conv(feat, w, pairs, m_out, inverse=kind == 'inverse')
        want.backward(gout)
        wd = w.detach().to(dev)
        fwd = ops.spconv_forward(feat.detach().to(dev), ops.spconv_transpose_weight(wd), table)
        g_feat = ops.spconv_forward(gout.to(dev), wd, table_t)
        ip, num = ops.rulebook_to_pairs(table)
        g_w = ops.spconv_backward_weight(feat.detach().to(dev), gout.to(dev), ip, num)
        r = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
        per_k = [float((g_w[k].cpu() - w.grad[k]).abs().max() / w.grad.abs().max()) for k in range(27)]
        print(f'{kind:8s} m_in {m_in} m_out {m_out}: fwd {r(fwd, want.detach()):.2e} dgrad {r(g_feat, feat.grad):.2e} wgrad {r(g_w, w.grad):.2e}')
        print('  per-offset wgrad err', ' '.join(f'{e:.0e}' for e in per_k))
        print('  num per offset', num.cpu().tolist())
