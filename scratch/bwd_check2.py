import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from fullysparsefusion_amd import hip_ops as ops
from oracle import spconv as osp
from test_hip_ops import surface_sites, sparse_sites
dev = torch.device('cuda:0')
for (cin, cout, m, shape, pad) in [(128, 128, 1000, (10, 128, 128), (1, 1, 1)), (128, 256, 300, (5, 64, 64), (0, 1, 1)), (256, 512, 80, (3, 32, 32), (1, 1, 1)),
                                   (128, 128, 3000, (20, 256, 256), (1, 1, 1))]:
    for kind in ['strided', 'inverse']:
        rng = np.random.default_rng(cin + m)
        idx = sparse_sites(rng, 1, shape, m)
        out_idx, pairs, _ = osp.build_rulebook(idx, 1, shape, (3, 3, 3), (2, 2, 2), pad, (1, 1, 1), False)
        _, nbr, nbr_inv, _ = ops.rulebook_strided(torch.from_numpy(idx).to(dev), 1, shape, (3, 3, 3), (2, 2, 2), pad)
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
        print(f'{kind:8s} cin {ci} cout {co} m_in {m_in} m_out {m_out}: fwd {r(fwd, want.detach()):.2e} dgrad {r(g_feat, feat.grad):.2e} wgrad {r(g_w, w.grad):.2e} pairs {int(num.sum())}')
