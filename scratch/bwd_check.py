import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from fullysparsefusion_amd import hip_ops as ops
from oracle import spconv as osp
from test_hip_ops import surface_sites, sparse_sites
dev = torch.device('cuda:0')
for (cin, cout, m, shape) in [(256, 256, 1500, (12, 40, 40)), (512, 512, 1500, (12, 40, 40)), (512, 256, 300, (6, 20, 20)), (512, 512, 60, (3, 8, 8)),
                              (128, 128, 60, (3, 8, 8)), (256, 256, 300, (6, 20, 20)), (1024, 512, 200, (6, 20, 20))]:
    rng = np.random.default_rng(cin + m)
    idx = sparse_sites(rng, 1, shape, m)
    _, pairs, _ = osp.build_rulebook(idx, 1, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True)
    nbr = ops.rulebook_subm(torch.from_numpy(idx).to(dev), 1, shape)
    feat = torch.from_numpy(rng.standard_normal((m, cin)).astype(np.float32)).requires_grad_()
    w = torch.from_numpy((rng.standard_normal((27, cin, cout)) / np.sqrt(cin * 6)).astype(np.float32)).requires_grad_()
    gout = torch.from_numpy(rng.standard_normal((m, cout)).astype(np.float32))
    want = osp.indice_conv(feat, w, pairs, m)
    want.backward(gout)
    wd = w.detach().to(dev)
    g_feat = ops.spconv_forward(gout.to(dev), wd.flip(0).contiguous(), nbr)
    ip, num = ops.rulebook_to_pairs(nbr)
    g_w = ops.spconv_backward_weight(feat.detach().to(dev), gout.to(dev), ip, num)
    e1 = float((g_feat.cpu() - feat.grad).abs().max() / feat.grad.abs().max())
    e2 = float((g_w.cpu() - w.grad).abs().max() / w.grad.abs().max())
    fwd = ops.spconv_forward(feat.detach().to(dev), ops.spconv_transpose_weight(wd), nbr)
    e0 = float((fwd.cpu() - want.detach()).abs().max() / want.detach().abs().max())
    print(f'cin {cin} cout {cout} m {m}: fwd {e0:.2e} dgrad {e1:.2e} wgrad {e2:.2e}  pairs {int(num.sum())}')
