import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from fullysparsefusion_amd import hip_ops as ops
from oracle import spconv as osp
from test_hip_ops import surface_sites
dev = torch.device('cuda:0')
for cin, cout in [(16,16),(32,20),(64,64),(64,128)]:
    rng = np.random.default_rng(cin*1000+cout)
    shape=(16,48,48)
    idx = surface_sites(rng, 2, shape, 3000)
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    w = (rng.standard_normal((27, cin, cout)) / np.sqrt(cin*6)).astype(np.float32)
    _, pairs, _ = osp.build_rulebook(idx, 2, shape, (3,3,3),(1,1,1),(1,1,1),(1,1,1), True)
    want = osp.indice_conv(feat, w, pairs, idx.shape[0]).numpy()
    nbr = ops.rulebook_subm(torch.from_numpy(idx).to(dev), 2, shape)
    wt = ops.spconv_transpose_weight(torch.from_numpy(w).to(dev))
    out = ops.spconv_forward(torch.from_numpy(feat).to(dev), wt, nbr).cpu().numpy()
    err = np.abs(out-want)
    bad = err > 1e-4
    print(cin, cout, 'bad frac', bad.mean(), 'bad rows', bad.any(1).sum(), 'of', bad.shape[0], 'bad cols', np.nonzero(bad.any(0))[0][:20], 'max', err.max())
    rows = np.nonzero(bad.any(1))[0]
    print('  bad rows mod 64 hist', np.bincount(rows % 64, minlength=64)[:64].tolist() if len(rows) else None)
    print('  tiles with bad rows', np.unique(rows//64)[:30].tolist())
