"""K9b on rulebooks built straight from the synthetic frame's voxels (no model forward: usable with ablation builds)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fullysparsefusion_amd import hip_ops, synthetic
dev = torch.device('cuda:0')
f = synthetic.make_frame(num_sweeps=10, seed=0)
xyz = f["points"][:, :3]
lo = np.array([-51.2, -51.2, -5.0], np.float32)
c = np.floor((xyz - lo) / 0.2).astype(np.int64)
ok = ((c >= 0) & (c < np.array([512, 512, 40]))).all(1)
c = c[ok]
def t(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
for lvl, shape in ((0, [40, 512, 512]), (1, [20, 256, 256]), (2, [10, 128, 128])):
    cl = c >> lvl
    lin = np.unique((cl[:, 2] * shape[1] + cl[:, 1]) * shape[2] + cl[:, 0])
    z, y, x = lin // (shape[1] * shape[2]), lin // shape[2] % shape[1], lin % shape[2]
    idx = torch.from_numpy(np.stack([np.zeros_like(z), z, y, x], 1).astype(np.int32)).to(dev)
    nbr = hip_ops.rulebook_subm(idx, 1, shape, (3, 3, 3), (1, 1, 1))
    m = idx.size(0)
    for cin, cout in ((64, 64), (128, 128), (256, 128)):
        feat = torch.randn(m, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) / (27 * cin) ** 0.5
        planes = hip_ops.spconv_prepare_weight_split(w)
        us = t(lambda: hip_ops.spconv_forward_split(feat, planes, 27, cout, nbr))
        pairs = float((nbr >= 0).sum())
        print(f"level {lvl} m {m:7d} pairs/out {pairs / m:5.2f} {cin:4d}->{cout:4d}: {us:8.1f} us  {2 * pairs * cin * cout / us / 1e6:6.1f} TF/s-equiv")
