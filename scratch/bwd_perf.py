"""Backbone fwd+bwd timing on the 10-sweep frame (training-mode norms) + per-op TFLOP/s of the biggest layer."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import hip_ops as ops
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
seg = model.segmentor
with torch.no_grad():
    pts = [inp['points'][0][:, :5].contiguous()]
    bp, coors = seg.voxelize(pts)
    vf, vc, inv = seg.voxel_encoder(bp, coors, return_inv=True)
m = vc.shape[0]
print('voxels', m)

def ev():
    return torch.cuda.Event(enable_timing=True)

# ---- single biggest layer: subm 128 -> 128 on all voxels
nbr = ops.rulebook_subm(vc.int().contiguous(), 1, seg.backbone.sparse_shape)
pairs, num = ops.rulebook_to_pairs(nbr)
npairs = int(num.sum())
for cin, cout in [(128, 128), (64, 64), (256, 256)]:
    feat = torch.randn(m, cin, device=dev)
    gout = torch.randn(m, cout, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    wt = ops.spconv_transpose_weight(w)
    fl = 2.0 * npairs * cin * cout
    for name, fn in [('fwd', lambda: ops.spconv_forward(feat, wt, nbr)),
                     ('dgrad', lambda: ops.spconv_forward(gout, w.flip(0).contiguous(), nbr)),
                     ('wgrad', lambda: ops.spconv_backward_weight(feat, gout, pairs, num))]:
        for _ in range(3):
            fn()
        a, b = ev(), ev()
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f'{cin}x{cout} {name}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  (pairs {npairs})')

# ---- whole backbone fwd + bwd, training mode
bb = seg.backbone.train()
vf_g = vf.detach().clone().requires_grad_()
probe = torch.randn(m, 128, device=dev)
def step():
    for p in bb.parameters():
        p.grad = None
    out = bb(dict(voxel_feats=vf_g, voxel_coors=vc, batch_size=1))[0]['voxel_feats']
    torch.cuda.synchronize(); t1 = time.perf_counter()
    (out * probe).sum().backward()
    torch.cuda.synchronize(); return t1
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t1 = step()
    t2 = time.perf_counter()
    print(f'iter {i}: fwd {1e3 * (t1 - t0):.1f} ms  bwd {1e3 * (t2 - t1):.1f} ms')
