import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import hip_ops, _lib
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
seg = model.segmentor
with torch.no_grad():
    bp, coors = seg.voxelize([inp['points'][0][:, :5].contiguous()])
    vf, vc, inv = seg.voxel_encoder(bp, coors, return_inv=True)
    calls = []
    orig = hip_ops.spconv_forward
    def rec(feat, wt, nbr, **kw):
        calls.append((feat, wt, nbr, kw)); return orig(feat, wt, nbr, **kw)
    hip_ops.spconv_forward = rec
    seg.backbone(dict(voxel_feats=vf, voxel_coors=vc, batch_size=1))
    hip_ops.spconv_forward = orig
    feat, wt, nbr, kw = calls[3]   # level-2 subm 128->128
    for _ in range(3): orig(feat, wt, nbr, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(feat, wt, nbr, **kw); e1.record(); torch.cuda.synchronize()
    print('layer us', e0.elapsed_time(e1)*1e3, 'm_out', nbr.shape[0])
h = _lib.lib()
n = 1580
buf = (ctypes.c_longlong * (n*8))()
print('rc', h.fsf_debug_read(buf, n*8))
a = np.array(buf[:], dtype=np.int64).reshape(n, 8)
d = np.diff(a[:, :6], axis=1)
print('stages/tile mean', a[:,6].mean())
names = ['zero', 'rowlists', 'prologue', 'mainloop', 'epilogue']
for i, nm in enumerate(names):
    print(f'{nm:10s} mean {d[:,i].mean():10.0f} cyc  p50 {np.median(d[:,i]):10.0f}  max {d[:,i].max():10.0f}')
tot = a[:,5]-a[:,0]
print('total per tile mean', tot.mean(), 'sum/1e6', tot.sum()/1e6)
t0 = a[:,0].min(); print('kernel span cycles', a[:,5].max()-t0)
# concurrency over time
starts = np.sort(a[:,0]-t0); ends = np.sort(a[:,5]-t0)
print('start quantiles', np.quantile(starts, [0, .25, .5, .75, 1]).astype(int))
print('end quantiles', np.quantile(ends, [0, .25, .5, .75, 1]).astype(int))
print('main loop cycles per stage', (d[:,3]/np.maximum(a[:,6],1)).mean())
