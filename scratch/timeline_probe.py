"""Workgroup timeline of one spconv layer (build with -DFSF_ABL_TIMING): phases, concurrency over time, tail loss."""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import hip_ops, _lib
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
seg = model.segmentor
layers = [int(v) for v in sys.argv[1:]] or [3]
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
h = _lib.lib()
NB = 4096
for li in layers:
    feat, wt, nbr, kw = calls[li]
    with torch.no_grad():
        for _ in range(3): orig(feat, wt, nbr, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(feat, wt, nbr, **kw); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    buf = (ctypes.c_longlong * (NB * 8))()
    h.fsf_debug_read(buf, NB * 8)
    a = np.array(buf[:], dtype=np.int64).reshape(NB, 8)
    m_out = nbr.shape[0]
    # only blocks of this launch: stamps within the last launch window
    t_end = a[:, 5].max()
    live = (a[:, 5] > t_end - int(us * 100 * 3)) & (a[:, 0] > 0)
    a = a[live]
    t0 = a[:, 0].min()
    T = (a[:, 5].max() - t0) / 100.0  # us  (100 MHz counter)
    d = np.diff(a[:, :6], axis=1) / 100.0
    print(f'=== layer {li}: m_out {m_out} cin {wt.shape[2]} cout {wt.shape[1]} event {us:.1f} us, stamped blocks {len(a)}, span {T:.1f} us, stages/blk {a[:,6].mean():.1f}')
    for i, nm in enumerate(['rowlists', 'zero', 'prologue', 'mainloop', 'epilogue']):
        print(f'   {nm:9s} mean {d[:, i].mean():7.2f} us  p50 {np.median(d[:, i]):7.2f}  max {d[:, i].max():7.2f}')
    dur = (a[:, 5] - a[:, 0]) / 100.0
    print(f'   block duration mean {dur.mean():.1f} p10 {np.quantile(dur,.1):.1f} p50 {np.median(dur):.1f} p90 {np.quantile(dur,.9):.1f} max {dur.max():.1f} us')
    print(f'   sum of block durations / span = avg concurrency {dur.sum() / T:.1f} (512 slots)')
    # concurrency over time
    grid = np.linspace(0, T, 21)
    st, en = (a[:, 0] - t0) / 100.0, (a[:, 5] - t0) / 100.0
    conc = [int(((st <= t) & (en > t)).sum()) for t in grid]
    print('   concurrency at 5% steps:', conc)
    per_stage = d[:, 3] / np.maximum(a[:, 6], 1)
    print(f'   main loop per stage mean {per_stage.mean():.3f} us')
    xcc = (a[:, 7] >> 32) & 0xf
    print('   blocks per XCC', np.bincount(xcc, minlength=8).tolist())
