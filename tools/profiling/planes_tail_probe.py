"""How much of a K9c launch is the tail of its last round of workgroups?  The same layer with the neighbour table cut to m output rows:
time per row block as the block count crosses multiples of the resident workgroup slots (2 per CU x 256).
usage (GPU box): python tools/profiling/planes_tail_probe.py [layer indices]"""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
from fullysparsefusion_amd.mmdet3d_plugin.ops import spconv as sp
want = [int(a) for a in sys.argv[1:]] or [2, 4]
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
calls = []
orig = sp.SparseConvolution.forward
def spy(self, x, scale=None, shift=None, residual=None, relu=False):
    out = orig(self, x, scale=scale, shift=shift, residual=residual, relu=relu)
    if self.subm:
        calls.append((self, x.features, self._rulebook(x).nbr, dict(scale=scale, shift=shift, residual=residual, relu=relu)))
    return out
sp.SparseConvolution.forward = spy
with torch.no_grad(): model.segmentor.extract_feat([inp["points"][0][:, :5].contiguous()], None)
sp.SparseConvolution.forward = orig
def t(f, it=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
for i in want:
    m, feat, nbr, kw = calls[i]
    kvol = math.prod(m.kernel_size); cin, cout = m.in_channels, m.out_channels
    wpl = hip_ops.spconv_prepare_weight_planes(m.weight.detach().reshape(kvol, cin, cout))
    srcs = [hip_ops.to_planes(feat)] if cin <= 128 else [hip_ops.to_planes(feat[:, :cin // 2]), hip_ops.to_planes(feat[:, cin // 2:])]
    kw = dict(kw); res = kw.pop("residual", None)
    print(f"L{i} m={nbr.shape[0]} {cin}->{cout}")
    for blocks in [128, 256, 384, 512, 520, 576, 640, 768, 1024, 1032, 1280, 1536, 1544, 2048]:
        rows = min(blocks * 64, nbr.shape[0])
        nb = nbr[:rows].contiguous()
        r = res[:rows].contiguous() if res is not None else None
        us = t(lambda: hip_ops.spconv_forward_planes(srcs, wpl, kvol, cout, nb, want_planes=True, residual=r, **kw))
        print(f"   blocks {math.ceil(rows / 64):5d}  rows {rows:7d}  {us:8.1f} us   {us / math.ceil(rows / 64) * 512:8.1f} us per 512 blocks")
        if rows == nbr.shape[0]: break
