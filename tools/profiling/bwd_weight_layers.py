"""K10 (weight gradient) alone on the distinct conv layers of the 10-sweep frame.  usage: bwd_weight_layers.py  (GPU box;
FSF_BWD_TARGET_WGS=<n> overrides the workgroup target of the pair-range split)"""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
from fullysparsefusion_amd.mmdet3d_plugin.ops import spconv as sp
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
calls = []
orig = sp.SparseConvolution.forward
def spy(self, x, scale=None, shift=None, residual=None, relu=False):
    out = orig(self, x, scale=scale, shift=shift, residual=residual, relu=relu)
    calls.append((self, x.features, self._rulebook(x), out.features.size(0)))
    return out
sp.SparseConvolution.forward = spy
with torch.no_grad(): model.segmentor.extract_feat([inp["points"][0][:, :5].contiguous()], None)
sp.SparseConvolution.forward = orig
def t(f, it=20):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
seen, total = set(), 0.0
for m, feat, rb, m_out in calls:
    key = (feat.size(0), m_out, m.in_channels, m.out_channels, m.inverse)
    pairs, num = rb.pairs(m.inverse)
    grad = torch.randn(m_out, m.out_channels, device=dev)
    us = t(lambda: hip_ops.spconv_backward_weight(feat, grad, pairs, num))
    total += us
    if key in seen: continue
    seen.add(key)
    npairs = int(num.sum())
    print(f"m_in={feat.size(0):7d} m_out={m_out:7d} {m.in_channels:4d}->{m.out_channels:4d} pairs={npairs:8d}  {us:7.1f} us  {2 * npairs * m.in_channels * m.out_channels / us / 1e6:6.1f} TF")
print(f"sum over the {len(calls)} layers: {total / 1e3:.2f} ms")
