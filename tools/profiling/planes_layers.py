"""Per-layer timing of K9c (pre-split f16 planes) against K9b (split bf16) on the submanifold layers of the 10-sweep frame's
U-Net: every SubM launch of one forward is captured with its tensors and replayed through both kernels.
usage (GPU box): python tools/profiling/planes_layers.py > gpurun_out/planes_layers.txt"""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
from fullysparsefusion_amd.mmdet3d_plugin.ops import spconv as sp
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(2): bench.step(model, inp)
calls = []
orig = sp.SparseConvolution.forward
def spy(self, x, scale=None, shift=None, residual=None, relu=False):
    out = orig(self, x, scale=scale, shift=shift, residual=residual, relu=relu)
    if self.subm:
        calls.append((self, x.features, self._rulebook(x).nbr, dict(scale=scale, shift=shift, residual=residual, relu=relu)))
    return out
sp.SparseConvolution.forward = spy
with torch.no_grad(): bench.step(model, inp, hot_path_only=True)
sp.SparseConvolution.forward = orig
def t(f, it=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
tot = [0.0, 0.0, 0.0]
print(f"{'i':>3} {'m_out':>7} {'cin':>5} {'cout':>5} {'p/out':>6} {'K9b us':>9} {'K9c us':>9} {'toplanes':>9} {'maxdiff/scale':>13}  TF/s(K9c)")
for i, (m, feat, nbr, kw) in enumerate(calls):
    kvol = math.prod(m.kernel_size); cin, cout = m.in_channels, m.out_channels
    w = m.weight.detach().reshape(kvol, cin, cout)
    pairs = float((nbr >= 0).sum())
    b = t(lambda: hip_ops.spconv_forward_split(feat, m._weight_split(), kvol, cout, nbr, **kw))
    cins = [cin] if cin <= 128 else [cin // 2, cin - cin // 2]
    if hip_ops.spconv_planes_supported(cins, cout, kvol):
        wpl = hip_ops.spconv_prepare_weight_planes(w)
        srcs = [hip_ops.to_planes(feat)] if cin <= 128 else [hip_ops.to_planes(feat[:, :cin // 2]), hip_ops.to_planes(feat[:, cin // 2:])]
        c = t(lambda: hip_ops.spconv_forward_planes(srcs, wpl, kvol, cout, nbr, want_planes=cout <= 128, **kw))
        tp = t(lambda: hip_ops.to_planes(feat[:, :min(cin, 128)]))
        ref = hip_ops.spconv_forward_split(feat, m._weight_split(), kvol, cout, nbr, **kw)
        got = hip_ops.spconv_forward_planes(srcs, wpl, kvol, cout, nbr, **kw)[0]
        d = float((ref - got).abs().max()) / max(1.0, float(ref.abs().max()))
        tf = 2 * pairs * cin * cout / c / 1e6
    else:
        c, tp, d, tf = float('nan'), float('nan'), float('nan'), float('nan')
    tot[0] += b; tot[1] += c if c == c else b; tot[2] += min(b, c) if c == c else b
    xp_us = xp_conv = float('nan')
    if hasattr(hip_ops, "spconv_forward_split_planes") and hip_ops.spconv_split_planes_supported(cin, cout) and hip_ops.rows_to_planes_supported(feat):
        w16 = hip_ops.spconv_prepare_weight_split_f16(w)
        xpl = hip_ops.rows_to_planes(feat)
        xp_us = t(lambda: hip_ops.spconv_forward_split_planes(xpl, w16, kvol, cout, nbr, **kw))
        xp_conv = t(lambda: hip_ops.rows_to_planes(feat))
    print(f"{i:3d} {nbr.shape[0]:7d} {cin:5d} {cout:5d} {pairs / nbr.shape[0]:6.2f} {b:9.1f} {c:9.1f} {tp:9.1f} {d:13.2e}  {tf:7.1f}   K9b-XP {xp_us:7.1f} + planes {xp_conv:5.1f}")
print('total K9b', round(tot[0], 1), 'K9c-where-supported', round(tot[1], 1), 'best-of', round(tot[2], 1))
