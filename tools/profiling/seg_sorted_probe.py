"""Would group-sorted rows help the SIR stack's kernels?  The 510 k-row segmented max / group->point gather / grouped K22 of the
frame, as they are and with the rows permuted into segment order (plan.order = identity).  (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); model.test_cfg['concurrent_query_branches'] = False
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(2): bench.step(model, inp)
calls = []
orig = hip_ops.segment_reduce
def rec(*a, **k):
    out = orig(*a, **k); calls.append((a, k)); return out
hip_ops.segment_reduce = rec
with torch.no_grad(): bench.step(model, inp)
hip_ops.segment_reduce = orig
def t(f, it=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
a, k = max(calls, key=lambda c: c[0][0].shape[0] * c[0][0].shape[1])
feat, plan = a[0].contiguous(), a[1]
n = feat.shape[0]
order = plan.order.long()
feat_s = feat[order].contiguous()
inv_s = plan.inv[order].contiguous()
plan_s = hip_ops.SegmentPlan(inv=inv_s, order=torch.arange(n, device=dev, dtype=torch.int32), seg_offsets=plan.seg_offsets, m=plan.m, cnt=plan.cnt)
o1 = orig(feat, plan, 'max'); o2 = orig(feat_s, plan_s, 'max')
assert torch.equal(o1, o2)
print(f"n={n} m={plan.m}  seg max: unsorted {t(lambda: orig(feat, plan, 'max')):.1f} us   sorted {t(lambda: orig(feat_s, plan_s, 'max')):.1f} us")
g = o1
print(f"gather_rows(group -> point): unsorted {t(lambda: hip_ops.gather_rows(g, plan.inv)):.1f} us   sorted {t(lambda: hip_ops.gather_rows(g, inv_s)):.1f} us")
w = torch.randn(128, 128, device=dev) / 11
planes = hip_ops.linear_prepare_weight(w)
gam, bet = torch.ones(128, device=dev), torch.zeros(128, device=dev)
f1 = lambda: hip_ops.linear_norm_act(feat, planes, 128, norm='ln', gamma=gam, beta=bet, eps=1e-3, act='gelu', row_add=g, row_add_index=plan.inv)
f2 = lambda: hip_ops.linear_norm_act(feat_s, planes, 128, norm='ln', gamma=gam, beta=bet, eps=1e-3, act='gelu', row_add=g, row_add_index=inv_s)
print(f"grouped K22: unsorted {t(f1):.1f} us   sorted {t(f2):.1f} us")
