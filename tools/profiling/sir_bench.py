"""K21 (sir_input), K22 (fused Linear + LN + GELU, plain and grouped) and the segmented max at the LiDAR-query SIR stack's shapes
(510 k points, 10 k groups) — one line per kernel; run once per library build (FSF_LIB_PATH=...) for a same-box A/B."""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.environ.get('FSF_ROOT') or os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')))
from fullysparsefusion_amd import hip_ops as ops
dev = torch.device('cuda:0')
def t(f, it=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
torch.manual_seed(0)
n, g = int(os.environ.get("N", 510652)), 10397
out = []
for c_in, fc in [(180, 175), (133, 128)]:
    pts = torch.randn(n, 5, device=dev); feats = torch.randn(n, fc, device=dev); fcl = torch.randn(n, 3, device=dev)
    mk = lambda o, i: (torch.randn(o, i, device=dev) / i ** 0.5, torch.rand(o, device=dev) + 0.5, torch.randn(o, device=dev) * 0.1)
    layers = (mk(16, 3), mk(32, 16), mk(c_in, 32), 1e-3)
    out.append(f"sir_input c={c_in}: {t(lambda: ops.sir_input(pts, feats, fcl, [20., 20., 4.], layers, 'gelu', 10.0)):7.1f} us")
    x = ops.sir_input(pts, feats, fcl, [20., 20., 4.], layers, 'gelu', 10.0)
    w = torch.randn(128, c_in, device=dev) / c_in ** 0.5
    gam = torch.rand(128, device=dev) + 0.5; bet = torch.randn(128, device=dev) * 0.1
    planes = ops.linear_prepare_weight(w)
    out.append(f"K22 k={c_in}->128 ln gelu: {t(lambda: ops.linear_norm_act(x, planes, 128, norm='ln', gamma=gam, beta=bet, eps=1e-3, act='gelu')):7.1f} us")
x = torch.randn(n, 128, device=dev)
w = torch.randn(128, 128, device=dev) / 128 ** 0.5
planes = ops.linear_prepare_weight(w)
inv = torch.sort(torch.randint(0, g, (n,), device=dev))[0]
table = torch.randn(g, 128, device=dev)
out.append(f"K22 grouped k=128->128: {t(lambda: ops.linear_norm_act(x, planes, 128, norm='ln', gamma=gam, beta=bet, eps=1e-3, act='gelu', row_add=table, row_add_index=inv)):7.1f} us")
ids = torch.randint(0, g, (n,), device=dev)
ids[:120000] = 17
plan = ops.segment_plan_from_inverse(ids, g)
out.append(f"segment max [n,128] (random order, one 120 k segment): {t(lambda: ops.segment_reduce(x, plan, 'max')):7.1f} us")
if hasattr(ops, "linear_norm_act_segmax"):
    sid = torch.sort(ids)[0]
    u, sinv = torch.unique(sid, return_inverse=True)
    splan = ops.segment_plan_from_inverse(sinv, u.numel())
    so = torch.full((u.numel(), 128), float("-inf"), device=dev)
    tb = torch.randn(u.numel(), 128, device=dev)
    out.append(f"K22s grouped k=128->128 + segmax (sorted rows): {t(lambda: ops.linear_norm_act_segmax(x, planes, 128, sinv, so, norm='ln', gamma=gam, beta=bet, eps=1e-3, act='gelu', row_add=tb, row_add_index=sinv)):7.1f} us")
    out.append(f"K22s (no rows written): {t(lambda: ops.linear_norm_act_segmax(x, planes, 128, sinv, so, norm='ln', gamma=gam, beta=bet, eps=1e-3, act='gelu', row_add=tb, row_add_index=sinv, want_rows=False)):7.1f} us")
    out.append(f"segment max on sorted rows (plan path): {t(lambda: ops.segment_reduce(x, splan, 'max')):7.1f} us")
xn = torch.randn(n, 1024, device=dev)[:20000]
gam2 = torch.rand(1024, device=dev); bet2 = torch.randn(1024, device=dev)
out.append(f"norm_act 20000 x 1024 ln gelu: {t(lambda: ops.norm_act(xn, gam2, bet2, 1e-3, 'ln', 'gelu')):7.1f} us")
print(" | ".join(out))
