"""Where wave 0 of every K22-family workgroup spends its clocks (needs the profiling build: -DFSF_LNA_TIMELINE, see lna_timeline.sh).
Phases: 0 chunk work (MFMA issue + split), 1 waiting at the top of a chunk (x, weights, LDS reads, barrier), 2 epilogue: segment context +
bias + per-row addend (gather wait), 3 LayerNorm statistics, 4 affine + activation + stores (+ scan), 5 the block's slot merge."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from fullysparsefusion_amd import hip_ops as ops, _lib
dev = torch.device('cuda:0')
L = _lib.lib()
L.fsf_debug_lna_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
NAMES = ["chunk work (MFMA issue, split)", "wait at chunk top (x, W, LDS, barrier)", "epilogue: seg ctx + bias + addend gather", "LayerNorm statistics",
         "affine + act + stores (+ scan)", "slot merge"]
def measure(title, fn, rows, reps=5):
    fn(); torch.cuda.synchronize()
    L.fsf_debug_lna_timeline(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    L.fsf_debug_lna_timeline(ctypes.cast(buf, ctypes.c_void_p), 0)
    v = [int(x) for x in buf]
    wgs = max(v[8], 1)
    tot = sum(v[:6])
    blocks = reps * ((rows + 127) // 128)
    print(f"## {title}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch (instrumented build), {wgs // reps} workgroups per launch, "
          f"{tot / max(blocks, 1):.0f} clocks of wave 0 per 128-row block")
    for i, nm in enumerate(NAMES):
        if v[i]: print(f"   {100.0 * v[i] / tot:5.1f} %  {v[i] / max(blocks, 1):8.0f} clk/block  {nm}")
torch.manual_seed(0)
n, g = 510652, 10397
x = torch.randn(n, 128, device=dev); w = torch.randn(128, 128, device=dev) / 128 ** 0.5
gam = torch.rand(128, device=dev) + 0.5; bet = torch.randn(128, device=dev) * 0.1
planes = ops.linear_prepare_weight(w)
ids = torch.randint(0, g, (n,), device=dev); ids[:120000] = 17
sid = torch.sort(ids)[0]; u, sinv = torch.unique(sid, return_inverse=True)
so = torch.full((u.numel(), 128), float("-inf"), device=dev); tb = torch.randn(u.numel(), 128, device=dev)
measure("K22s grouped 128 -> 128, 510 652 rows, LN + GELU + segmented max, rows written",
        lambda: ops.linear_norm_act_segmax(x, planes, 128, sinv, so, norm='ln', gamma=gam, beta=bet, eps=1e-3, act='gelu', row_add=tb, row_add_index=sinv), n)
measure("K22s grouped, no rows written",
        lambda: ops.linear_norm_act_segmax(x, planes, 128, sinv, so, norm='ln', gamma=gam, beta=bet, eps=1e-3, act='gelu', row_add=tb, row_add_index=sinv, want_rows=False), n)
x180 = torch.randn(n, 180, device=dev); w180 = torch.randn(128, 180, device=dev) / 180 ** 0.5; p180 = ops.linear_prepare_weight(w180)
measure("K22s 180 -> 128 (first layer of a stack), rows written",
        lambda: ops.linear_norm_act_segmax(x180, p180, 128, sinv, so, norm='ln', gamma=gam, beta=bet, eps=1e-3, act='gelu'), n)
measure("K22 128 -> 128, LN + GELU, 510 652 rows (no segments)",
        lambda: ops.linear_norm_act(x, planes, 128, norm='ln', gamma=gam, beta=bet, eps=1e-3, act='gelu'), n)
n2 = 310615; x2 = torch.randn(n2, 128, device=dev)
measure("K22 128 -> 128, affine + ReLU, 310 615 rows (segmentation head)",
        lambda: ops.linear_norm_act(x2, planes, 128, norm='affine', gamma=gam, beta=bet, act='relu'), n2)
n3 = 10641; x3 = torch.randn(n3, 1024, device=dev).clamp_min(0); w3 = torch.randn(1024, 1024, device=dev) / 32
wp = ops.linear_prepare_weight_f16(w3, 128); xp = ops.rows_to_planes(x3); b3 = torch.randn(1024, device=dev)
measure("K22h 10 641 x 1024 -> 1024 (8 slices: blocks counted per slice)", lambda: ops.linear_planes_norm_act(xp, wp, 1024, 128, bias=b3), n3 * 8)
