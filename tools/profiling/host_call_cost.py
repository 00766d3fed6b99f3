"""Host cost per call of the hot wrappers (tiny inputs: the device is never the bound): wall time of N back-to-back calls / N.  (GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from fullysparsefusion_amd import hip_ops as ops, _lib
dev = torch.device("cuda:0")
def cost(name, fn, n=3000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{name:44s} {(t1 - t0) / n * 1e6:7.2f} us / call")
x = torch.randn(256, 128, device=dev); w = torch.randn(128, 128, device=dev); g = torch.ones(128, device=dev); b = torch.zeros(128, device=dev)
planes = ops.linear_prepare_weight(w)
idx = torch.arange(256, device=dev)
cost("torch.empty((256,128))", lambda: torch.empty((256, 128), dtype=torch.float32, device=dev))
cost("x.data_ptr()", lambda: x.data_ptr())
cost("_lib.ptr(x)", lambda: _lib.ptr(x))
cost("_lib.stream_ptr()", lambda: _lib.stream_ptr())
cost("torch.cuda.current_stream()", lambda: torch.cuda.current_stream())
cost("ops.linear_norm_act (ln, gelu)", lambda: ops.linear_norm_act(x, planes, 128, bias=b, norm="ln", gamma=g, beta=b, eps=1e-3, act="gelu"))
cost("ops.norm_act", lambda: ops.norm_act(x, g, b, 1e-3, "ln", "gelu", inplace=False))
cost("ops.gather_rows", lambda: ops.gather_rows(x, idx))
cost("torch.cat([x, x], 1)", lambda: torch.cat([x, x], 1))
cost("x.index_select(0, idx)", lambda: x.index_select(0, idx))
cost("x + x", lambda: x + x)
cost("x.record_stream(stream)", (lambda s: (lambda: x.record_stream(s)))(torch.cuda.current_stream()))
lin = torch.nn.Linear(128, 128).to(dev)
cost("module attribute (lin.weight)", lambda: lin.weight)
from fullysparsefusion_amd.mmdet3d_plugin.ops import sst_ops
cost("sst_ops.linear_norm_act(lin, LN, gelu)", lambda: sst_ops.linear_norm_act(x, lin, torch.nn.LayerNorm(128).to(dev) if False else ln, "gelu")) if False else None
