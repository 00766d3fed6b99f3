import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
cap = {}
orig = hip_ops.connected_components_grouped
def rec(*a, **k):
    cap['a'] = a; cap['k'] = k; return orig(*a, **k)
hip_ops.connected_components_grouped = rec
with torch.no_grad(): bench.step(model, inp)
a, k = cap['a'], cap['k']
for _ in range(3): orig(*a, **k)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): r = orig(*a, **k)
e1.record(); torch.cuda.synchronize()
print('ccl grouped call us', e0.elapsed_time(e1) * 100, 'n', a[0].shape[0], 'components', r[1] if isinstance(r, tuple) else None)
import ctypes
from fullysparsefusion_amd import _lib
h = _lib.lib()
if hasattr(h, 'fsf_debug_read_ccl'):
    buf = (ctypes.c_ulonglong * 8)()
    h.fsf_debug_read_ccl(buf, 1)
    orig(*a, **k); torch.cuda.synchronize()
    h.fsf_debug_read_ccl(buf, 1)
    print('tile pairs processed', buf[0], 'links', buf[1], 'unions attempted', buf[2])
