"""Library GEMMs (aten mm / addmm / bmm) left in one frame: shapes and call sites.  (GPU box)"""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device('cuda:0')
model = bench.build_model(dev)
model.test_cfg['concurrent_query_branches'] = False
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(3): bench.step(model, inp)
rows = []
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ('aten.mm', 'aten.addmm', 'aten.bmm', 'aten.linear', 'aten.matmul')):
            st = [f"{os.path.basename(fs.filename)}:{fs.lineno} {fs.name}" for fs in traceback.extract_stack()[:-1] if 'fullysparsefusion_amd' in fs.filename][-3:]
            rows.append((name, [tuple(a.shape) for a in args if torch.is_tensor(a)], ' < '.join(reversed(st))))
        return func(*args, **(kwargs or {}))
with Spy():
    bench.step(model, inp)
for r in rows: print(r[0], r[1], r[2])
