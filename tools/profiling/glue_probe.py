"""Largest torch glue ops of a frame (cat / index_select / gather / boolean & long indexing): shapes, time, call sites."""
import os, sys, torch, collections, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
model.test_cfg['concurrent_query_branches'] = False
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(2): bench.step(model, inp)
rec = []
def site():
    for fs in reversed(traceback.extract_stack()[:-2]):
        if 'fullysparsefusion_amd' in fs.filename and 'hip_ops' not in fs.filename:
            return f"{os.path.basename(fs.filename)}:{fs.lineno} {fs.name}"
    return '?'
def wrap(mod, name, label, shape_of):
    f = getattr(mod, name)
    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(*a, **k); e1.record()
        try: rec.append((label, shape_of(a, k, r), site(), e0, e1))
        except Exception: pass
        return r
    setattr(mod, name, g)
wrap(torch, 'cat', 'cat', lambda a, k, r: tuple(r.shape))
wrap(torch.Tensor, 'index_select', 'index_select', lambda a, k, r: tuple(r.shape))
wrap(torch.Tensor, '__getitem__', 'getitem', lambda a, k, r: tuple(r.shape) if torch.is_tensor(r) else ())
wrap(torch.Tensor, 'repeat', 'repeat', lambda a, k, r: tuple(r.shape))
wrap(torch.Tensor, 'contiguous', 'contiguous', lambda a, k, r: tuple(r.shape))
wrap(torch.Tensor, 'float', 'float', lambda a, k, r: tuple(r.shape))
wrap(torch.Tensor, 'softmax', 'softmax', lambda a, k, r: tuple(r.shape))
wrap(torch.Tensor, 'sum', 'sum', lambda a, k, r: tuple(a[0].shape))
wrap(torch.Tensor, 'max', 'max', lambda a, k, r: tuple(a[0].shape))
bench.step(model, inp)
torch.cuda.synchronize()
rows = [(e0.elapsed_time(e1) * 1e3, l, s, st) for l, s, st, e0, e1 in rec]
rows.sort(reverse=True)
tot = collections.Counter()
for us, l, s, st in rows: tot[l] += us
print({k: round(v) for k, v in tot.items()})
for us, l, s, st in rows[:40]: print(f'{us:8.1f} us  {l:13s} {str(s):22s} {st}')
