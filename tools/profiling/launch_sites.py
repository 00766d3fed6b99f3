"""Where do the launches of one frame come from?  (GPU box)
(A) every aten op that launches device work (kernel / memcpy / memset), grouped by the innermost fullysparsefusion_amd frame
    of its Python stack (torch.profiler, with_stack);
(B) every hip_ops (C-ABI) call, grouped by function and call site."""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import hip_ops
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
model = bench.build_model(dev)
model.test_cfg['concurrent_query_branches'] = False
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(3): bench.step(model, inp)

# ---- (B)
cnt = collections.Counter()
def site(depth_skip=2):
    for fs in reversed(traceback.extract_stack()[:-depth_skip]):
        if 'fullysparsefusion_amd' in fs.filename and 'hip_ops' not in fs.filename:
            return f"{os.path.basename(fs.filename)}:{fs.lineno} {fs.name}"
    return '?'
saved = {}
for name in dir(hip_ops):
    f = getattr(hip_ops, name)
    if callable(f) and not name.startswith('_') and getattr(f, '__module__', '') == hip_ops.__name__ and not isinstance(f, type):
        saved[name] = f
        def mk(f, name):
            def g(*a, **k):
                cnt[(name, site())] += 1
                return f(*a, **k)
            return g
        setattr(hip_ops, name, mk(f, name))
bench.step(model, inp)
for name, f in saved.items(): setattr(hip_ops, name, f)
print("== (B) hip_ops calls per frame:", sum(cnt.values()))
byf = collections.Counter()
for (n, s), c in cnt.items(): byf[n] += c
print(dict(byf.most_common()))
for (n, s), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]: print(f"{c:4d}  {n:28s} {s}")

# ---- (A)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.step(model, inp)
    torch.cuda.synchronize()
agg = collections.Counter(); agg_us = collections.Counter()
nk = 0
for e in prof.events():
    ks = getattr(e, 'kernels', None)
    if not ks: continue
    if e.cpu_parent is not None and getattr(e.cpu_parent, 'kernels', None):  # count at the outermost op that owns the kernels
        continue
    st = '?'
    for fr in (e.stack or []):
        if 'fullysparsefusion_amd' in fr and 'hip_ops' not in fr:
            st = fr.split('fullysparsefusion_amd/')[-1]; break
    key = (e.name, st)
    agg[key] += len(ks); agg_us[key] += sum(k.duration for k in ks); nk += len(ks)
print("== (A) device launches under aten ops per frame:", nk)
for key, c in sorted(agg.items(), key=lambda kv: -kv[1])[:90]:
    print(f"{c:4d} {agg_us[key]:8.1f} us  {key[0]:38s} {key[1]}")
