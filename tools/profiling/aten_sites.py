"""Which Python lines issue the aten ops of one frame?  TorchDispatchMode sees every aten call; each is attributed to the
innermost fullysparsefusion_amd frame of the Python stack.  (GPU box)  usage: python tools/profiling/aten_sites.py [top]"""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device('cuda:0')
model = bench.build_model(dev)
model.test_cfg['concurrent_query_branches'] = False
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(3): bench.step(model, inp)
cnt = collections.Counter()
VIEW = {'view', 'slice', 'select', 'unsqueeze', 'squeeze', 'expand', 't', 'transpose', 'permute', 'alias', 'detach', 'as_strided',
        '_unsafe_view', 'reshape', 'empty', 'empty_strided', 'size', 'stride', 'lift_fresh', 'unbind', 'split', 'narrow', 'new_empty',
        'empty_like', 'unfold', 'sym_size', 'sym_stride', 'sym_numel', 'is_pinned', 'split_with_sizes', 'diagonal', 'resize_'}
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        if name.split('.')[0] not in VIEW:
            st = '?'
            for fs in reversed(traceback.extract_stack()[:-1]):
                if 'fullysparsefusion_amd' in fs.filename:
                    st = f"{os.path.basename(fs.filename)}:{fs.lineno} {fs.name}"; break
            cnt[(st, name)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    bench.step(model, inp)
torch.cuda.synchronize()
print("aten ops (non-view) per frame:", sum(cnt.values()))
bysite = collections.Counter()
for (s, n), c in cnt.items(): bysite[s] += c
top = int(sys.argv[1]) if len(sys.argv) > 1 else 70
for s, c in bysite.most_common(top):
    ops = ', '.join(f"{n}x{k}" if k > 1 else n for (ss, n), k in sorted(cnt.items(), key=lambda kv: -kv[1]) if ss == s)
    print(f"{c:4d}  {s:55s} {ops[:150]}")
