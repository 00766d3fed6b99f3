"""Where do the host syncs (nonzero / item / boolean-mask indexing) of one frame come from?  (GPU box)"""
import os, sys, torch, collections, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
model.test_cfg['concurrent_query_branches'] = False
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(2): bench.step(model, inp)
cnt = collections.Counter()
def site():
    for fs in reversed(traceback.extract_stack()[:-2]):
        if 'fullysparsefusion_amd' in fs.filename:
            return f"{os.path.basename(fs.filename)}:{fs.lineno} {fs.name}"
    return '?'
orig_nz = torch.nonzero
def nz(*a, **k):
    cnt[('nonzero', site())] += 1; return orig_nz(*a, **k)
torch.nonzero = nz
orig_tnz = torch.Tensor.nonzero
torch.Tensor.nonzero = lambda self, *a, **k: (cnt.update({('nonzero', site()): 1}), orig_tnz(self, *a, **k))[1]
orig_item = torch.Tensor.item
torch.Tensor.item = lambda self: (cnt.update({('item', site()): 1}), orig_item(self))[1]
orig_gi = torch.Tensor.__getitem__
def gi(self, idx):
    def isb(i): return torch.is_tensor(i) and i.dtype == torch.bool
    if isb(idx) or (isinstance(idx, tuple) and any(isb(i) for i in idx)): cnt[('bool-index', site())] += 1
    return orig_gi(self, idx)
torch.Tensor.__getitem__ = gi
orig_si = torch.Tensor.__setitem__
def si(self, idx, val):
    def isb(i): return torch.is_tensor(i) and i.dtype == torch.bool
    if self.is_cuda and (isb(idx) or (isinstance(idx, tuple) and any(isb(i) for i in idx))): cnt[('bool-setitem', site())] += 1
    elif self.is_cuda and (torch.is_tensor(idx) or (isinstance(idx, tuple) and any(torch.is_tensor(i) for i in idx))): cnt[('index-setitem', site())] += 1
    return orig_si(self, idx, val)
torch.Tensor.__setitem__ = si
for nm in ('__int__', '__bool__', '__float__', '__index__'):
    o = getattr(torch.Tensor, nm)
    def mk(o, nm):
        def f(self, *a):
            if self.is_cuda: cnt[(nm, site())] += 1
            return o(self, *a)
        return f
    setattr(torch.Tensor, nm, mk(o, nm))
orig_cpu = torch.Tensor.cpu
torch.Tensor.cpu = lambda self, *a, **k: (cnt.update({('cpu', site()): 1}) if self.is_cuda else None, orig_cpu(self, *a, **k))[1]
orig_tolist = torch.Tensor.tolist
torch.Tensor.tolist = lambda self: (cnt.update({('tolist', site()): 1}) if self.is_cuda else None, orig_tolist(self))[1]
bench.step(model, inp)
tot = collections.Counter()
for (k, s), c in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f'{k:12s} {c:3d}  {s}'); tot[k] += c
print(dict(tot))
