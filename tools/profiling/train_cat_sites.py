"""Which forward-pass ATen ops of one TRAINING step are the big memory movers: cat / copy_ / clone / index_select / mul / add with their
shapes and the plugin frame that issued them.  usage: train_cat_sites.py  (GPU box)"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
dev = torch.device("cuda:0"); torch.manual_seed(0)
model = bench.build_model(dev)
_, inp = bench.make_inputs(10, 0, dev)
ts = bench.TrainStep(model)
for _ in range(2): ts(inp)
agg = collections.defaultdict(lambda: [0, 0])
WATCH = {"cat", "copy_", "clone", "index_select", "mul", "add", "add_", "index", "_to_copy", "contiguous", "sub", "div", "zeros", "zeros_like", "new_zeros", "fill_", "zero_"}
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "").split(".")[0]
        if name in WATCH and torch.is_tensor(out) and out.is_cuda:
            nbytes = out.numel() * out.element_size()
            if nbytes >= (1 << 20):
                st = [f for f in traceback.extract_stack() if "fullysparsefusion_amd" in f.filename and "hip_ops.py" not in f.filename]
                f = st[-1] if st else None
                where = f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" if f else "(autograd engine / optimizer)"
                agg[(name, tuple(out.shape), where)][0] += 1
                agg[(name, tuple(out.shape), where)][1] += nbytes
        return out
with Spy():
    ts(inp)
torch.cuda.synchronize()
tot = sum(v[1] for v in agg.values())
print(f"outputs >= 1 MB of the watched ops in one training step: {tot / 1e6:.0f} MB written")
for (name, shape, where), (n, b) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{b / 1e6:8.1f} MB x{n:2d}  {name:12s} {str(shape):22s} {where}")
