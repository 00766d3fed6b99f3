"""Where one frame makes the host wait for the device: every `.item()` / `bool()` / `nonzero` / boolean-mask index / device->host copy
(ATen level) and every C-ABI call that reads a count back, with the Python frame that asked for it.  usage: sync_sites.py [sweeps]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from fullysparsefusion_amd import _lib, hip_ops

dev = torch.device("cuda:0")
model = bench.build_model(dev)
_, inp = bench.make_inputs(int(sys.argv[1]) if len(sys.argv) > 1 else 10, 0, dev)
model.test_cfg["concurrent_query_branches"] = False
for _ in range(2):
    bench.step(model, inp)
sites = collections.Counter()

def site(tag):
    st = [f for f in traceback.extract_stack() if "fullysparsefusion_amd" in f.filename and "hip_ops.py" not in f.filename and "_lib.py" not in f.filename]
    f = st[-1] if st else traceback.extract_stack()[-4]
    sites[(tag, f"{os.path.basename(f.filename)}:{f.lineno} {f.name}")] += 1

class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "").split(".")[0]
        a0 = args[0] if args else None
        if name in ("nonzero", "masked_select") or (name == "_local_scalar_dense" and torch.is_tensor(a0) and a0.is_cuda):
            site(name)  # (.item() on a host tensor waits for nothing)
        elif name in ("_to_copy", "copy_"):
            src = args[1] if name == "copy_" else args[0]
            dst_dev = (args[0].device if name == "copy_" else (kwargs or {}).get("device", None))
            if torch.is_tensor(src) and src.is_cuda and dst_dev is not None and torch.device(dst_dev).type == "cpu":
                site("d2h copy")
        return func(*args, **(kwargs or {}))

# the library counts its own stream waits (FSF_OPT_HOST_WAITS = 2): the difference across a C-ABI call is what that call waited
orig = _lib.check
state = dict(last=int(_lib.lib().fsf_get_option(2)))
def check(status, what):
    now = int(_lib.lib().fsf_get_option(2))
    for _ in range(now - state["last"]):
        site("C-ABI read-back " + what)
    state["last"] = now
    return orig(status, what)
_lib.check = hip_ops.check = check
with Spy():
    bench.step(model, inp)
_lib.check = hip_ops.check = orig
print(f"# host waits in one frame: {sum(sites.values())}")
for (tag, where), c in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{c:3d}  {tag:40s} {where}")
