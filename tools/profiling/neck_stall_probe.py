"""What holds the main stream for ~150 us between the U-Net's last kernel and the neck (voxel2point)?  The idle in front of voxel2point under
torch.profiler, for the side-stream switches one at a time.  (GPU box)"""
import json, os, sys, tempfile, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from fullysparsefusion_amd import switches
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
frames = [bench.make_inputs(10, s, dev)[1] for s in range(2)]
def gap(tag):
    model = bench.build_model(dev)
    for i in range(4): bench.step(model, frames[i % 2])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        bench.step(model, frames[0]); bench.step(model, frames[1]); torch.cuda.synchronize()
    path = os.path.join(tempfile.mkdtemp(), 't.json'); prof.export_chrome_trace(path)
    ev = [e for e in json.load(open(path))['traceEvents'] if e.get('ph') == 'X' and e.get('cat') in ('kernel', 'gpu_memcpy', 'gpu_memset')]
    ev.sort(key=lambda e: e['ts'])
    out = []
    end = 0
    for e in ev:
        if 'voxel2point' in e['name']: out.append(round(e['ts'] - end, 1))
        end = max(end, e['ts'] + e['dur'])
    print(f"{tag:40s} idle in front of voxel2point: {out} us", flush=True)
gap("default")
for name in ("UNET_LATERAL_STREAM", "UNET_PLAN_STREAM", "UNET_MASK_ORDER"):
    setattr(switches, name, False); gap(name + "=0"); setattr(switches, name, True)
switches.UNET_LATERAL_STREAM = switches.UNET_PLAN_STREAM = False
gap("both side streams off")
