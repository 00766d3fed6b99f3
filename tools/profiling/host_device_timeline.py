"""Host launch time against device start time for every dispatch of one frame (torch.profiler's chrome trace: the runtime's
hipLaunchKernel / hipMemcpyAsync / hipMemsetAsync events carry the correlation id of the device activity they start).
For each device activity: when the host issued it, when the device started it, how long it sat in the queue, how long the device
had been idle before it — an idle window in front of an activity that did NOT wait in the queue is the host's (Python between two
launches, or a host wait); one in front of an activity that did is the device's own (stream dependency).  (GPU box)
usage: host_device_timeline.py [sweeps] [--serial] [--list]"""
import collections, json, os, sys, tempfile, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from torch.profiler import profile, ProfilerActivity
args = [a for a in sys.argv[1:] if not a.startswith('--')]
sweeps = int(args[0]) if args else 10
dev = torch.device('cuda:0')
model = bench.build_model(dev)
if '--serial' in sys.argv:
    model.test_cfg['concurrent_query_branches'] = False
frames = [bench.make_inputs(sweeps, s, dev)[1] for s in range(2)]
for i in range(4): bench.step(model, frames[i % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    bench.step(model, frames[0])
    bench.step(model, frames[1])
    torch.cuda.synchronize()
path = os.path.join(tempfile.mkdtemp(), 'trace.json')
prof.export_chrome_trace(path)
ev = json.load(open(path))['traceEvents']
launch, acts = {}, []
for e in ev:
    if e.get('ph') != 'X': continue
    cat, a = e.get('cat', ''), e.get('args', {})
    c = a.get('correlation')
    if c is None: continue
    if cat in ('cuda_runtime', 'cuda_driver'):
        launch[c] = (e['ts'], e['dur'], e['name'], e.get('tid'))
    elif cat in ('kernel', 'gpu_memcpy', 'gpu_memset'):
        acts.append((e['ts'], e['dur'], e['name'], c, a.get('stream', e.get('tid'))))
acts.sort()
# second frame only: from the first activity launched after the first frame's last one
t_mid = None
# find the boundary: the largest host-side gap is not reliable; use the voxelize kernel's second occurrence
vox = [a for a in acts if 'voxelize_dynamic' in a[2]]
t0 = vox[-1][0] - 400 if len(vox) >= 2 else acts[0][0]
acts = [a for a in acts if a[0] >= t0]
busy_end = acts[0][0]
rows, host_idle, dev_idle, nsmall = [], 0.0, 0.0, 0
by_site = collections.Counter()
for ts, dur, name, c, st in acts:
    l = launch.get(c)
    q = ts - (l[0] + l[1]) if l else float('nan')
    idle = max(0.0, ts - busy_end)
    kind = ''
    if idle > 3:
        if l and q < 25: host_idle += idle; kind = 'HOST'
        else: dev_idle += idle; kind = 'dev'
    rows.append((ts - t0, st, idle, kind, q, dur, name[:70]))
    busy_end = max(busy_end, ts + dur)
    if dur <= 15: nsmall += 1
span = busy_end - t0
print(f"# {len(acts)} device activities in the frame, span {span:.0f} us; idle in front of activities that did not queue (host-bound) "
      f"{host_idle:.0f} us, in front of queued ones {dev_idle:.0f} us; {nsmall} activities <= 15 us")
big = sorted([r for r in rows if r[2] > 15], key=lambda r: -r[2])[:40]
print("# largest idle windows: t_us stream idle_us kind queue_us dur_us name")
for r in big: print(f"{r[0]:9.1f} s{r[1]} idle {r[2]:7.1f} {r[3]:4s} queued {r[4]:8.1f} dur {r[5]:7.1f}  {r[6]}")
if '--list' in sys.argv:
    print("# every activity")
    for r in rows: print(f"{r[0]:9.1f} s{r[1]} idle {r[2]:7.1f} {r[3]:4s} queued {r[4]:8.1f} dur {r[5]:7.1f}  {r[6]}")
