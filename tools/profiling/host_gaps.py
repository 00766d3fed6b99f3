"""Host timeline of one frame WITHOUT a profiler: every C-ABI call of the library on every host thread as (thread, t_in, t_out) from
time.perf_counter — a proxy around the ctypes handle, ~0.3 us per call.  Prints, for the median frame of N, the calls of the main thread
with the Python time in front of each (gap) and the time inside it (which includes its host waits).  (GPU box)
usage: python tools/profiling/host_gaps.py [sweeps] [min_gap_us]"""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from fullysparsefusion_amd import hip_ops, _lib

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
dev = torch.device("cuda:0")
model = bench.build_model(dev)
frames = [bench.make_inputs(sweeps, s, dev)[1] for s in (0, 131, 262, 393)]
log = []
real = hip_ops._L()  # (argtypes configured)


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith("fsf_") or name.endswith("_bytes") or name in ("fsf_get_option", "fsf_set_option", "fsf_status_string"):
            return fn

        def call(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            log.append((threading.get_ident(), name, t0, time.perf_counter()))
            return r
        return call


proxy = Proxy()
hip_ops._L = lambda: proxy
for i in range(3):
    bench.step(model, frames[i % 4])
per_frame = []
for i in range(8):
    torch.cuda.synchronize()
    log.clear()
    t0 = time.perf_counter()
    bench.step(model, frames[i % 4])
    t1 = time.perf_counter()
    per_frame.append((t1 - t0, t0, list(log)))
per_frame.sort(key=lambda x: x[0])
dt, t0, calls = per_frame[len(per_frame) // 2]
main = threading.get_ident()
mc = [c for c in calls if c[0] == main]
print(f"# {sweeps}-sweep frame, median of 8: {dt * 1e3:.2f} ms wall; {len(calls)} C-ABI calls ({len(mc)} on the main thread)")
print(f"# main thread: time inside C-ABI calls {sum(c[3] - c[2] for c in mc) * 1e3:.2f} ms, Python between them {(dt - sum(c[3] - c[2] for c in mc)) * 1e3:.2f} ms")
print("#   t_in_us   gap_us  inside_us  call   (gaps >= %.0f us or calls >= 100 us)" % min_gap)
prev = t0
for _, name, a, b in mc:
    gap, inside = (a - prev) * 1e6, (b - a) * 1e6
    if gap >= min_gap or inside >= 100:
        print(f"  {(a - t0) * 1e6:8.0f} {gap:8.0f} {inside:9.0f}  {name}")
    prev = b
print(f"  {dt * 1e6:8.0f} {(t0 + dt - prev) * 1e6:8.0f}            (frame end)")
