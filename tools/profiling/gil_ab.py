"""Two host threads drive the two query branches: how much of the frame is the interpreter's thread hand-over?  One process, settings
interleaved: (a) default switch interval 5 ms, (b) 0.1 ms, (c) 20 us, (d) both branches on one thread.  (GPU box)  usage: gil_ab.py [sweeps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda:0')
model = bench.build_model(dev)
pool = [bench.make_inputs(sweeps, s, dev)[1] for s in range(4)]
def run(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(k): bench.step(model, pool[i % 4])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
run(8)
settings = [("switch 5 ms", 5e-3, True), ("switch 0.1 ms", 1e-4, True), ("switch 20 us", 2e-5, True), ("one thread", 5e-3, False)]
for rep in range(3):
    for name, si, conc in settings:
        sys.setswitchinterval(si)
        model.test_cfg['concurrent_query_branches'] = conc
        run(4)
        print(f"rep {rep} {sweeps}-sweep [{name:14s}] {run(30):7.3f} ms", flush=True)
