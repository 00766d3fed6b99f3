import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
def run(n=30):
    for _ in range(5): bench.step(model, inp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): bench.step(model, inp)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    for rep in range(3):
        for flag in (False, True):
            model.test_cfg['concurrent_query_branches'] = flag
            print('concurrent', flag, round(run(), 3), 'ms')
