"""Throughput with 1 / 2 / 3 frames in flight on one GPU: N model replicas, one host thread + stream per replica, frames dealt round
robin.  usage: inflight_probe.py [steps]"""
import os, sys, time, threading, copy
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
import bench
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
pool = [bench.make_inputs(10, j, dev)[1] for j in range(4)]
base = bench.build_model(dev)
for nfl in (1, 2, 3, 1, 2):
    models = [base] + [bench.build_model(dev) for _ in range(nfl - 1)]  # (same seed: the same weights)
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    def worker(t, n_steps, out):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[t]):
            for i in range(t, n_steps, nfl):
                out.append(bench.step(models[t], pool[i % 4]))
            streams[t].synchronize()
    def run(n_steps):
        outs = [[] for _ in range(nfl)]
        ths = [threading.Thread(target=worker, args=(t, n_steps, outs[t])) for t in range(nfl)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for th in ths: th.start()
        for th in ths: th.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run(6)
    dt = run(steps)
    print(f"frames in flight {nfl}: {steps / dt:6.2f} frames/s  ({dt / steps * 1e3:.2f} ms per frame of throughput)", flush=True)
    del models
