"""Where the HOST time of one forward goes (cProfile, GPU running asynchronously)."""
import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(3):
    bench.step(model, inp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    bench.step(model, inp)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'per step: host issue {t_issue / 5 * 1e3:.1f} ms, wall {t_all / 5 * 1e3:.1f} ms')
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    bench.step(model, inp)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
