import sys, time, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
pool = [bench.make_inputs(10, seed=j, device=dev)[1] for j in range(4)]
ts = []
for i in range(45):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.step(model, pool[i % 4], False, pool[(i + 1) % 4])
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.2f}" for t in ts))
st = torch.cuda.memory_stats()
print("num_alloc_retries", st["num_alloc_retries"], "segments", st["segment.all.allocated"], "reserved GB", st["reserved_bytes.all.current"] / 1e9)
