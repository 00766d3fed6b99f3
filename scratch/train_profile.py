import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
ts = bench.TrainStep(model)
for _ in range(3): ts(inp)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as p:
    ts(inp); torch.cuda.synchronize()
print(p.key_averages().table(sort_by='cuda_time_total', row_limit=40, max_name_column_width=70))
