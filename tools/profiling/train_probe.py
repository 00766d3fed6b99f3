"""Which ATen / library ops carry the training step: torch.profiler with shapes, top ops by device time."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from torch.profiler import ProfilerActivity, profile

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
torch.manual_seed(0)
model = bench.build_model(dev)
_, inp = bench.make_inputs(10, 0, dev)
ts = bench.TrainStep(model)
for _ in range(3):
    ts(inp)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ts(inp)
    torch.cuda.synchronize()
tab = prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=80, max_name_column_width=40,
                                                         max_shapes_column_width=70)
print(tab)
