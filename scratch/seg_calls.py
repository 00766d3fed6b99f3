"""Shapes of the segmented reductions of one frame (which ones take the narrow-row VEC = 1 path)."""
import os, sys, torch, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from fullysparsefusion_amd import hip_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
with torch.no_grad(): bench.step(model, inp)
orig = hip_ops.segment_reduce
cnt = collections.Counter()
def rec(feat, plan, mode, return_argmax=False):
    cnt[(tuple(feat.shape), feat.stride(0), plan.m, mode, feat.data_ptr() % 16)] += 1
    return orig(feat, plan, mode, return_argmax)
hip_ops.segment_reduce = rec
import fullysparsefusion_amd.mmdet3d_plugin.ops.sst_ops as so
with torch.no_grad(): bench.step(model, inp)
for k, v in sorted(cnt.items(), key=lambda kv: -kv[0][0][0] * kv[0][0][1]): print(v, k)
