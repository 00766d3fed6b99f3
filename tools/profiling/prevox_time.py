"""pre_voxelize's means (scatter_mean_multi over the 0.1 m voxels of the 10-sweep frame) timed in place with HIP events.  (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from fullysparsefusion_amd.mmdet3d_plugin.ops import sst_ops
dev = torch.device('cuda:0')
model = bench.build_model(dev); frame, inp = bench.make_inputs(10, 0, dev)
cap = {}
orig = sst_ops.scatter_mean_multi
import fullysparsefusion_amd.mmdet3d_plugin.models.detectors.single_stage_fsd as ssf
def spy(feats, new_coors, unq_inv):
    cap['a'] = (feats, new_coors, unq_inv); return orig(feats, new_coors, unq_inv)
ssf.scatter_mean_multi = spy
for _ in range(2): bench.step(model, inp)
ssf.scatter_mean_multi = orig
feats, nc, inv = cap['a']
print([tuple(f.shape) + (f.stride(0),) for f in feats], nc.shape[0])
def t(f, it=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
print(f"scatter_mean_multi: {t(lambda: orig(feats, nc, inv)):.1f} us")
