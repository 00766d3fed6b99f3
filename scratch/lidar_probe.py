"""Sub-step timing of the LiDAR-query branch (sync + wall clock; 10-sweep frame, sequential branches)."""
import os, sys, time, torch, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
model.test_cfg['concurrent_query_branches'] = False
frame, inp = bench.make_inputs(10, 0, dev)
acc = collections.OrderedDict()
def wrapf(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize()
        acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t0) * 1e3; return r
    if isinstance(obj, torch.nn.Module) and isinstance(f.__self__ if hasattr(f, '__self__') else None, torch.nn.Module) and name != 'forward' and isinstance(getattr(type(obj), name, None), property): return
    object.__setattr__(obj, name, g)
with torch.no_grad():
    for _ in range(2): bench.step(model, inp)
    wrapf(model, 'fsd_forward', 'fsd_forward (whole branch)')
    wrapf(model, 'pre_voxelize', 'pre_voxelize')
    wrapf(model, 'grouped_sample_and_cluster', 'grouped_sample_and_cluster')
    wrapf(model, 'extract_feat', 'extract_feat (SIR backbone)')
    wrapf(model.bbox_head, 'forward', 'bbox_head')
    for i, blk in enumerate(model.backbone.block_list if hasattr(model.backbone, 'block_list') else []):
        wrapf(blk, 'forward_parts', f'  sir block {i}')
    n = 5
    for _ in range(n): bench.step(model, inp)
for k, v in acc.items(): print(f'{k:36s} {v / n:8.3f} ms')
