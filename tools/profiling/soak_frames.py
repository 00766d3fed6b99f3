"""Different frames (sweeps 1..10, seeds): full forward must run, be deterministic and equal between branch modes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
cfg = model.test_cfg
with torch.no_grad():
    for sweeps in (1, 2, 3, 5, 7, 10):
        for seed in (1, 2):
            frame, inp = bench.make_inputs(sweeps, seed, dev)
            cfg['concurrent_query_branches'] = True
            a = bench.step(model, inp)[0]
            b = bench.step(model, inp)[0]
            cfg['concurrent_query_branches'] = False
            c = bench.step(model, inp)[0]
            ok = all(torch.equal(a[k] if k != 'boxes_3d' else a[k].tensor, x[k] if k != 'boxes_3d' else x[k].tensor) for x in (b, c) for k in ('boxes_3d', 'scores_3d', 'labels_3d'))
            print(f'sweeps {sweeps:2d} seed {seed}: points {inp["points"][0].shape[0]:7d} boxes {a["boxes_3d"].tensor.shape[0]:4d} finite {bool(torch.isfinite(a["boxes_3d"].tensor).all())} identical {ok}')
cfg.pop('concurrent_query_branches', None)
# memory stability over many frames (rotating inputs): allocator high-water mark must settle
pool = [bench.make_inputs(10, s, dev)[1] for s in range(4)]
torch.cuda.reset_peak_memory_stats()
marks = []
for i in range(120):
    bench.step(model, pool[i % 4])
    if i % 30 == 29:
        torch.cuda.synchronize(); marks.append((torch.cuda.memory_allocated() >> 20, torch.cuda.max_memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
print('MiB (allocated, peak, reserved) every 30 frames:', marks)
