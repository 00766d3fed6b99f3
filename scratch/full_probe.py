"""Timing of the full FSF.simple_test (stages 1-3 + heads + refine + NMS) vs the hot path, 10-sweep frame."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(int(sys.argv[1]) if len(sys.argv) > 1 else 10, 0, dev)
def run(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
with torch.no_grad():
    ms_hot, out = run(lambda: model.forward_hot_path(inp['points'], inp['img_metas'], inp['mask_data'], inp['mask_anno']))
    print('hot path ms', round(ms_hot, 2), 'queries', out['frustum_obj_feats'].shape[0], out['fsd_obj_feats'].shape[0])
    ms_q, bl = run(lambda: model.forward_queries(inp['points'], inp['img_metas'], inp['mask_data'], inp['mask_anno']))
    print('forward_queries ms', round(ms_q, 2), 'boxes', len(bl[0][0]))
    ms_full, res = run(lambda: model.simple_test(inp['points'], inp['img_metas'], inp['mask_data'], inp['mask_anno']))
    print('simple_test ms', round(ms_full, 2))
    # stage split of the tail
    import torch.autograd.profiler as prof
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as p:
        model.forward_queries(inp['points'], inp['img_metas'], inp['mask_data'], inp['mask_anno'])
        torch.cuda.synchronize()
    print(p.key_averages().table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=60))
