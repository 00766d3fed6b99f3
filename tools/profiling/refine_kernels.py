"""Kernels of the refine stage (multi_stage_refine_test + get_bboxes) of one 10-sweep frame: time, count, and how much of the
stage's wall time the device is busy.  (GPU box)"""
import collections, os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(3): bench.step(model, inp)
with torch.no_grad():
    points, img_metas, mask_data, mask_anno = inp['points'], inp['img_metas'], inp['mask_data'], inp['mask_anno']
    pts, infos = model.split_points_last_3dim(points)
    seg_tuple = model.segmentor.simple_test(pts, img_metas, extract_feat_only=True, rescale=False)
    seg = model.segmentor_feat_inhance_test(seg_tuple, infos, mask_anno, mask_data, img_metas)
    f = model.frustum_forward(seg, mask_anno, mask_data, infos, img_metas, cluster_center=None)
    l = model.fsd_forward(seg, img_metas)
    comb = model.combine_frustum_and_fsd(f[1], f[2], f[3], f[0], f[4], l[1], l[2], l[3], l[0])
    def refine():
        return model.multi_stage_refine_test(comb[0], comb[1], comb[2], seg['seg_points'], infos, seg['seg_feats'], seg['batch_idx'],
                                             mask_data, mask_anno, comb[4], img_metas, comb[3])
    for _ in range(2): refine()
    torch.cuda.synchronize(); t0 = time.perf_counter(); refine(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        refine(); torch.cuda.synchronize()
ks = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
agg = collections.defaultdict(lambda: [0, 0.0])
for e in ks:
    agg[e.name][0] += 1; agg[e.name][1] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
tot = sum(v[1] for v in agg.values())
print(f"refine stage wall {wall:.2f} ms; {len(ks)} device ops, {tot / 1e3:.2f} ms of device time")
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{c:4d} {us:9.1f} us  {n[:120]}")
print("pooling launches in order:")
for e in sorted(ks, key=lambda e: e.time_range.start):
    if 'pool_' in e.name or 'PoolScan' in e.name:
        print(f"  {e.time_range.start - ks[0].time_range.start:9.1f} us  +{(e.device_time if hasattr(e, 'device_time') else e.cuda_time):7.1f} us  {e.name[:70]}")
