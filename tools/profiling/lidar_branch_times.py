"""Wall time of the pieces of the LiDAR-query branch (fsd_forward) on the 10-sweep frame, one at a time with syncs.  (GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(3): bench.step(model, inp)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
acc = {}
def add(k, t0):
    t1 = T(); acc[k] = acc.get(k, 0.0) + (t1 - t0) * 1e3; return t1
N = 5
with torch.no_grad():
    for _ in range(N):
        model._gather_cache = None; model._fg_cache = None
        points, img_metas, mask_data, mask_anno = inp['points'], inp['img_metas'], inp['mask_data'], inp['mask_anno']
        pts, infos = model.split_points_last_3dim(points)
        seg_tuple = model.segmentor.simple_test(pts, img_metas, extract_feat_only=True, rescale=False)
        seg = model.segmentor_feat_inhance_test(seg_tuple, infos, mask_anno, mask_data, img_metas)
        model._batch_size_hint = 1
        d = dict(seg_points=seg["seg_points"], seg_logits=seg["seg_logits"], seg_vote_preds=seg["seg_vote_preds"],
                 seg_feats=seg["seg_feats"], batch_idx=seg["batch_idx"], vote_offsets=seg["offsets"])
        t = T()
        d = model.pre_voxelize(d); t = add("pre_voxelize", t)
        points_s, logits, votes, feats, centers, inds = model.grouped_sample_and_cluster(d); t = add("grouped_sample_and_cluster", t)
        pts_feats = model._grouped_feats_concat; model._grouped_feats_concat = None
        ex = model.extract_feat(points_s, pts_feats, inds, img_metas, centers); t = add("extract_feat (SIR x3)", t)
        outs = model.bbox_head(ex["cluster_feats"]); t = add("bbox_head", t)
for k, v in acc.items(): print(f"{k:32s} {v / N:7.2f} ms")
# kernels inside grouped_sample_and_cluster
import collections
from torch.profiler import profile, ProfilerActivity
with torch.no_grad():
    d0 = dict(seg_points=seg["seg_points"], seg_logits=seg["seg_logits"], seg_vote_preds=seg["seg_vote_preds"],
              seg_feats=seg["seg_feats"], batch_idx=seg["batch_idx"], vote_offsets=seg["offsets"])
    d1 = model.pre_voxelize(d0)
    model.grouped_sample_and_cluster(d1); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        model.grouped_sample_and_cluster(d1); torch.cuda.synchronize()
ks = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
agg = collections.defaultdict(lambda: [0, 0.0])
for e in ks:
    agg[e.name][0] += 1; agg[e.name][1] += e.device_time
print(f"grouped_sample_and_cluster: {len(ks)} device ops, {sum(v[1] for v in agg.values()) / 1e3:.2f} ms of device time")
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f"{c:4d} {us:9.1f} us  {n[:130]}")
