"""Wall time per stage of FSF.simple_test (sync between stages), 10-sweep frame."""
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
N = 5
with torch.no_grad():
    for _ in range(N):
        model._gather_cache = None
        points, img_metas, mask_data, mask_anno = inp['points'], inp['img_metas'], inp['mask_data'], inp['mask_anno']
        t0 = T()
        pts, infos = model.split_points_last_3dim(points)
        seg_tuple = model.segmentor.simple_test(pts, img_metas, extract_feat_only=True, rescale=False)
        t1 = T()
        seg = model.segmentor_feat_inhance_test(seg_tuple, infos, mask_anno, mask_data, img_metas)
        t2 = T()
        f = model.frustum_forward(seg, mask_anno, mask_data, infos, img_metas, cluster_center=None)
        t3 = T()
        l = model.fsd_forward(seg, img_metas)
        t4 = T()
        comb = model.combine_frustum_and_fsd(f[1], f[2], f[3], f[0], f[4], l[1], l[2], l[3], l[0])
        t5 = T()
        bl = model.multi_stage_refine_test(comb[0], comb[1], comb[2], seg['seg_points'], infos, seg['seg_feats'], seg['batch_idx'],
                                           mask_data, mask_anno, comb[4], img_metas, comb[3])
        t6 = T()
        for k, v in zip(['segmentor(voxelize+VFE+UNet+neck)', 'image fusion + seg head', 'camera queries', 'lidar queries',
                         'combine', 'refine + boxes + NMS'], [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5]):
            acc[k] = acc.get(k, 0) + v
for k, v in acc.items(): print(f'{k:38s} {v / N * 1e3:7.2f} ms')
print('sum', sum(acc.values()) / N * 1e3)
