import os, sys, time, torch, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev)
frame, inp = bench.make_inputs(10, 0, dev)
for _ in range(3): bench.step(model, inp)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    for rep in range(3):
        model._gather_cache = None
        points, img_metas, mask_data, mask_anno = inp['points'], inp['img_metas'], inp['mask_data'], inp['mask_anno']
        pts, infos = model.split_points_last_3dim(points)
        seg_tuple = model.segmentor.simple_test(pts, img_metas, extract_feat_only=True, rescale=False)
        seg = model.segmentor_feat_inhance_test(seg_tuple, infos, mask_anno, mask_data, img_metas)
        f = model.frustum_forward(seg, mask_anno, mask_data, infos, img_metas, cluster_center=None)
        l = model.fsd_forward(seg, img_metas)
        comb = model.combine_frustum_and_fsd(f[1], f[2], f[3], f[0], f[4], l[1], l[2], l[3], l[0])
        t0 = T()
        pr = cProfile.Profile(); pr.enable()
        bl = model.multi_stage_refine_test(comb[0], comb[1], comb[2], seg['seg_points'], infos, seg['seg_feats'], seg['batch_idx'],
                                           mask_data, mask_anno, comb[4], img_metas, comb[3])
        torch.cuda.synchronize(); pr.disable()
        print('refine ms', (T() - t0) * 1e3)
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats('tottime').print_stats(12); print(buf.getvalue()[:3500])
