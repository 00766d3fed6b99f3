"""K30 (csrc/lidar_frontend.hip): the LiDAR-query branch's clustering front end as one native call against the Python sequence of the
same entry points it replaces (SingleStageFSD.grouped_sample_and_cluster + extract_feat's centroids: detectors/single_stage_fsd.py
:802-982, :458-474) — same kernels, same order, same arguments, so EVERYTHING must be bit-identical: the sampled rows, the vote
centres, (group, sample, cluster id), the stack's unique and its plan, the centroids, and the SIR group features computed from them."""
import pytest
import torch

import bench
from fullysparsefusion_amd.mmdet3d_plugin.ops import sst_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(device):
    return bench.build_model(device)


@pytest.mark.parametrize("sweeps,seed", [(1, 0), (10, 1)])
def test_native_front_end_equals_the_python_sequence(device, model, sweeps, seed):
    inp = bench.make_inputs(sweeps, seed, device)[1]
    with torch.no_grad():
        model._gather_cache = model._fg_cache = model._img_pre = None
        pts, infos = model.split_points_last_3dim(inp["points"])
        seg_tuple = model.segmentor.simple_test(pts, inp["img_metas"], extract_feat_only=True, rescale=False)
        seg = model.segmentor_feat_inhance_test(seg_tuple, infos, inp["mask_anno"], inp["mask_data"], inp["img_metas"])
        outs = {}
        for native in (True, False):
            model.native_cluster_frontend = native
            sst_ops.clear_unique_cache()
            cap = {}
            fwd = model.backbone.forward

            def spy(points, features, coors, f_cluster=None, _cap=cap, _fwd=fwd):
                _cap["sir_in"] = (points, features.materialize(), coors, f_cluster.materialize() if hasattr(f_cluster, "materialize") else f_cluster)
                new_coors, inv, cnt = sst_ops.unique_with_plan(coors)
                plan = sst_ops.plan_of(inv, new_coors.size(0))
                _cap["unique"] = (new_coors, inv, cnt, plan.order, plan.seg_offsets)
                return _fwd(points, features, coors, f_cluster)

            model.backbone.forward = spy
            try:
                feats, xyz, inds, _ = model.fsd_forward(seg, inp["img_metas"], run_head=False)
            finally:
                model.backbone.forward = fwd
                model.native_cluster_frontend = True
            outs[native] = (feats, xyz, inds) + cap["sir_in"] + cap["unique"]
    sst_ops.clear_unique_cache()
    assert outs[True][0].shape[0] > 10
    for a, b in zip(outs[True], outs[False]):
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)
