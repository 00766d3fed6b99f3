"""Oracle: module-level CPU restatement of the hot path.  TEST INFRASTRUCTURE ONLY.

Every function takes a CPU copy of the product's nn.Module ONLY as a parameter container (Linear / norm weights
are data) and restates the forward wiring with the oracle primitives: torch.unique + scatter_reduce
(oracle.scatter), the spconv-v1 restatement (oracle.spconv), the numpy projection (oracle.project).

PARITY UNPINNED for DynamicScatterVFE / SIRLayer / SimpleSparseUNet: their sources are in the authors' mmdet3d fork,
not in the reference tree; restated from the published SST/FSD modules.  SIR's block wiring and the
neck are pinned by tests/golden/{sir_flow,neck}.npz; the FSF glue by tests/golden/{project,frustum_glue}.npz.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import project as oproj
from . import scatter as oscatter
from . import spconv as osp


# ------------------------------------------------------------------------------- dense layers, restated
def apply_module(m, x):
    """Evaluates a dense sub-module from its PARAMETERS ONLY (the product's `forward` is never called): the product's
    modules are weight containers here.  Structure restated from the reference's `build_mlp`
    (projects/mmdet3d_plugin/ops/sst_ops.py:808-833): a Sequential of [Linear(bias) -> norm -> act (-> Dropout)] blocks,
    the last entry a plain Linear(bias=True) when `is_head`; norm = LayerNorm / (naiveSync)BatchNorm1d in eval mode
    (batch statistics under `unet_forward(train=True)`), act = GELU (erf) / ReLU."""
    name = type(m).__mro__
    if isinstance(m, torch.nn.Sequential):
        for child in m._modules.values():
            x = apply_module(child, x)
        return x
    if isinstance(m, torch.nn.Linear):
        return F.linear(x, m.weight, m.bias)
    if isinstance(m, torch.nn.LayerNorm):
        return F.layer_norm(x, m.normalized_shape, m.weight, m.bias, m.eps)
    if isinstance(m, torch.nn.BatchNorm1d):
        return _bn_eval(m, x)
    if isinstance(m, torch.nn.GELU):
        return F.gelu(x)
    if isinstance(m, torch.nn.ReLU):
        return torch.relu(x)
    if isinstance(m, (torch.nn.Dropout, torch.nn.Identity)):
        return x
    raise TypeError(f"oracle.apply_module: no restatement for {name[0].__name__}")


# ------------------------------------------------------------------------------------- DynamicScatterVFE
def vfe_forward(vfe, features, coors):
    """Published SST `DynamicScatterVFE.forward(features, coors, return_inv=True)`; called at
    projects/mmdet3d_plugin/models/detectors/single_stage_fsd.py:232."""
    new_coors, unq_inv = torch.unique(coors, return_inverse=True, return_counts=False, dim=0)
    m = new_coors.size(0)
    ls = [features]
    if vfe._with_cluster_center:
        voxel_mean = oscatter.segment_mean(features, unq_inv, m)
        ls.append(features[:, :3] - voxel_mean[unq_inv][:, :3])
    if vfe._with_voxel_center:
        f_center = features.new_zeros((features.size(0), 3))
        f_center[:, 0] = features[:, 0] - (coors[:, 3].type_as(features) * vfe.vx + vfe.x_offset)
        f_center[:, 1] = features[:, 1] - (coors[:, 2].type_as(features) * vfe.vy + vfe.y_offset)
        f_center[:, 2] = features[:, 2] - (coors[:, 1].type_as(features) * vfe.vz + vfe.z_offset)
        ls.append(f_center)
    x = torch.cat(ls, dim=-1)
    for i, layer in enumerate(vfe.vfe_layers):
        point_feats = apply_module(layer.act, apply_module(layer.norm, apply_module(layer.linear, x)))
        voxel_feats, _ = oscatter.segment_max(point_feats, unq_inv, m)
        if i != len(vfe.vfe_layers) - 1:
            x = torch.cat([point_feats, voxel_feats[unq_inv]], dim=1)
    return voxel_feats, new_coors, unq_inv


# ---------------------------------------------------------------------------------------------- SIRLayer
def sir_layer_forward(layer, features, coors, f_cluster, unq_inv, new_coors):
    """Published FSD `SIRLayer.forward` (built by projects/mmdet3d_plugin/models/backbones/sir.py:41-61)."""
    xyz_norm = torch.tensor(layer.xyz_normalizer, dtype=features.dtype)
    x = torch.cat([features[:, :3] / xyz_norm[None, :], features[:, 3:]], dim=1)
    x = x * apply_module(layer.rel_mlp, f_cluster / layer.rel_dist_scaler)
    m = new_coors.size(0)
    outs = []
    for i, vfe in enumerate(layer.vfe_layers):
        point_feats = apply_module(vfe.act, apply_module(vfe.norm, apply_module(vfe.linear, x)))
        grp, _ = oscatter.segment_max(point_feats, unq_inv, m)
        outs.append(grp)
        if i != len(layer.vfe_layers) - 1:
            x = torch.cat([point_feats, grp[unq_inv]], dim=1)
    return point_feats, torch.cat(outs, dim=1)


def sir_forward(sir, points, features, coors, f_cluster):
    """projects/mmdet3d_plugin/models/backbones/sir.py:65-85."""
    new_coors, unq_inv = torch.unique(coors, return_inverse=True, return_counts=False, dim=0)
    out_feats = features
    cluster_feats = []
    for block in sir.block_list:
        in_feats = torch.cat([points, out_feats], 1)
        out_feats, grp = sir_layer_forward(block, in_feats, coors, f_cluster, unq_inv, new_coors)
        cluster_feats.append(grp)
    return out_feats, torch.cat(cluster_feats, dim=1), new_coors


# --------------------------------------------------------------------------------------- SimpleSparseUNet
class _SpT:
    def __init__(self, features, indices, shape, batch_size, rulebooks):
        self.features, self.indices, self.shape, self.batch_size, self.rb = features, indices, list(shape), batch_size, rulebooks


_TRAIN = [False]  # unet_forward(train=True): batch statistics in the norms, autograd through the conv weights
_GRAD = [False]   # keep the conv weights attached to the autograd graph (eval-mode norms): gradient parity tests


def _bn_eval(bn, x):
    if _TRAIN[0]:
        return F.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.0, bn.eps)
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)


def _conv(conv, x):
    """spconv v1 SubMConv3d / SparseConv3d / SparseInverseConv3d with indice_key caching."""
    key = conv.indice_key
    w = conv.weight if (_TRAIN[0] or _GRAD[0]) else conv.weight.detach()
    if conv.inverse:
        out_idx, pairs, in_idx, in_shape = x.rb[key]
        feat = osp.indice_conv(x.features, w, pairs, in_idx.shape[0], inverse=True)
        return _SpT(feat, in_idx, in_shape, x.batch_size, x.rb)
    if key not in x.rb:
        if conv.subm:
            pad = [(k - 1) // 2 * d for k, d in zip(conv.kernel_size, conv.dilation)]
            out_idx, pairs, oshape = osp.build_rulebook(x.indices, x.batch_size, x.shape, conv.kernel_size, (1, 1, 1), pad,
                                                        conv.dilation, True)
        else:
            out_idx, pairs, oshape = osp.build_rulebook(x.indices, x.batch_size, x.shape, conv.kernel_size, conv.stride,
                                                        conv.padding, conv.dilation, False)
        x.rb[key] = (out_idx, pairs, x.indices, x.shape)
        x.rb[key + "/oshape"] = oshape
    out_idx, pairs, _, _ = x.rb[key]
    feat = osp.indice_conv(x.features, w, pairs, out_idx.shape[0])
    return _SpT(feat, out_idx, x.rb[key + "/oshape"], x.batch_size, x.rb)


def _convmodule(seq, x):
    mods = list(seq._modules.values())
    x = _conv(mods[0], x)
    x.features = torch.relu(_bn_eval(mods[1], x.features))
    return x


def _basic_block(blk, x):
    identity = x.features
    out = _conv(blk.conv1, x)
    out.features = torch.relu(_bn_eval(blk.norm1, out.features))
    out = _conv(blk.conv2, out)
    out.features = torch.relu(_bn_eval(blk.norm2, out.features) + identity)
    return out


def unet_forward(unet, voxel_feats, voxel_coors, batch_size, train=False, grad=False):
    """Published SST `SimpleSparseUNet.forward` (called at single_stage_fsd.py:234); eval mode unless `train`; `grad`
    keeps the conv weights in the autograd graph."""
    _TRAIN[0], _GRAD[0] = bool(train), bool(grad)
    try:
        return _unet_forward(unet, voxel_feats, voxel_coors, batch_size)
    finally:
        _TRAIN[0] = _GRAD[0] = False


def _unet_forward(unet, voxel_feats, voxel_coors, batch_size):
    x = _SpT(voxel_feats, voxel_coors.int().numpy(), unet.sparse_shape, batch_size, {})
    x = _convmodule(unet.conv_input, x)
    enc = []
    for stage in unet.encoder_layers._modules.values():
        for block in stage._modules.values():
            x = _convmodule(block, x)
        enc.append(x)
    x = enc[-1]
    for i in range(unet.stage_num, 0, -1):
        lat = _basic_block(getattr(unet, f"lateral_layer{i}"), enc[i - 1])
        cat = _SpT(torch.cat((x.features, lat.features), dim=1), lat.indices, lat.shape, lat.batch_size, lat.rb)
        merged = _convmodule(getattr(unet, f"merge_layer{i}"), cat)
        n, c_out = merged.features.shape
        reduced = cat.features.view(n, c_out, -1).sum(dim=2)
        cat.features = merged.features + reduced
        x = _convmodule(getattr(unet, f"upsample_layer{i}"), cat)
    return x.features


# --------------------------------------------------------------------------------------------------- neck
def neck_forward(neck, points, pts_coors, voxel_feats, inv, padding=-1):
    """projects/mmdet3d_plugin/models/necks/voxel2point_neck.py:27-70 (with_xyz, not normalised)."""
    pts_feats = voxel_feats[inv]
    mask = ~((pts_feats == padding).all(1))
    pts_feats, pts_coors, points = pts_feats[mask], pts_coors[mask], points[mask]
    vs = torch.tensor(neck.voxel_size, dtype=torch.float32).reshape(1, 3)
    mn = torch.tensor(neck.point_cloud_range[:3], dtype=torch.float32).reshape(1, 3)
    centers = ((pts_coors[:, [3, 2, 1]].float() + 0.5) * vs + mn).to(points.dtype)
    return torch.cat([pts_feats, points[:, :3] - centers], 1), mask


# ----------------------------------------------------------------------------------------- VoteSegmentor
def segmentor_extract_feat(seg, points_list, grad=False, dtype=torch.float32):
    """VoteSegmentor.extract_feat (single_stage_fsd.py:228-245)."""
    from . import voxelize as ovox

    pts, coors = ovox.voxelize_batch([p.float().numpy() for p in points_list], seg.voxel_size, seg.point_cloud_range)
    pts, coors = torch.from_numpy(pts).to(dtype), torch.from_numpy(coors)
    voxel_feats, voxel_coors, inv = vfe_forward(seg.voxel_encoder, pts, coors)
    unet_out = unet_forward(seg.backbone, voxel_feats, voxel_coors, len(points_list), grad=grad)
    out, mask = neck_forward(seg.decode_neck, pts, coors, unet_out, inv)
    return dict(neck=out, mask=mask, coors=coors, points=pts, voxel_feats=voxel_feats, voxel_coors=voxel_coors, inv=inv,
                unet=unet_out)


# --------------------------------------------------------------------------------------------------- FSF
def point_image_feat(fsf, obj_id, mask_anno, encode_mlp, img_hw, dtype=torch.float32):
    """FSF.img_cross_attn (FSF.py:694-728), one sample, on the gathered ids `obj_id` [n, cams, classes]: the camera with the
    largest id sum (:716-718, first maximum), that camera's id row, `get_all_cls_preds_2d` (:506-535: anno row id - 1, id 0
    -> zeros with category = the number of id planes), `encode_2d_feats` (:537-552) and the MLP.  nuScenes keeps the score
    column only (:472-473, `cam_select_score`); Argoverse 2 (`is_argo`) encodes box / W H, score and the one-hot category of
    every id plane (:459-470)."""
    obj_id = np.asarray(obj_id)
    anno = torch.as_tensor(np.asarray(mask_anno), dtype=torch.float32)
    if not fsf.is_argo and not fsf.encode_label_only:
        _, score = oproj.cam_select_score(obj_id, anno.numpy())
        return apply_module(encode_mlp, torch.from_numpy(score).to(dtype))
    cam = obj_id.sum(-1).argmax(-1)
    ids = torch.from_numpy(obj_id[np.arange(obj_id.shape[0]), cam, :])          # [n, planes]
    valid = ids > 0
    preds = anno[(ids - 1).clamp(min=0)] * valid.unsqueeze(-1)                  # [n, planes, 9]
    preds[..., 5] = torch.where(valid, preds[..., 5], torch.full((), float(ids.shape[-1])))
    preds = preds.reshape(-1, preds.shape[-1])
    onehot = F.one_hot(preds[:, 5].long(), fsf.num_classes + 1).float()
    if fsf.encode_label_only:
        enc = onehot
    else:
        bbox = preds[:, :4].clone()
        bbox[:, 0::2] /= img_hw[1]
        bbox[:, 1::2] /= img_hw[0]
        enc = torch.cat([bbox, preds[:, 4:5], onehot], -1)
    return apply_module(encode_mlp, enc.to(dtype))


def fsf_stage1(fsf, points8, mask_data, mask_anno, lidar2img, grad=False, dtype=torch.float32):
    """FSF.simple_test step 1 (FSF.py:1123-1130): segmentor features + image branch + seg head, one sample."""
    points = [points8[:, :-3]]
    infos = points8[:, -3:]
    ex = segmentor_extract_feat(fsf.segmentor, points, grad=grad, dtype=dtype)
    assert bool(ex["mask"].all())
    obj_id, _ = oproj.points_in_mask(infos.numpy(), mask_data.numpy(), lidar2img.numpy())
    img_feat = point_image_feat(fsf, obj_id, mask_anno, fsf.segmentor_updated_mlp, mask_data.shape[-2:], dtype)
    pts_feats = ex["neck"] + img_feat
    head = fsf.segmentor.segmentation_head
    h = apply_module(head.pre_seg_conv, pts_feats)
    seg_logits, vote_preds = apply_module(head.conv_seg, h), apply_module(head.voting, h)
    return dict(ex=ex, obj_id=torch.from_numpy(obj_id), seg_points=ex["points"], seg_logits=seg_logits,
                seg_vote_preds=vote_preds, offsets=vote_preds * vote_preds.abs(), seg_feats=pts_feats,
                batch_idx=ex["coors"][:, 0])


def double_overlap_pts(pts_feat, bz_coor, points, obj_id_tensor, w):
    """FSF.double_overlap_pts (FSF.py:260-297); pinned by tests/golden/frustum_glue.npz."""
    obj = obj_id_tensor.reshape(obj_id_tensor.shape[0], -1)
    overlaps = (obj > 0).sum(-1)
    raw = obj.max(-1)[0]
    feats, bzs, pts, ws, ids = [pts_feat], [bz_coor], [points], [w], [raw]
    for k in range(2, int(overlaps.max()) + 1):
        msk = overlaps == k
        if msk.sum() == 0:
            continue
        feats.append(pts_feat[msk].repeat(k - 1, 1))
        bzs.append(bz_coor[msk].repeat(k - 1, 1))
        pts.append(points[msk].repeat(k - 1, 1))
        ws.append(w[msk].repeat(k - 1))
        sv = obj[msk].topk(k, dim=-1)[0]
        for j in range(1, k):
            ids.append(sv[:, j])
    return torch.cat(feats), torch.cat(bzs), torch.cat(pts), torch.cat(ids), torch.cat(ws)


def fsf_stage2(fsf, s1, mask_anno, img_hw):
    """FSF.frustum_forward without the head (FSF.py:607-650)."""
    w = 1 - s1["seg_logits"].softmax(1)[:, -1]
    obj = s1["obj_id"]
    fg = obj.sum((-2, -1)) > 0
    a = (s1["seg_feats"][fg], s1["batch_idx"][fg].unsqueeze(-1), s1["seg_points"][fg], obj[fg], w[fg])
    feat, bz, pts, ids, ww = double_overlap_pts(*a)
    sir_coors = torch.cat([bz, torch.zeros_like(bz), ids.unsqueeze(-1)], dim=-1)
    pw = ww.unsqueeze(-1).clamp(min=1e-5)
    mean, mcoors, inv = oscatter.scatter_v2(torch.cat([pts[:, :3] * pw, pw], -1), sir_coors, "avg")
    center = mean[:, :3] / mean[:, 3:4]
    f_cluster = pts[:, :3] - center[inv]
    _, cluster_feats, out_coors = sir_forward(fsf.frustum_sir, pts, feat, sir_coors, f_cluster)
    ids_k = out_coors[:, 2]
    preds = torch.zeros((out_coors.size(0), 9), dtype=feat.dtype)  # (float64 when the chain runs in float64: test arbitration)
    valid = ids_k > 0
    preds[valid] = mask_anno[ids_k[valid] - 1].to(feat.dtype)
    preds[~valid, 5] = fsf.num_classes
    bbox = preds[:, :4].clone()
    bbox[:, 0::2] /= img_hw[1]
    bbox[:, 1::2] /= img_hw[0]
    enc = torch.cat([bbox, preds[:, 4:5], F.one_hot(preds[:, 5].long(), fsf.num_classes + 1).to(feat.dtype)], -1)
    img_feat = apply_module(fsf.encode_2d_mlp, enc)
    return dict(obj_feat=torch.cat([cluster_feats, img_feat], -1), obj_coors=out_coors, obj_centers=center,
                sir_coors=sir_coors, f_cluster=f_cluster, preds_2d=preds)


def connected_components_xy(points, dist):
    """find_connected_componets_single_batch (single_stage_fsd.py:69-82): the reference builds the dense n x n matrix of
    fp32 distances `((p_i - p_j) ** 2).sum(2) ** 0.5 < dist` and hands it to scipy.  Beyond a few thousand centres that
    matrix does not fit (8.4e4 centres on the 10-sweep frame), so the same adjacency is built sparsely: candidate pairs
    from a k-d tree with a 1 % larger radius, then the reference's own fp32 expression decides each candidate.  scipy labels
    components in order of their smallest node either way, so the labels are those of the dense call (checked against it in
    tests/test_oracle_golden.py)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    p = points[:, :2]
    n = p.shape[0]
    if n <= 4096:
        d = p[:, None, :] - p[None, :, :]
        d = (d ** 2).sum(2) ** 0.5
        return torch.from_numpy(connected_components((d < dist).numpy(), directed=False)[1]).int()
    from scipy.spatial import cKDTree

    pn = p.numpy().astype(np.float64)
    pairs = cKDTree(pn).query_pairs(float(dist) * 1.01 + 1e-6, output_type="ndarray")
    a, b = torch.from_numpy(pairs[:, 0]), torch.from_numpy(pairs[:, 1])
    d = ((p[a] - p[b]) ** 2).sum(1) ** 0.5  # fp32, the reference's expression on the candidate pairs
    keep = (d < dist).numpy()
    i, j = pairs[keep, 0], pairs[keep, 1]
    adj = coo_matrix((np.ones(2 * i.size, dtype=bool), (np.concatenate([i, j]), np.concatenate([j, i]))), shape=(n, n)).tocsr()
    return torch.from_numpy(connected_components(adj, directed=False)[1]).int()


def fsf_stage3(fsf, s1, record=None, replay=None):
    """FSF.fsd_forward without the head (FSF.py:569-600): pre_voxelize, group_sample, ClusterAssigner, SIR.

    `record` (a dict) receives every INTEGER / boolean decision of the stage — the pre-voxelization cells, per class group the
    foreground mask, the arg-max tie weights of the vote, the density-filter mask, the cluster-voxel map and the component labels.
    `replay` (such a dict) takes them from there instead of deriving them: the float64 arbitration chain of
    tests/test_e2e_agreement_gpu.py runs the stage's ARITHMETIC in float64 on the integer structure the fp32 chain decided (a float64
    threshold test or floor would move a handful of borderline points between cells / clusters and the two chains would no longer
    describe the same groups)."""
    from . import voxelize as ovox

    cfg = fsf.cfg
    d = dict(seg_points=s1["seg_points"], seg_logits=s1["seg_logits"], seg_vote_preds=s1["seg_vote_preds"],
             seg_feats=s1["seg_feats"], batch_idx=s1["batch_idx"], vote_offsets=s1["offsets"])
    rng = fsf.cluster_assigner.point_cloud_range
    if replay is not None:
        new_coors, inv = replay["pre_new_coors"], replay["pre_inv"]
    else:
        coors = torch.from_numpy(ovox.divfloor_coors(d["seg_points"][:, :3].numpy(), cfg["pre_voxelization_size"], rng[:3],
                                                     "zyx", d["batch_idx"].numpy()))
        new_coors, inv = torch.unique(coors, return_inverse=True, dim=0)
    if record is not None:
        record.update(pre_new_coors=new_coors, pre_inv=inv, groups=[])
    vox = {k: oscatter.segment_mean(v, inv, new_coors.size(0)) for k, v in d.items() if v.is_floating_point()}
    vox["batch_idx"] = new_coors[:, 0]
    seg_logits = vox["seg_logits"]
    scores = seg_logits.softmax(1)
    offset = vox["vote_offsets"].reshape(-1, fsf.num_classes + 1, 3)
    names = cfg["class_names"]
    pts_all, feats_all, inds_all, centers_all = [], [], [], []
    for gi, group in enumerate(cfg["group_names"]):
        idx = [names.index(n) for n in group]
        rp = replay["groups"][gi] if replay is not None else None
        if rp is not None:
            fg, wgt = rp["fg"], rp["wgt"].to(seg_logits.dtype)
        else:
            fg = scores[:, idx].sum(1) > cfg["score_thresh"][gi]
            if fg.sum() == 0:
                fg[0] = True
            lg = seg_logits[:, idx][fg]
            wgt = ((lg - lg.max(1)[0][:, None]).abs() < 1e-6).float()
            wgt = wgt / wgt.sum(1)[:, None]
        centers = vox["seg_points"][fg, :3] + (offset[:, idx, :][fg] * wgt[:, :, None]).sum(1)
        vs = fsf.cluster_assigner.cluster_voxel_size[gi]
        bidx = vox["batch_idx"][fg].int()
        if rp is not None:
            valid, vinv, comp = rp["valid"], rp["vinv"], rp["comp"]
            cpts = centers[valid]
        else:
            cc = torch.from_numpy(ovox.divfloor_coors(centers.numpy(), vs, rng[:3], "xyz", bidx.numpy())).int()
            _, cinv, ccnt = torch.unique(cc, return_inverse=True, return_counts=True, dim=0)
            valid = ccnt[cinv] >= fsf.cluster_assigner.min_points
            if not valid.any():
                valid = ~valid
            cpts, cco = centers[valid], cc[valid]
            vc, vcoors, vinv = oscatter.scatter_v2(cpts, cco, "avg")
            comp = connected_components_xy(vc, fsf.cluster_assigner.connected_dist[gi])
        if record is not None:
            record["groups"].append(dict(fg=fg, wgt=wgt, valid=valid, vinv=vinv, comp=comp))
        per_pt = comp[vinv]
        inds_all.append(torch.stack([torch.full_like(per_pt, gi), bidx[valid], per_pt], 1))
        pts_all.append(vox["seg_points"][fg][valid])
        feats_all.append(torch.cat([vox["seg_logits"][fg][valid], vox["seg_vote_preds"][fg][valid], vox["seg_feats"][fg][valid]], 1))
        centers_all.append(cpts)
    points, feats = torch.cat(pts_all), torch.cat(feats_all)
    cluster_inds, center_preds = torch.cat(inds_all), torch.cat(centers_all)
    cxyz, _, cinv2 = oscatter.scatter_v2(center_preds, cluster_inds, "avg")
    f_cluster = points[:, :3] - cxyz[cinv2]
    _, cluster_feats, out_coors = sir_forward(fsf.backbone, points, feats, cluster_inds, f_cluster)
    return dict(cluster_feats=cluster_feats, cluster_xyz=cxyz, cluster_inds=out_coors, pts_cluster_inds=cluster_inds,
                points=points, pre_voxel_coors=new_coors)


# --------------------------------------------------------------------------------------- query refinement
def refine_head_forward(head, pts_xyz, pts_features, pts_info, roi_inds, rois):
    """FullySparseBboxHead.forward (projects/mmdet3d_plugin/models/roi_heads/bbox_heads/fsd_bbox_head.py:96-151) with its
    DynamicClusterVFE blocks restated as SIR layers (PARITY UNPINNED: un-vendored, see module docstring)."""
    rois = rois[:, 1:]
    rel_xyz = pts_xyz[:, :3] - rois[:, :3][roi_inds]
    new_coors, unq_inv = torch.unique(roi_inds, return_inverse=True)
    f_cluster = torch.cat([pts_info["local_xyz"], pts_info["boundary_offset"], pts_info["is_in_margin"][:, None], rel_xyz], -1)
    out_feats, cluster = pts_features, []
    for i, block in enumerate(head.block_list):
        in_feats = torch.cat([pts_xyz, out_feats], 1)
        if head.geo_input:
            in_feats = torch.cat([in_feats, f_cluster / 10], 1)
        out_feats, grp = sir_layer_forward(block, in_feats, roi_inds, f_cluster, unq_inv, new_coors)
        if i == head.num_blocks - 1 or head.use_middle_cluster_feature:
            cluster.append(grp)
    feats = torch.cat(cluster, 1)
    out = feats.new_zeros((rois.size(0), feats.size(1)))
    ok = new_coors >= 0
    out[new_coors[ok]] = feats[ok]
    mask = torch.zeros(rois.size(0), dtype=torch.bool)
    mask[new_coors[ok]] = True
    return out, mask


def multiclass_nms(boxes, scores, score_thr, nms_thr, max_num, rotated=True):
    """mmdet3d box3d_multiclass_nms over (x, y, z, w, l, h, yaw, ...) boxes and [n, C] scores (no background column),
    with the float64 IoU oracle; returns (row index into boxes, score, class) in the published output order."""
    from . import refine as orefine

    bev = boxes[:, [0, 1, 3, 4, 6]].double().numpy()
    xyxyr = np.stack([bev[:, 0] - bev[:, 2] / 2, bev[:, 1] - bev[:, 3] / 2, bev[:, 0] + bev[:, 2] / 2,
                      bev[:, 1] + bev[:, 3] / 2, bev[:, 4]], 1)
    rows, scs, labs = [], [], []
    for c in range(scores.shape[1]):
        sel = torch.nonzero(scores[:, c] > score_thr).squeeze(1)
        if sel.numel() == 0:
            continue
        order = torch.argsort(scores[sel, c], descending=True, stable=True)
        cand = sel[order]
        keep = orefine.nms_from_iou(orefine.iou_bev_matrix(xyxyr[cand.numpy()], rotated), nms_thr)
        rows.append(cand[keep])
        scs.append(scores[cand[keep], c])
        labs.append(torch.full((len(keep),), c, dtype=torch.long))
    if not rows:
        return torch.zeros(0, dtype=torch.long), torch.zeros(0), torch.zeros(0, dtype=torch.long)
    rows, scs, labs = torch.cat(rows), torch.cat(scs), torch.cat(labs)
    if rows.numel() > max_num:
        top = torch.argsort(scs, descending=True, stable=True)[:max_num]
        rows, scs, labs = rows[top], scs[top], labs[top]
    return rows, scs, labs


# ------------------------------------------------------------------ heads -> query combination -> refine -> boxes
def cluster_head_forward(head, feats):
    """SparseClusterHeadV2.forward (projects/mmdet3d_plugin/models/dense_heads/sparse_cluster_head_v2.py:134-167): shared
    MLP, then per task the FSDSeparateHead branches (`:18-60`); regression = cat(center, dim, rot[, vel])."""
    if head.shared_mlp is not None:
        feats = apply_module(head.shared_mlp, feats)
    cls, reg = [], []
    for h in head.task_heads:
        ret = {name: apply_module(getattr(h, name), feats) for name in h.attrs}
        parts = [ret["center"], ret["dim"], ret["rot"]] + ([ret["vel"]] if "vel" in ret else [])
        reg.append(torch.cat(parts, dim=-1))
        cls.append(ret["score"])
    return dict(cls_logits=cls, reg_preds=reg)


def coder_decode(reg_preds, base_points, eps=1e-6):
    """BasePointBBoxCoder.decode (core/bbox/coders/base_point_bbox_coder.py:59-82), code_size 10 (nuScenes: velocity rides
    along) or 8 (Argoverse 2); pinned bit-exact by tests/golden/refine_glue.npz."""
    velo = reg_preds[:, 8:]
    r = reg_preds[:, :8]
    dims = r[:, 3:6].exp() - eps
    xyz = r[:, :3] + base_points
    yaw = torch.atan2(r[:, 6:7], r[:, 7:8])
    return torch.cat([xyz, dims, yaw, velo], dim=1)


def combine_frustum_and_fsd(fsf, f_centers, f_coors, f_result, f_feats, f_preds_2d, l_centers, l_coors, l_result, l_feats):
    """FSF.combine_frustum_and_fsd (FSF.py:657-692)."""
    obj_centers = torch.cat([f_centers, l_centers], 0)
    l_re = l_coors.clone()
    l_re[:, 0] = l_coors[:, 1]
    l_re[:, 1] = l_coors[:, 0]
    l_re[:, 2] += fsf.fsd_begin_idx
    obj_coors = torch.cat([f_coors, l_re], 0)
    obj_result = {k: [torch.cat([f_result[k][t], l_result[k][t]], 0) for t in range(len(f_result[k]))] for k in f_result}
    obj_feats = torch.cat([apply_module(fsf.combine_frustum_feat_mlp, f_feats), apply_module(fsf.combine_fsd_feat_mlp, l_feats)], 0)
    preds_2d = torch.cat([f_preds_2d, f_preds_2d.new_zeros((l_feats.shape[0], f_preds_2d.shape[1]))], 0)
    return obj_centers, obj_coors, obj_result, obj_feats, preds_2d


def decode_stage_bboxes(obj_centers, bz_coors, reg_preds):
    """FSF.decode_stage_bboxes (FSF.py:1085-1094) at batch size 1 with one task."""
    assert len(reg_preds) == 1
    return torch.cat([bz_coors.unsqueeze(-1).to(obj_centers.dtype), coder_decode(reg_preds[0], obj_centers)], -1)


def query_feat_refine(fsf, i_stage, seg_points, seg_feats, obj_id, mask_anno, rois, pool, img_hw=(900, 1600)):
    """FSF.query_feat_refine (FSF.py:1000-1044) on a given pooling result `pool` = (point idx, roi idx, feats [k,13]):
    per-point image feature of the pooled points (img_cross_attn :694-728 with ext_pts_inds), then FullySparseBboxHead."""
    inds, roi_inds, info = pool
    img = point_image_feat(fsf, obj_id.numpy()[inds.numpy()], mask_anno, fsf.refine_img_mlp[i_stage], img_hw)
    feats = torch.cat([seg_feats[inds], img], -1)
    pts_info = dict(local_xyz=info[:, 3:6], boundary_offset=info[:, 6:-1], is_in_margin=info[:, -1])
    return refine_head_forward(fsf.refine_sir_layers[i_stage], seg_points[inds], feats, pts_info, roi_inds, rois)[0]


def refined_query(fsf, i_stage, lidar_img_feat, res_query_feat, obj_centers):
    """FSF.each_stage_refine (FSF.py:1076-1083): query update + refined head."""
    cur = apply_module(fsf.lidar_img_mlp[i_stage], lidar_img_feat)
    pos = apply_module(fsf.position_encoder[i_stage], obj_centers)
    query = apply_module(fsf.out_proj[i_stage], cur + res_query_feat + pos)
    return cluster_head_forward(fsf.frustum_refined_head[i_stage], query), query


def get_bboxes_single(cfg, cls_logits, reg_preds, cluster_xyz, delta=1e-4, near_tol=1e-3):
    """FrustumClusterHead._get_bboxes_single (frustum_cluster_head.py:587-698), one sample / one task, nms_pre = -1:
    sigmoid scores, decode, per-class rotated BEV NMS (mmdet3d box3d_multiclass_nms), top max_num.  Returns
    (row into the queries, score, label, decoded boxes, margin).  `margin` = the smallest |IoU - thr| over the NMS decisions
    that can reach the output: a class scan settles its boxes in descending score order and a box's fate depends only on
    boxes scored above it, so when the `max_num` cut keeps scores >= s*, only decisions between boxes scored >= s* matter
    (with ~1e4 boxes per class the scan takes ~1e6 decisions per class, of which some always sit within 1e-5 of the
    threshold — far behind the cut).  Reported exactly below `near_tol`, as `near_tol` otherwise."""
    from . import refine as orefine

    scores = cls_logits.sigmoid()
    boxes = coder_decode(reg_preds, cluster_xyz)
    bev = boxes[:, [0, 1, 3, 4, 6]].double().numpy()
    xyxyr = np.stack([bev[:, 0] - bev[:, 2] / 2, bev[:, 1] - bev[:, 3] / 2, bev[:, 0] + bev[:, 2] / 2,
                      bev[:, 1] + bev[:, 3] / 2, bev[:, 4]], 1)
    rows, scs, labs, close = [], [], [], []
    for c in range(scores.shape[1]):
        sel = torch.nonzero(scores[:, c] > cfg["score_thr"]).squeeze(1)
        if sel.numel() == 0:
            continue
        cand = sel[torch.argsort(scores[sel, c], descending=True, stable=True)]
        keep, _, near = orefine.nms_lazy(xyxyr[cand.numpy()], cfg["nms_thr"], rotated=cfg.get("use_rotate_nms", True),
                                         near_tol=near_tol)
        if near.shape[0]:  # (score of the LATER box of the decision, distance from the threshold)
            close.append(np.stack([scores[cand[near[:, 1].astype(np.int64)], c].double().numpy(), near[:, 2]], 1))
        rows.append(cand[keep])
        scs.append(scores[cand[keep], c])
        labs.append(torch.full((len(keep),), c, dtype=torch.long))
    if not rows:
        return torch.zeros(0, dtype=torch.long), torch.zeros(0), torch.zeros(0, dtype=torch.long), boxes, np.inf
    rows, scs, labs = torch.cat(rows), torch.cat(scs), torch.cat(labs)
    cut = -np.inf
    if rows.numel() > cfg["max_num"]:
        top = torch.argsort(scs, descending=True, stable=True)[:cfg["max_num"]]
        rows, scs, labs = rows[top], scs[top], labs[top]
        cut = float(scs.min())
    margin = near_tol
    if close:
        close = np.concatenate(close)
        reach = close[close[:, 0] >= cut]
        if reach.shape[0]:
            margin = float(reach[:, 1].min())
    return rows, scs, labs, boxes, margin


# ------------------------------------------------------------------------------ the whole frame, un-restarted
def simple_test(fsf, points8, mask_data, mask_anno, lidar2img, i_stage=0):
    """FSF.simple_test (FSF.py:1114-1178) for one sample, every stage fed by the ORACLE's own previous stage (no restart
    from device intermediates): segmentor + image fusion + segmentation head (:1123-1130), camera queries (:607-650) and
    LiDAR queries (:569-600) with their heads, combine_frustum_and_fsd (:657-692), one refinement stage (:1046-1083:
    decode_stage_bboxes, RoI point pooling, refine SIR, query update, refined head) and get_bboxes
    (frustum_cluster_head.py:587-698).  Returns the final (boxes, scores, labels) and the intermediates an end-to-end
    agreement test reports on (tests/test_e2e_agreement_gpu.py)."""
    from . import refine as orefine

    img_hw = tuple(mask_data.shape[-2:])
    s1 = fsf_stage1(fsf, points8, mask_data, mask_anno, lidar2img)
    s2 = fsf_stage2(fsf, s1, mask_anno, img_hw)
    s3_record = {}
    s3 = fsf_stage3(fsf, s1, record=s3_record)
    f_res = cluster_head_forward(fsf.frustum_obj_head, s2["obj_feat"])
    l_res = cluster_head_forward(fsf.bbox_head, s3["cluster_feats"])
    centers, coors, result, feats, p2d = combine_frustum_and_fsd(
        fsf, s2["obj_centers"], s2["obj_coors"], f_res, s2["obj_feat"], s2["preds_2d"],
        s3["cluster_xyz"], s3["cluster_inds"], l_res, s3["cluster_feats"])
    rois = decode_stage_bboxes(centers, coors[:, 0], result["reg_preds"])
    ext = fsf.roi_extractor
    wp, wr, wf = orefine.dynamic_point_pool(rois[:, 1:8].numpy(), s1["seg_points"][:, :3].numpy(), ext.extra_wlh,
                                            ext.max_inbox_point, ext.max_all_pts, stop_at_cap=True)
    pool = (torch.from_numpy(wp), torch.from_numpy(wr), torch.from_numpy(wf))
    lidar_img = query_feat_refine(fsf, i_stage, s1["seg_points"], s1["seg_feats"], s1["obj_id"], mask_anno, rois, pool,
                                  img_hw)
    res, query = refined_query(fsf, i_stage, lidar_img, feats, rois[:, 1:4])
    cfg = fsf.frustum_refined_head[i_stage].test_cfg
    rows, scs, labs, boxes, margin = get_bboxes_single(cfg, res["cls_logits"][0], res["reg_preds"][0], rois[:, 1:4])
    return dict(boxes=boxes[rows], scores=scs, labels=labs, rows=rows, margin=margin, s1=s1, s2=s2, s3=s3, s3_decisions=s3_record,
                query_coors=coors, query_feats=feats, rois=rois, refined_query=query, cls_logits=res["cls_logits"][0],
                reg_preds=res["reg_preds"][0], all_boxes=boxes)
