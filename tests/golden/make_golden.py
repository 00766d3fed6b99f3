#!/usr/bin/env python
"""Generates the committed golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN PYTHON.

Runs only in the build container (needs /root/reference).  Nothing of the reference's source is copied:
`projects/mmdet3d_plugin/ops/sst_ops.py` is imported in place with four stub modules standing in for its
un-installable dependencies, and pure-torch methods of `FSF` / `Voxel2PointScatterNeck` / `SIR` are lifted
from their files with `ast` and executed.  Only inputs + outputs are saved (.npz).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixtures are DATA (seeded inputs, expected outputs); the GPU box never sees /root/reference.
"""
import ast
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
PLUGIN = os.path.join(REF, "projects/mmdet3d_plugin")
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------------------------- stubs
def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls

    def build_norm_layer(cfg, c):
        t = cfg.get("type")
        if t == "LN":
            return "ln", nn.LayerNorm(c, eps=cfg.get("eps", 1e-5))
        return "bn", nn.BatchNorm1d(c, eps=cfg.get("eps", 1e-5), momentum=cfg.get("momentum", 0.1))

    # torch_scatter shim over Tensor.scatter_reduce (an independent implementation of the same semantics)
    def scatter(src, index, dim=0, reduce="sum"):
        m = int(index.max()) + 1 if index.numel() else 0
        out = torch.zeros((m,) + tuple(src.shape[1:]), dtype=src.dtype)
        idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
        return out.scatter_reduce(0, idx, src, reduce={"sum": "sum", "mean": "mean"}[reduce], include_self=False)

    def scatter_max(src, index, dim=0):
        m = int(index.max()) + 1 if index.numel() else 0
        out = torch.zeros((m,) + tuple(src.shape[1:]), dtype=src.dtype)
        idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
        out = out.scatter_reduce(0, idx, src, reduce="amax", include_self=False)
        return out, None

    def ingroup_forward(g, out):
        order = torch.argsort(g, stable=True)
        sg = g[order]
        n = g.numel()
        head = torch.ones(n, dtype=torch.bool)
        head[1:] = sg[1:] != sg[:-1]
        pos = torch.arange(n)
        start = torch.cummax(torch.where(head, pos, torch.zeros_like(pos)), 0)[0]
        out[order] = pos - start

    mod("mmdet3d")
    mod("mmdet3d.ops", spconv=object(), voxel=object())
    sys.modules["mmdet3d"].ops = sys.modules["mmdet3d.ops"]
    mod("mmcv")
    mod("mmcv.cnn", build_norm_layer=build_norm_layer, ConvModule=object)
    mod("torch_scatter", scatter=scatter, scatter_max=scatter_max)
    mod("ingroup_indices", forward=ingroup_forward)
    mod("mmdet")
    mod("mmdet.models", NECKS=_Reg(), BACKBONES=_Reg())


def import_sst_ops():
    spec = importlib.util.spec_from_file_location("ref_sst_ops", os.path.join(PLUGIN, "ops/sst_ops.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def lift_methods(path, class_name, names, extra_globals):
    """exec the named FunctionDefs of `class_name` from `path` in a fresh namespace (no source is stored)."""
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == class_name][0]
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names]
    for f in fns:
        f.decorator_list = []
    module = ast.Module(body=fns, type_ignores=[])
    ns = dict(extra_globals)
    exec(compile(module, path, "exec"), ns)
    return {n: ns[n] for n in names}


# -------------------------------------------------------------------------------------- input builders
def make_lidar2img(ncam=6, fx=1266.4, cx=816.2, cy=491.5, scale=1.0):
    mats = []
    for c in range(ncam):
        yaw = c * 2 * math.pi / ncam
        R = np.array([[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]])
        cam_from_l = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], dtype=np.float64) @ R.T
        T = np.eye(4)
        T[:3, :3] = cam_from_l
        T[:3, 3] = [0.05 * c, -0.3, 0.2]
        K = np.eye(4)
        K[0, 0] = fx * scale
        K[1, 1] = fx * scale
        K[0, 2] = cx * scale
        K[1, 2] = cy * scale
        mats.append((K @ T).astype(np.float32))
    return np.stack(mats)


def make_mask(rng, ncam, ncls, H, W, num_inst, dtype):
    mask = np.zeros((ncam, ncls, H, W), dtype=dtype)
    inst = 1
    anno = np.zeros((num_inst, 9), dtype=np.float32)
    while inst <= num_inst:
        cam = rng.integers(ncam)
        cls = rng.integers(ncls)
        w = int(rng.integers(max(2, W // 40), W // 4))
        h = int(rng.integers(max(2, H // 40), H // 3))
        x1 = int(rng.integers(0, W - w))
        y1 = int(rng.integers(0, H - h))
        mask[cam, cls, y1:y1 + h, x1:x1 + w] = inst
        anno[inst - 1] = [x1, y1, x1 + w, y1 + h, rng.uniform(0.1, 1.0), cls, cam, inst, 1]
        inst += 1
    return mask, anno


# ------------------------------------------------------------------------------------------- fixtures
def gen_scatter(sst, rng):
    cases = {}

    def run(name, feat, coors, mode, **kw):
        out = sst.scatter_v2(torch.from_numpy(feat), torch.from_numpy(coors), mode, **kw)
        cases[name + "__feat"] = feat
        cases[name + "__coors"] = coors
        cases[name + "__mode"] = np.array(mode)
        cases[name + "__min_points"] = np.array(kw.get("min_points", 0))
        cases[name + "__new_feat"] = out[0].numpy()
        cases[name + "__new_coors"] = out[1].numpy()
        if len(out) > 2:
            cases[name + "__inv"] = out[2].numpy()

    n = 1200
    for mode in ("avg", "sum", "max"):
        # k = 4 voxel-like keys with a -1 row block, duplicates guaranteed
        coors = np.stack([rng.integers(0, 2, n), rng.integers(-1, 6, n), rng.integers(0, 40, n), rng.integers(0, 40, n)], 1).astype(np.int64)
        coors[:37] = -1
        run(f"k4_{mode}", rng.standard_normal((n, 32)).astype(np.float32), coors, mode)
        # k = 3 cluster keys (cls, batch, cluster) with one giant and many singleton groups
        c3 = np.stack([rng.integers(0, 3, n), rng.integers(0, 2, n), rng.integers(0, 5000, n)], 1).astype(np.int64)
        c3[: n // 3] = [1, 0, 7]
        run(f"k3_{mode}", rng.standard_normal((n, 5)).astype(np.float32), c3, mode)
        # k = 1
        c1 = rng.integers(-3, 50, (n // 3, 1)).astype(np.int64)
        run(f"k1_{mode}", rng.standard_normal((n // 3, 131)).astype(np.float32), c1, mode)
    # min_points path (sst_ops.py:160-165)
    coors = np.stack([rng.integers(0, 2, n), rng.integers(0, 30, n), rng.integers(0, 30, n)], 1).astype(np.int64)
    run("k3_minpts_avg", rng.standard_normal((n, 4)).astype(np.float32), coors, "avg", min_points=2)
    # tiny inputs
    run("single_row", rng.standard_normal((1, 8)).astype(np.float32), np.array([[0, 3, 2, 1]], dtype=np.int64), "max")
    run("all_same", rng.standard_normal((130, 16)).astype(np.float32), np.tile(np.array([[1, 2, 3]], dtype=np.int64), (130, 1)), "avg")
    np.savez_compressed(os.path.join(OUT, "scatter_v2.npz"), **cases)


def gen_project(fsf, rng):
    self_ns = types.SimpleNamespace(num_classes=10, num_cams=6, encode_label_only=False, is_argo=False)
    self_ns.prj_points_2d = types.MethodType(fsf["prj_points_2d"], self_ns)
    out = {}
    for tag, (ncam, ncls, H, W, dtype, scale) in {
        "nusc_small": (6, 10, 90, 160, np.uint8, 0.1),
        "nusc_mid": (6, 10, 450, 800, np.uint8, 0.5),
        "av2_small": (7, 1, 155, 205, np.int32, 0.1),
    }.items():
        n = 6000
        pts = np.empty((n, 3), dtype=np.float32)
        pts[:, 0] = rng.uniform(-50, 50, n)
        pts[:, 1] = rng.uniform(-50, 50, n)
        pts[:, 2] = rng.uniform(-4.99, 2.99, n)
        L = make_lidar2img(ncam, scale=scale, cx=W / 2 * 1.02 / scale, cy=H / 2 * 1.09 / scale)
        # special rows: behind the cameras / on the optical centre / tiny depth / far away / exact border ray
        pts[0] = [0.0, 0.0, 0.0]
        pts[1] = [1e-4, 0.0, 0.3]
        pts[2] = [5e-4, 1e-5, 0.3]
        pts[3] = [49.99, 49.99, 2.99]
        pts[4] = [-30.0, 0.01, -1.0]
        # points constructed to land on pixel-centre ties and image borders for camera 0 (optical axis = +x)
        fx = L[0][0, 1] * -1.0
        for j, u_pix in enumerate([0.0, 0.5, 1.0, 1.5, 2.5, W - 1.5, W - 1.0, W - 0.5, float(W)]):
            depth = 10.0
            # u = fx * (-y)/x + cx  (before the extrinsic translation); solve roughly, translation makes it inexact
            y = -(u_pix - L[0][0, 0]) / fx * depth
            pts[5 + j] = [depth, y, 0.0]
        mask, anno = make_mask(rng, ncam, ncls, H, W, 60, dtype)
        # overlaps: two classes on the same pixels, two cameras on the same frustum
        mask[0, 1 % ncls, H // 4: H // 2, W // 4: W // 2] = 61
        mask[0, 2 % ncls, H // 3: H // 2, W // 3: W // 2] = 62
        ids = fsf["points_in_mask"](self_ns, torch.from_numpy(pts), torch.from_numpy(mask), torch.from_numpy(L))
        p2d = fsf["prj_points_2d"](self_ns, torch.from_numpy(pts), torch.from_numpy(L), H, W)
        out[f"{tag}__points"] = pts
        out[f"{tag}__lidar2img"] = L
        out[f"{tag}__mask"] = mask
        out[f"{tag}__obj_id"] = ids.numpy()
        out[f"{tag}__pts_2d"] = p2d.numpy()
        if ncls == 10:
            # img_cross_attn middle part (FSF.py:716-719) + encode_2d_feats without the MLP (:537-551)
            anno_full = np.zeros((250, 9), dtype=np.float32)
            anno_full[: anno.shape[0]] = anno
            anno_full[60] = [1, 2, 3, 4, 0.77, 1, 0, 61, 1]
            anno_full[61] = [1, 2, 3, 4, 0.33, 2, 0, 62, 1]
            obj = ids
            cam_sel = obj.sum(-1).max(-1)[1]
            sel_mask = F.one_hot(cam_sel, ncam).bool().unsqueeze(-1)
            multi = obj.masked_select(sel_mask).reshape(-1, ncls)
            preds = fsf["get_all_cls_preds_2d"](self_ns, torch.from_numpy(anno_full)[None], torch.zeros(n, dtype=torch.long), multi)
            enc = fsf["encode_preds_2d"](self_ns, preds.reshape(-1, 9), W, H, encode_single_cls=False).reshape(n, ncls)
            out[f"{tag}__mask_anno"] = anno_full
            out[f"{tag}__cam_ids"] = multi.numpy()
            out[f"{tag}__score"] = enc.numpy()
    np.savez_compressed(os.path.join(OUT, "project.npz"), **out)


def gen_frustum_glue(fsf, sst, rng):
    """extract_fg_pts / double_overlap_pts / get_sir_coors / get_point_fg_weights /
    get_cluster_delta_weighted / get_single_cls_preds_2d (FSF.py:260-365, 476-504)."""
    self_ns = types.SimpleNamespace(num_classes=10, num_cams=6, encode_label_only=False, is_argo=False)
    self_ns.map_voxel_center_to_point = types.MethodType(fsf["map_voxel_center_to_point"], self_ns)
    n = 1500
    obj = np.zeros((n, 6, 10), dtype=np.int64)
    hit = rng.random(n) < 0.4
    for i in np.nonzero(hit)[0]:
        for _ in range(int(rng.integers(1, 4))):
            obj[i, rng.integers(6), rng.integers(10)] = rng.integers(1, 40)
    feat = rng.standard_normal((n, 12)).astype(np.float32)
    bz = rng.integers(0, 2, (n, 1)).astype(np.int64)
    pts = rng.uniform(-20, 20, (n, 5)).astype(np.float32)
    logits = rng.standard_normal((n, 11)).astype(np.float32)
    w = fsf["get_point_fg_weights"](self_ns, torch.from_numpy(logits))
    a = fsf["extract_fg_pts"](self_ns, torch.from_numpy(feat), torch.from_numpy(bz), torch.from_numpy(pts), torch.from_numpy(obj), w)
    b = fsf["double_overlap_pts"](self_ns, *a)
    sir_coors, obj_ids = fsf["get_sir_coors"](self_ns, b[1], b[3], b[4])
    f_cluster, center, ccoors = fsf["get_cluster_delta_weighted"](self_ns, b[2], sir_coors, b[4].unsqueeze(-1))
    anno = np.zeros((2, 250, 9), dtype=np.float32)
    anno[:, :, 4] = rng.uniform(0.1, 1, (2, 250))
    anno[:, :, 5] = rng.integers(0, 10, (2, 250))
    anno[:, :, 0:4] = rng.uniform(0, 100, (2, 250, 4))
    single = fsf["get_single_cls_preds_2d"](self_ns, torch.from_numpy(anno), ccoors)
    np.savez_compressed(
        os.path.join(OUT, "frustum_glue.npz"),
        obj_id=obj, feat=feat, bz=bz, points=pts, logits=logits, fg_weights=w.numpy(),
        fg_feat=a[0].numpy(), fg_bz=a[1].numpy(), fg_points=a[2].numpy(), fg_obj=a[3].numpy(), fg_w=a[4].numpy(),
        dup_feat=b[0].numpy(), dup_bz=b[1].numpy(), dup_points=b[2].numpy(), dup_obj=b[3].numpy(), dup_w=b[4].numpy(),
        sir_coors=sir_coors.numpy(), f_cluster=f_cluster.numpy(), cluster_center=center.numpy(),
        cluster_coors=ccoors.numpy(), mask_anno=anno, single_preds=single.numpy(),
    )


def gen_neck(rng):
    fwd = lift_methods(os.path.join(PLUGIN, "models/necks/voxel2point_neck.py"), "Voxel2PointScatterNeck", ["forward"],
                       {"torch": torch})["forward"]
    vs, rng_pc = [0.2, 0.2, 0.2], [-51.2, -51.2, -5, 51.2, 51.2, 3]
    self_ns = types.SimpleNamespace(voxel_size=vs, point_cloud_range=rng_pc, with_xyz=True, normalize_local_xyz=False,
                                    training=False)
    n, m, c = 900, 200, 128
    pts = np.empty((n, 5), dtype=np.float32)
    pts[:, 0] = rng.uniform(-50, 50, n)
    pts[:, 1] = rng.uniform(-50, 50, n)
    pts[:, 2] = rng.uniform(-4.99, 2.99, n)
    pts[:, 3:] = rng.random((n, 2))
    coors = np.concatenate([np.zeros((n, 1)), np.floor((pts[:, [2, 1, 0]] - np.array(rng_pc[:3], dtype=np.float32)[[2, 1, 0]]) / 0.2)], 1).astype(np.int64)
    vf = rng.standard_normal((m, c)).astype(np.float32)
    vf[5] = -1.0  # a padded (dropped) voxel row
    vf[9] = -1.0
    inv = rng.integers(0, m, n).astype(np.int64)
    out, mask = fwd(self_ns, torch.from_numpy(pts), torch.from_numpy(coors), torch.from_numpy(vf), torch.from_numpy(inv), -1)
    np.savez_compressed(os.path.join(OUT, "neck.npz"), points=pts, coors=coors, voxel_feats=vf, inv=inv,
                        out=out.numpy(), mask=mask.numpy(), voxel_size=np.array(vs, dtype=np.float32),
                        pc_range=np.array(rng_pc, dtype=np.float32))


def gen_divfloor(rng):
    """The expression of single_stage_fsd.py:591 / :948 evaluated by torch itself, incl. exact boundaries."""
    out = {}
    for tag, vs in {"v01": (0.1, 0.1, 0.1), "v03": (0.3, 0.3, 8.0), "v005": (0.05, 0.05, 8.0), "v02": (0.2, 0.2, 0.2)}.items():
        mn = np.array([-51.2, -51.2, -5.0], dtype=np.float32)
        v = np.array(vs, dtype=np.float32)
        n = 6000
        p = np.empty((n, 3), dtype=np.float32)
        p[:, 0] = rng.uniform(-50, 50, n)
        p[:, 1] = rng.uniform(-50, 50, n)
        p[:, 2] = rng.uniform(-4.99, 2.99, n)
        # exact grid boundaries and their +-1 ulp neighbours
        k = np.arange(0, 513, dtype=np.float32)
        b = (k * v[0] + mn[0]).astype(np.float32)
        bb = np.concatenate([b, np.nextafter(b, np.float32(1e9)), np.nextafter(b, np.float32(-1e9))])
        bb = bb[(bb > -51.2) & (bb < 51.2)]
        q = np.zeros((bb.size, 3), dtype=np.float32)
        q[:, 0] = bb
        q[:, 1] = bb[::-1]
        q[:, 2] = np.clip(bb / 12.8, -4.99, 2.99)
        p = np.concatenate([p, q], 0)
        coors = torch.div(torch.from_numpy(p) - torch.from_numpy(mn)[None], torch.from_numpy(v)[None], rounding_mode="floor").long()
        out[f"{tag}__points"] = p
        out[f"{tag}__voxel"] = v
        out[f"{tag}__min"] = mn
        out[f"{tag}__coors_xyz"] = coors.numpy()
    np.savez_compressed(os.path.join(OUT, "divfloor.npz"), **out)


def gen_sir_flow(sst, rng):
    """SIR.forward control flow (models/backbones/sir.py:65-85) with a recording stand-in for the
    un-vendored SIRLayer: pins what each block receives (cat of points and feats, shared unique) and how
    the outputs are concatenated."""
    fwd = lift_methods(os.path.join(PLUGIN, "models/backbones/sir.py"), "SIR", ["forward"], {"torch": torch})["forward"]

    class FakeLayer:
        def __init__(self, idx):
            self.idx = idx
            self.seen = None

        def __call__(self, in_feats, coors, f_cluster, return_both=False, unq_inv_once=None, new_coors_once=None):
            self.seen = (in_feats.clone(), unq_inv_once.clone(), new_coors_once.clone())
            m = new_coors_once.size(0)
            pts = in_feats[:, :4] * (self.idx + 1)
            grp = torch.zeros(m, 3).index_add_(0, unq_inv_once, in_feats[:, :3]) + self.idx
            if return_both:
                return pts, grp, new_coors_once
            return pts, grp

    layers = [FakeLayer(i) for i in range(3)]
    self_ns = types.SimpleNamespace(unique_once=True, num_blocks=3, block_list=layers)
    n = 500
    points = rng.standard_normal((n, 5)).astype(np.float32)
    feats = rng.standard_normal((n, 7)).astype(np.float32)
    coors = np.stack([rng.integers(0, 2, n), np.zeros(n), rng.integers(0, 30, n)], 1).astype(np.int64)
    f_cluster = rng.standard_normal((n, 3)).astype(np.float32)
    out_feats, cluster_feats, out_coors = fwd(self_ns, torch.from_numpy(points), torch.from_numpy(feats),
                                              torch.from_numpy(coors), torch.from_numpy(f_cluster))
    np.savez_compressed(os.path.join(OUT, "sir_flow.npz"), points=points, feats=feats, coors=coors, f_cluster=f_cluster,
                        out_feats=out_feats.numpy(), cluster_feats=cluster_feats.numpy(), out_coors=out_coors.numpy(),
                        block1_in=layers[1].seen[0].numpy(), block2_in=layers[2].seen[0].numpy(),
                        unq_inv=layers[0].seen[1].numpy(), new_coors=layers[0].seen[2].numpy())


def gen_refine_glue(rng):
    """Query-refinement glue that IS vendored python in the reference: BasePointBBoxCoder.encode/decode
    (core/bbox/coders/base_point_bbox_coder.py:36-82), FSF.combine_frustum_and_fsd / decode_stage_bboxes
    (FSF.py:657-692, 1085-1094), FullySparseBboxHead.get_nonempty_roi_mask / align_roi_feature_and_rois
    (fsd_bbox_head.py:153-197) and FrustumClusterHead._get_bboxes_single (frustum_cluster_head.py:593-697) with
    the un-vendored mmdet3d symbols it calls replaced by recording stand-ins (no suppression: every box above the
    score threshold is kept, class by class), which pins everything around the NMS call."""
    coder_fns = lift_methods(os.path.join(PLUGIN, "core/bbox/coders/base_point_bbox_coder.py"), "BasePointBBoxCoder",
                             ["encode", "decode"], {"torch": torch})
    coder = types.SimpleNamespace(code_size=10, EPS=1e-6)
    coder.decode = types.MethodType(coder_fns["decode"], coder)
    n = 300
    base = torch.from_numpy(rng.uniform(-40, 40, (n, 3)).astype(np.float32))
    reg = torch.from_numpy(rng.normal(0, 0.6, (n, 10)).astype(np.float32))
    boxes = coder_fns["decode"](coder, reg, base)
    enc = coder_fns["encode"](coder, boxes, base)

    fsf = lift_methods(os.path.join(PLUGIN, "models/detectors/FSF.py"), "FSF", ["combine_frustum_and_fsd", "decode_stage_bboxes"],
                       {"torch": torch})
    ns = types.SimpleNamespace(fsd_begin_idx=1000, bbox_coder=coder, combine_frustum_feat_mlp=lambda x: x[:, :6] * 2.0,
                               combine_fsd_feat_mlp=lambda x: x[:, :6] - 1.0)
    nf, nl = 40, 110
    f_centers = torch.from_numpy(rng.uniform(-30, 30, (nf, 3)).astype(np.float32))
    l_centers = torch.from_numpy(rng.uniform(-30, 30, (nl, 3)).astype(np.float32))
    f_coors = torch.stack([torch.zeros(nf), torch.zeros(nf), torch.arange(1, nf + 1).float()], 1).long()
    l_coors = torch.stack([torch.from_numpy(rng.integers(0, 6, nl)), torch.zeros(nl).long(), torch.arange(nl)], 1).long()
    f_res = dict(cls_logits=[torch.from_numpy(rng.normal(0, 1, (nf, 10)).astype(np.float32))],
                 reg_preds=[torch.from_numpy(rng.normal(0, 0.5, (nf, 10)).astype(np.float32))])
    l_res = dict(cls_logits=[torch.from_numpy(rng.normal(0, 1, (nl, 10)).astype(np.float32))],
                 reg_preds=[torch.from_numpy(rng.normal(0, 0.5, (nl, 10)).astype(np.float32))])
    f_feats = torch.from_numpy(rng.normal(0, 1, (nf, 9)).astype(np.float32))
    l_feats = torch.from_numpy(rng.normal(0, 1, (nl, 8)).astype(np.float32))
    f_p2d = torch.from_numpy(rng.uniform(0, 1, (nf, 9)).astype(np.float32))
    c_centers, c_coors, c_res, c_feats, c_p2d = fsf["combine_frustum_and_fsd"](ns, f_centers, f_coors, f_res, f_feats, f_p2d,
                                                                              l_centers, l_coors, l_res, l_feats)
    rois = fsf["decode_stage_bboxes"](ns, c_centers, c_coors[:, 0], c_res["reg_preds"])

    head = lift_methods(os.path.join(PLUGIN, "models/roi_heads/bbox_heads/fsd_bbox_head.py"), "FullySparseBboxHead",
                        ["get_nonempty_roi_mask", "align_roi_feature_and_rois"], {"torch": torch})
    hns = types.SimpleNamespace(training=False)
    out_coors = torch.tensor([-1, 0, 3, 4, 9])
    feats = torch.from_numpy(rng.normal(0, 1, (5, 7)).astype(np.float32))
    mask = head["get_nonempty_roi_mask"](hns, out_coors, 12)
    aligned = head["align_roi_feature_and_rois"](hns, feats, out_coors, 12)

    class Boxes:  # what `input_meta['box_type_3d']` has to offer to _get_bboxes_single
        def __init__(self, tensor, box_dim=7):
            self.tensor = tensor

        @property
        def bev(self):
            return self.tensor[:, [0, 1, 3, 4, 6]]

    def xywhr2xyxyr(b):
        out = torch.zeros_like(b)
        out[:, 0], out[:, 1] = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2
        out[:, 2], out[:, 3], out[:, 4] = b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2, b[:, 4]
        return out

    seen = {}

    def keep_all_nms(bboxes, bboxes_for_nms, scores, score_thr, max_num, cfg):
        seen["for_nms"] = bboxes_for_nms.clone()
        bb, ss, ll = [], [], []
        for i in range(scores.shape[1] - 1):
            sel = scores[:, i] > score_thr
            bb.append(bboxes[sel]); ss.append(scores[sel, i]); ll.append(torch.full((int(sel.sum()),), i, dtype=torch.long))
        return torch.cat(bb), torch.cat(ss), torch.cat(ll)

    gb = lift_methods(os.path.join(PLUGIN, "models/dense_heads/frustum_cluster_head.py"), "FrustumClusterHead",
                      ["_get_bboxes_single"], {"torch": torch, "xywhr2xyxyr": xywhr2xyxyr, "box3d_multiclass_nms": keep_all_nms})
    classes = ["car", "truck", "trailer", "bus", "construction_vehicle", "bicycle", "motorcycle", "pedestrian", "traffic_cone",
               "barrier"]
    task_names = ["bus", "car", "pedestrian"]  # task-local label order differs from the global one on purpose
    cfg = dict(use_rotate_nms=True, nms_pre=150, nms_thr=0.35, score_thr=0.3, min_bbox_size=0, max_num=500)

    class Cfg(dict):
        __getattr__ = dict.__getitem__

    hd = types.SimpleNamespace(as_rpn=False, training=False, test_cfg=Cfg(cfg), tasks=[dict(class_names=task_names)],
                               box_code_size=10, bbox_coder=coder, vis_dir=None, class_names=classes)
    cls_logits = torch.from_numpy(rng.normal(0, 1.5, (200, 3)).astype(np.float32))
    reg2 = torch.from_numpy(rng.normal(0, 0.5, (200, 10)).astype(np.float32))
    xyz2 = torch.from_numpy(rng.uniform(-30, 30, (200, 3)).astype(np.float32))
    ob, os_, ol = gb["_get_bboxes_single"](hd, 0, cls_logits, None, reg2, torch.zeros(200, 9), xyz2, dict(box_type_3d=Boxes))
    np.savez_compressed(
        os.path.join(OUT, "refine_glue.npz"),
        base=base.numpy(), reg=reg.numpy(), boxes=boxes.numpy(), enc=enc.numpy(),
        f_centers=f_centers.numpy(), l_centers=l_centers.numpy(), f_coors=f_coors.numpy(), l_coors=l_coors.numpy(),
        f_cls=f_res["cls_logits"][0].numpy(), f_reg=f_res["reg_preds"][0].numpy(), l_cls=l_res["cls_logits"][0].numpy(),
        l_reg=l_res["reg_preds"][0].numpy(), f_feats=f_feats.numpy(), l_feats=l_feats.numpy(), f_p2d=f_p2d.numpy(),
        c_centers=c_centers.numpy(), c_coors=c_coors.numpy(), c_cls=c_res["cls_logits"][0].numpy(),
        c_reg=c_res["reg_preds"][0].numpy(), c_feats=c_feats.numpy(), c_p2d=c_p2d.numpy(), rois=rois.numpy(),
        out_coors=out_coors.numpy(), roi_feats=feats.numpy(), roi_mask=mask.numpy(), roi_aligned=aligned.numpy(),
        gb_cls=cls_logits.numpy(), gb_reg=reg2.numpy(), gb_xyz=xyz2.numpy(), gb_for_nms=seen["for_nms"].numpy(),
        gb_boxes=ob.tensor.numpy(), gb_scores=os_.numpy(), gb_labels=ol.numpy())


def gen_input_pipeline(rng):
    """Input side by the reference's own pipeline classes, lifted from
    projects/mmdet3d_plugin/datasets/pipelines/loading.py: LoadMaskFromFiles.load_nusc / reorg_anno_multi_cls /
    reorg_anno_single_cls / pad_tensor (:213-339), MyLoadPointsFromFile.__call__ (:660-700),
    MyLoadPointsFromMultiSweeps.__call__ / _remove_close (:781-877), SaveNoAugPoints (:341-354), NormalizePoints
    (:537-563).  cv2.imread(path, -1) is served by PIL; mmcv.FileClient raises ConnectionError so the np.fromfile branch
    runs.  Stores the synthetic files' CONTENTS and the classes' outputs."""
    import json
    import tempfile
    from PIL import Image

    path = os.path.join(PLUGIN, "datasets/pipelines/loading.py")

    class Pts:  # stand-in for mmdet3d BasePoints / LiDARPoints as far as these methods use it
        def __init__(self, tensor, points_dim=None, attribute_dims=None):
            self.tensor = torch.as_tensor(np.ascontiguousarray(tensor), dtype=torch.float32).reshape(-1, points_dim or np.shape(tensor)[-1]).clone()

        def new_point(self, data):
            return Pts(data, self.tensor.shape[1])

        @classmethod
        def cat(cls, lst):
            return cls(torch.cat([p.tensor for p in lst], 0))

        def __getitem__(self, item):
            return Pts(self.tensor[item])

    class FileClient:
        def __init__(self, **kw):
            pass

        def get(self, name):
            raise ConnectionError

    cv2 = types.SimpleNamespace(imread=lambda p, flag: np.array(Image.open(p)))
    mmcv = types.SimpleNamespace(FileClient=FileClient, check_file_exist=lambda p: None)
    g = {"torch": torch, "np": np, "os": os, "json": json, "cv2": cv2, "mmcv": mmcv, "BasePoints": Pts,
         "get_points_type": lambda coord: Pts}
    mask_fns = lift_methods(path, "LoadMaskFromFiles", ["load_nusc", "reorg_anno_multi_cls", "reorg_anno_single_cls", "pad_tensor"], g)
    file_fns = lift_methods(path, "MyLoadPointsFromFile", ["__call__", "_load_points"], g)
    sweep_fns = lift_methods(path, "MyLoadPointsFromMultiSweeps", ["__call__", "_load_points", "_remove_close"], g)
    save_fns = lift_methods(path, "SaveNoAugPoints", ["__call__"], g)
    norm_fns = lift_methods(path, "NormalizePoints", ["__call__"], g)

    classes = ["car", "truck", "trailer", "bus", "construction_vehicle", "bicycle", "motorcycle", "pedestrian", "traffic_cone",
               "barrier"]
    H, W = 45, 80
    planes = np.zeros((6, 10, H, W), dtype=np.uint8)
    anno = [dict() for _ in range(6)]
    ids = rng.permutation(np.arange(1, 41))  # obj ids NOT in camera order: reorg sorts by id
    for n, oid in enumerate(ids):
        cam, cls = int(rng.integers(6)), int(rng.integers(10))
        x1, y1 = int(rng.integers(0, W - 12)), int(rng.integers(0, H - 8))
        w, h = int(rng.integers(3, 12)), int(rng.integers(3, 8))
        planes[cam, cls, y1:y1 + h, x1:x1 + w] = oid
        anno[cam].setdefault(classes[cls], []).append(dict(bbox=[float(x1), float(y1), float(x1 + w), float(y1 + h)],
                                                           score=float(rng.uniform(0.1, 1)), category=cls, cam_id=cam,
                                                           obj_id=int(oid)))
    key = np.stack([rng.uniform(-60, 60, 3000), rng.uniform(-60, 60, 3000), rng.uniform(-6, 4, 3000), rng.uniform(0, 255, 3000),
                    rng.integers(0, 32, 3000)], 1).astype(np.float32)
    key[:50, :2] = rng.uniform(-0.9, 0.9, (50, 2))  # points the close filter has to drop
    sweeps, sweep_meta = [], []
    for k in range(3):
        sw = key + rng.normal(0, 0.3, key.shape).astype(np.float32)
        ang = 0.02 * (k + 1)
        rot = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
        sweeps.append(sw.astype(np.float32))
        sweep_meta.append(dict(timestamp=1.5e15 - 5e4 * (k + 1), sensor2lidar_rotation=rot.tolist(),
                               sensor2lidar_translation=[0.4 * (k + 1), -0.1 * k, 0.02]))
    with tempfile.TemporaryDirectory() as tmp:
        sdir = os.path.join(tmp, "masks", "sample0")
        os.makedirs(sdir)
        for cam in range(6):
            for ci, name in enumerate(classes):
                Image.fromarray(planes[cam, ci]).save(os.path.join(sdir, f"{cam}_{name}.png"))
        json.dump(anno, open(os.path.join(sdir, "anno.json"), "w"))
        key.tofile(os.path.join(tmp, "key.bin"))
        for k, sw in enumerate(sweeps):
            sw.tofile(os.path.join(tmp, f"sweep{k}.bin"))
            sweep_meta[k]["data_path"] = os.path.join(tmp, f"sweep{k}.bin")
        m_ns = types.SimpleNamespace(data_path=os.path.join(tmp, "masks"), obj_max_num=250, class_names=classes)
        for n in ("reorg_anno_multi_cls", "reorg_anno_single_cls", "pad_tensor"):
            setattr(m_ns, n, types.MethodType(mask_fns[n], m_ns))
        res = mask_fns["load_nusc"](m_ns, dict(sample_idx="sample0"))
        f_ns = types.SimpleNamespace(load_dim=5, use_dim=[0, 1, 2, 3, 4], coord_type="LIDAR", shift_height=False, use_color=False,
                                     virtual_path=None, file_client=None, file_client_args={})
        f_ns._load_points = types.MethodType(file_fns["_load_points"], f_ns)
        r = file_fns["__call__"](f_ns, dict(pts_filename=os.path.join(tmp, "key.bin")))
        loaded = r["points"].tensor.clone()
        s_ns = types.SimpleNamespace(load_dim=5, sweeps_num=9, use_dim=[0, 1, 2, 3, 4], pad_empty_sweeps=True, remove_close=True,
                                     test_mode=True, virtual_path=None, file_client=None, file_client_args={})
        s_ns._load_points = types.MethodType(sweep_fns["_load_points"], s_ns)
        s_ns._remove_close = types.MethodType(sweep_fns["_remove_close"], s_ns)
        as_np = [dict(m, sensor2lidar_rotation=np.array(m["sensor2lidar_rotation"]),
                      sensor2lidar_translation=np.array(m["sensor2lidar_translation"])) for m in sweep_meta]
        r.update(timestamp=1.5e9, sweeps=as_np)
        r = sweep_fns["__call__"](s_ns, r)
        multi = r["points"].tensor.clone()
        r_pad = file_fns["__call__"](f_ns, dict(pts_filename=os.path.join(tmp, "key.bin")))
        r_pad.update(timestamp=1.5e9, sweeps=[])
        s_ns.sweeps_num = 2
        padded = sweep_fns["__call__"](s_ns, r_pad)["points"].tensor.clone()
        r = save_fns["__call__"](types.SimpleNamespace(), r)
        saved = r["points"].tensor.clone()
        r = norm_fns["__call__"](types.SimpleNamespace(dims=[3], std=[255], mean=[0]), r)
        normed = r["points"].tensor.clone()
    single = mask_fns["reorg_anno_single_cls"](m_ns, [[dict(bbox=[1.0, 2.0, 3.0, 4.0], score=0.5, category=3, cam_id=0, obj_id=7)], [],
                                                      [dict(bbox=[5.0, 6.0, 7.0, 8.0], score=0.25, category=1, cam_id=2, obj_id=2)]])
    for m in sweep_meta:
        m.pop("data_path")
    np.savez_compressed(
        os.path.join(OUT, "input_pipeline.npz"),
        planes=planes, anno_json=np.frombuffer(json.dumps(anno).encode(), dtype=np.uint8), key=key, sweeps=np.stack(sweeps),
        sweep_meta_json=np.frombuffer(json.dumps(sweep_meta).encode(), dtype=np.uint8),
        mask_data=res["mask_data"].numpy(), mask_anno=res["mask_anno"].numpy(), loaded=loaded.numpy(), multi=multi.numpy(),
        padded=padded.numpy(), saved=saved.numpy(), normed=normed.numpy(), single_anno=single.numpy())


def main():
    assert os.path.isdir(REF), "the reference tree is only available in the build container"
    install_stubs()
    sst = import_sst_ops()
    fsf_names = ["prj_points_2d", "points_in_mask", "double_overlap_pts", "extract_fg_pts", "get_point_fg_weights",
                 "get_sir_coors", "encode_preds_2d", "get_single_cls_preds_2d", "get_all_cls_preds_2d",
                 "split_points_last_3dim", "combine_by_batch", "get_cluster_delta_weighted", "map_voxel_center_to_point"]
    fsf = lift_methods(os.path.join(PLUGIN, "models/detectors/FSF.py"), "FSF", fsf_names,
                       {"torch": torch, "F": F, "scatter_v2": sst.scatter_v2})
    rng = np.random.default_rng(20260928)
    torch.manual_seed(0)
    gen_scatter(sst, rng)
    gen_project(fsf, rng)
    gen_frustum_glue(fsf, sst, rng)
    gen_neck(rng)
    gen_divfloor(rng)
    gen_sir_flow(sst, rng)
    gen_refine_glue(np.random.default_rng(777))
    gen_input_pipeline(np.random.default_rng(4242))
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
