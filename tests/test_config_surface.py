"""CPU: the drop-in boundary b1: mmcv-style configs load through compat.Config, every `type=`
resolves against the plugin registries, and the model builds with the reference's own kwargs."""
import os

import pytest

from fullysparsefusion_amd import mmdet3d_plugin as plugin
from fullysparsefusion_amd.compat import Config
from fullysparsefusion_amd.mmdet3d_plugin import registry as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = "/root/reference/projects/configs/nuScenes/FSF_nuScenes_config.py"
# types owned by mmcv's runner / torch.optim / the conv_cfg-act_cfg mini-dialect, not by plugin registries
EXTERNAL = {"AdamW", "Conv1d", "ReLU", "EpochBasedRunner", "TensorboardLoggerHook", "TextLoggerHook", "cyclic", "CosineAnnealing"}
REGS = [R.MODELS, R.SEGMENTORS, R.VOXEL_ENCODERS, R.MIDDLE_ENCODERS, R.BBOX_CODERS, R.BBOX_ASSIGNERS, R.PIPELINES, R.DATASETS,
        R.HOOKS, R.NORM_LAYERS]


def unresolved(cfg_dict):
    missing = set()

    def walk(o):
        if isinstance(o, dict):
            t = o.get("type")
            if isinstance(t, str) and t not in EXTERNAL and not any(t in r for r in REGS):
                missing.add(t)
            for v in o.values():
                walk(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v)

    walk(cfg_dict)
    return missing


def test_own_config_builds():
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "fsf_nuscenes.py"))
    assert not unresolved(cfg.to_dict())
    model = plugin.build_model(cfg.model)
    assert type(model).__name__ == "FSF"
    # channel arithmetic that must close
    assert model.segmentor.voxel_encoder.vfe_layers[0].linear.in_features == 11
    assert model.segmentor.voxel_encoder.vfe_layers[1].linear.in_features == 128
    assert model.backbone.block_list[0].vfe_layers[0].linear.in_features == 5 + 11 + 33 + 131
    assert model.frustum_sir.block_list[0].vfe_layers[0].linear.in_features == 5 + 131
    assert model.segmentor.segmentation_head.conv_seg.out_features == 11
    assert model.segmentor.segmentation_head.voting.out_features == 33
    unet = model.segmentor.backbone
    n_conv = sum(1 for m in unet.modules() if type(m).__name__ in ("SubMConv3d", "SparseConv3d", "SparseInverseConv3d"))
    assert n_conv == 34
    assert float(model.segmentor_updated_mlp[-1].weight.detach().abs().sum()) == 0.0  # zero-init (FSF.py:142-143)
    keys = model.state_dict().keys()
    for k in ["segmentor.voxel_encoder.vfe_layers.0.linear.weight", "segmentor.backbone.conv_input.0.weight",
              "segmentor.backbone.encoder_layers.encoder_layer2.0.0.weight", "segmentor.backbone.lateral_layer5.conv1.weight",
              "segmentor.backbone.upsample_layer1.0.weight", "backbone.block_list.0.rel_mlp.0.0.weight",
              "frustum_sir.block_list.2.vfe_layers.1.norm.weight"]:
        assert k in keys, k
    assert tuple(model.state_dict()["segmentor.backbone.conv_input.0.weight"].shape) == (3, 3, 3, 64, 64)


def test_config_merge_semantics(tmp_path):
    base = tmp_path / "base.py"
    base.write_text("a = dict(x=1, y=dict(z=2, w=3))\nb = [1, 2]\n")
    child = tmp_path / "child.py"
    child.write_text("_base_ = ['base.py']\na = dict(y=dict(z=5))\nc = dict(_delete_=True, q=1)\n")
    cfg = Config.fromfile(str(child))
    assert cfg.a.x == 1 and cfg.a.y.z == 5 and cfg.a.y.w == 3 and cfg.b == [1, 2] and cfg.c == dict(q=1)
    cfg.merge_from_dict({"a.y.w": 9, "d": 4})
    assert cfg.a.y.w == 9 and cfg.d == 4


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason="reference tree only exists in the build container")
def test_reference_config_loads_and_builds_unchanged():
    cfg = Config.fromfile(REF_CFG)
    assert cfg.plugin is True and cfg.plugin_dir == "projects/mmdet3d_plugin/"
    assert not unresolved(cfg.to_dict()), unresolved(cfg.to_dict())
    model = plugin.build_model(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    own = Config.fromfile(os.path.join(ROOT, "configs", "fsf_nuscenes.py"))
    for key in ["segmentor", "backbone", "frustum_sir", "cluster_assigner", "test_cfg", "mlp_cfg", "bbox_coder", "roi_extractor",
                "single_refine_sir_layer", "refine_encode_2d_mlp_cfg"]:
        assert cfg.model[key] == own.model[key], key
    for head in ["bbox_head", "frustum_obj_head"]:  # inference-relevant head arguments (assigners / losses are train-time)
        for key in ["type", "num_classes", "bbox_coder", "in_channel", "shared_mlp_dims", "tasks", "class_names", "common_attrs",
                    "num_cls_layer", "cls_hidden_dim", "separate_head", "norm_cfg"]:
            assert cfg.model[head][key] == own.model[head][key], (head, key)
    assert cfg.model.refined_obj_head[0].test_cfg == own.model.refined_obj_head[0].test_cfg
    assert cfg.model.frustum_obj_head.test_cfg == own.model.frustum_obj_head.test_cfg
    own_model = plugin.build_model(own.model)
    ref_sd, own_sd = model.state_dict(), own_model.state_dict()
    assert {k: tuple(v.shape) for k, v in ref_sd.items()} == {k: tuple(v.shape) for k, v in own_sd.items()}
    assert model.num_extra_stages == own_model.num_extra_stages == 1
    assert sum(p.numel() for p in model.parameters()) > 80e6  # the full detector, refine stage and heads included


REF_AV2 = "/root/reference/projects/configs/Argoverse2/FSF_AV2_config.py"


@pytest.mark.skipif(not os.path.exists(REF_AV2), reason="reference tree only exists in the build container")
def test_reference_av2_config_resolves():
    cfg = Config.fromfile(REF_AV2)
    assert cfg.model.is_argo is True and cfg.model.num_cams == 7
    assert not unresolved(cfg.to_dict()), unresolved(cfg.to_dict())
    seg = plugin.registry.build_detector(cfg.model.segmentor)
    assert seg.backbone.sparse_shape == [32, 2048, 2048]
    assert seg.voxel_layer.grid_size.tolist() == [2048, 2048, 32]
    # the repository's own AV2 model file describes the same model (parameters and inference-relevant arguments)
    own = Config.fromfile(os.path.join(ROOT, "configs", "fsf_av2.py"))
    for key in ["segmentor", "backbone", "frustum_sir", "cluster_assigner", "test_cfg", "mlp_cfg", "bbox_coder", "roi_extractor",
                "single_refine_sir_layer", "refine_encode_2d_mlp_cfg", "encode_2d_mlp_cfg", "segmentor_updated_mlp", "is_argo",
                "num_cams", "num_classes"]:
        assert cfg.model[key] == own.model[key], key
    ref_model = plugin.build_model(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    own_model = plugin.build_model(own.model)
    assert {k: tuple(v.shape) for k, v in ref_model.state_dict().items()} == \
           {k: tuple(v.shape) for k, v in own_model.state_dict().items()}
    assert own_model.bbox_coder.code_size == 8 and own_model.is_argo


def test_grouped_concat_host_logic_cpu():
    """GroupedConcat (the deferred `cat([point_feats, group_feats[inv]], 1)` of SIRLayer / DynamicScatterVFE): shape
    protocol, and the grouped kernel route declines CPU tensors (the plugin ops themselves have no CPU path and raise)."""
    import pytest
    import torch

    from fullysparsefusion_amd._lib import FsfHipError
    from fullysparsefusion_amd.mmdet3d_plugin.ops import sst_ops

    torch.manual_seed(0)
    n, g, c = 300, 17, 8
    p, grp = torch.randn(n, c), torch.randn(g, c)
    inv = torch.randint(0, g, (n,))
    gc = sst_ops.GroupedConcat(p, grp, inv)
    assert gc.shape == (n, 2 * c) and gc.size(1) == 2 * c and gc.size() == (n, 2 * c)
    assert sst_ops._grouped_linear_norm_act(torch.nn.Linear(2 * c, c), torch.nn.LayerNorm(c), torch.nn.GELU(), gc) is None
    with pytest.raises(FsfHipError):
        gc.materialize()  # gathers through the HIP library: loud failure on a CPU tensor, never a silent fallback


def test_sorted_sir_path_refuses_a_last_layer_with_a_residual():
    """`SIRLayer._run_vfe` adds `vfe(features) + features` on the LAST layer whenever its output is as wide as its (concatenated) input —
    feat_channels=[64, 128]: cat(point 64, group 64) = 128 -> 128.  The sorted K22s path has no residual, so `sorted_supported()` must
    send such a block through the unsorted path; the FSF configs' blocks ([128, 128]: 256 -> 128) stay on K22s."""
    from fullysparsefusion_amd.mmdet3d_plugin.models.voxel_encoders.voxel_encoder import SIRLayer

    kw = dict(in_channels=36, rel_mlp_hidden_dims=[16, 32], norm_cfg=dict(type="LN", eps=1e-3), mode="max", act="gelu")
    residual = SIRLayer(feat_channels=[64, 128], with_shortcut=True, **kw).eval()
    assert residual.vfe_layers[-1].linear.in_features == residual.vfe_layers[-1].linear.out_features == 128
    assert not residual.sorted_supported()
    assert SIRLayer(feat_channels=[64, 128], with_shortcut=False, **kw).eval().sorted_supported()
    assert SIRLayer(feat_channels=[128, 128], with_shortcut=True, **kw).eval().sorted_supported()
    one = SIRLayer(in_channels=64, feat_channels=[64], with_shortcut=True, rel_mlp_hidden_dims=[16, 32],
                   norm_cfg=dict(type="LN", eps=1e-3), mode="max", act="gelu").eval()
    assert not one.sorted_supported()
