"""The registries the reference's configs select types from.  In the reference they come
from mmcv/mmdet/mmdet3d/mmseg (`DETECTORS`, `BACKBONES`, ... e.g. `from mmdet.models import BACKBONES`,
projects/mmdet3d_plugin/models/backbones/sir.py:1); here they are local `compat.Registry` instances with the
same names, populated by importing `fullysparsefusion_amd.mmdet3d_plugin`."""
import torch.nn as nn

from ..compat import Registry, build_from_cfg

# mmdet >= 2.12 keeps ONE model registry under several names (that is why the reference can `build_head` a
# BACKBONES-registered SIR, FSF.py:119); mmseg's SEGMENTORS and mmdet3d's encoder registries are separate.
MODELS = Registry("models")
DETECTORS = BACKBONES = NECKS = HEADS = ROI_EXTRACTORS = SHARED_HEADS = LOSSES = MODELS
SEGMENTORS = Registry("segmentor")
VOXEL_ENCODERS = Registry("voxel_encoder")
MIDDLE_ENCODERS = Registry("middle_encoder")
BBOX_CODERS = Registry("bbox_coder")
BBOX_ASSIGNERS = Registry("bbox_assigner")
PIPELINES = Registry("pipeline")
DATASETS = Registry("dataset")
HOOKS = Registry("hook")
NORM_LAYERS = Registry("norm_layer")
CONV_LAYERS = Registry("conv_layer")


def build_detector(cfg, train_cfg=None, test_cfg=None):
    args = {}
    if train_cfg is not None:
        args["train_cfg"] = train_cfg
    if test_cfg is not None:
        args["test_cfg"] = test_cfg
    reg = DETECTORS if cfg["type"] in DETECTORS else SEGMENTORS
    return build_from_cfg(cfg, reg, args or None)


build_model = build_detector


def build_backbone(cfg):
    return build_from_cfg(cfg, BACKBONES)


def build_neck(cfg):
    return build_from_cfg(cfg, NECKS)


def build_head(cfg):
    return build_from_cfg(cfg, HEADS)


def build_voxel_encoder(cfg):
    return build_from_cfg(cfg, VOXEL_ENCODERS)


def build_middle_encoder(cfg):
    return build_from_cfg(cfg, MIDDLE_ENCODERS)


def build_roi_extractor(cfg):
    return build_from_cfg(cfg, ROI_EXTRACTORS)


def build_loss(cfg):
    return build_from_cfg(cfg, LOSSES)


def build_norm_layer(cfg, num_features, postfix=""):
    """mmcv.cnn.build_norm_layer: returns (name, layer).  `requires_grad` and the layer kwargs come from cfg."""
    cfg = dict(cfg)
    layer_type = cfg.pop("type")
    requires_grad = cfg.pop("requires_grad", True)
    cls = NORM_LAYERS.get(layer_type)
    if cls is None:
        raise KeyError(f"Unrecognized norm type {layer_type}")
    if layer_type in ("LN",):
        layer = cls(num_features, **cfg)
        abbr = "ln"
    else:
        cfg.setdefault("eps", 1e-5)
        layer = cls(num_features, **cfg)
        abbr = "bn"
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


NORM_LAYERS.register_module("LN", module=nn.LayerNorm)
NORM_LAYERS.register_module("BN1d", module=nn.BatchNorm1d)
NORM_LAYERS.register_module("BN", module=nn.BatchNorm1d)
