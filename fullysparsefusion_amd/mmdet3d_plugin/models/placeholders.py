"""Registered-by-name stand-ins for the config types that are OUT OF SCOPE of the hot path (SURVEY.md §2.1 rows
6-12): unused head variants, label assigners, losses, dataset pipelines and hooks.  They exist so
that `projects/configs/nuScenes/FSF_nuScenes_config.py` resolves every `type=` and the model builds; calling one
raises, naming what is missing — they never compute anything (no silent fallbacks)."""
import torch.nn as nn

from ..registry import BBOX_ASSIGNERS, BBOX_CODERS, DATASETS, HOOKS, MODELS, PIPELINES, VOXEL_ENCODERS


def _module_placeholder(name, where):
    class _P(nn.Module):
        OUT_OF_SCOPE = True

        def __init__(self, **kwargs):
            super().__init__()
            self.cfg = kwargs

        def forward(self, *a, **k):
            raise NotImplementedError(f"{name} ({where}) is outside the MI355X hot path built so far")

    _P.__name__ = _P.__qualname__ = name
    return _P


def _plain_placeholder(name, where):
    class _P:
        OUT_OF_SCOPE = True

        def __init__(self, *args, **kwargs):
            self.args, self.cfg = args, kwargs

        def __call__(self, *a, **k):
            raise NotImplementedError(f"{name} ({where}) is outside the MI355X hot path built so far")

    _P.__name__ = _P.__qualname__ = name
    return _P


_HEADS = {
    "MultiStageRefineHead": "models/dense_heads/multi_stage_refine_head.py",
    "GroupCorrectionHead": "models/roi_heads/fsd_roi_head.py",
    "FocalLoss": "mmdet loss", "L1Loss": "mmdet loss", "SmoothL1Loss": "mmdet loss", "CrossEntropyLoss": "mmdet loss",
}
for _n, _w in _HEADS.items():
    MODELS.register_module(_n, module=_module_placeholder(_n, _w))
for _n in ("ABSPointBBoxCoder",):
    BBOX_CODERS.register_module(_n, module=_plain_placeholder(_n, "core/bbox/coders"))
for _n in ("HybridAssigner", "FrustumAssigner", "PointInBoxAssigner", "DistAssigner", "MaxIoUAssigner"):
    BBOX_ASSIGNERS.register_module(_n, module=_plain_placeholder(_n, "core/bbox/assigners"))
for _n in ("MyLoadPointsFromMultiSweeps",
           "LoadAnnotations3D", "ObjectSample",
           "ObjectRangeFilter", "ObjectNameFilter", "PointShuffle",
           "MyObjectSample", "MyObjectRangeFilter", "MyGlobalRotScaleTrans", "MyRandomFlip3D", "MyPointShuffle"):
    PIPELINES.register_module(_n, module=_plain_placeholder(_n, "datasets/pipelines"))
for _n in ("NuScenesDataset", "CBGSDataset", "Argo2Dataset", "My_Resample_Dataset", "RepeatDataset"):
    DATASETS.register_module(_n, module=_plain_placeholder(_n, "datasets"))
for _n in ("DisableAugmentationHook", "EnableFSDDetectionHook", "EnableFSDDetectionHookIter"):
    HOOKS.register_module(_n, module=_plain_placeholder(_n, "core/hook/fsd_hooks.py"))
