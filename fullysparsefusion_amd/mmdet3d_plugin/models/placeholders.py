"""Config `type=` names that are OUT OF SCOPE of the hot path (unused head variants, label
assigners, losses, dataset classes and hooks) resolve to ONE generic stand-in, so that
`projects/configs/nuScenes/FSF_nuScenes_config.py` loads and the model builds; using one raises, naming what is missing.
Nothing here computes anything (no silent fallbacks)."""
import torch.nn as nn

from ..registry import BBOX_ASSIGNERS, BBOX_CODERS, DATASETS, HOOKS, MODELS, PIPELINES


def out_of_scope(name, where, module=False):
    """A class called `name` whose instances only remember their config and raise when used."""
    def fail(self, *a, **k):
        raise NotImplementedError(f"{name} ({where}) is outside the MI355X hot path built so far")

    def init(self, *args, **kwargs):
        if module:
            nn.Module.__init__(self)
        self.args, self.cfg = args, kwargs

    return type(name, (nn.Module,) if module else (object,),
                {"OUT_OF_SCOPE": True, "__init__": init, "forward" if module else "__call__": fail})


for _reg, _where, _module, _names in (
        (MODELS, "heads / losses of the training path", True,
         "MultiStageRefineHead GroupCorrectionHead FocalLoss L1Loss SmoothL1Loss CrossEntropyLoss"),
        (BBOX_CODERS, "core/bbox/coders", False, "ABSPointBBoxCoder"),
        (BBOX_ASSIGNERS, "core/bbox/assigners", False, "HybridAssigner FrustumAssigner PointInBoxAssigner DistAssigner MaxIoUAssigner"),
        (PIPELINES, "datasets/pipelines (training augmentations)", False,
         "MyLoadPointsFromMultiSweeps LoadAnnotations3D ObjectSample ObjectRangeFilter ObjectNameFilter PointShuffle MyObjectSample "
         "MyObjectRangeFilter MyGlobalRotScaleTrans MyRandomFlip3D MyPointShuffle"),
        (DATASETS, "datasets", False, "NuScenesDataset CBGSDataset Argo2Dataset My_Resample_Dataset RepeatDataset"),
        (HOOKS, "core/hook/fsd_hooks.py", False, "DisableAugmentationHook EnableFSDDetectionHook EnableFSDDetectionHookIter")):
    for _n in _names.split():
        _reg.register_module(_n, module=out_of_scope(_n, _where, _module))
