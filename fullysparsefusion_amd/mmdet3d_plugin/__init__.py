"""Drop-in for `projects/mmdet3d_plugin` (the package the reference's configs name as `plugin_dir`,
projects/configs/nuScenes/FSF_nuScenes_config.py:7-8): importing it populates the registries with the MI355X-native
implementations of every type on the hot path (projects/mmdet3d_plugin/__init__.py:1-18 does the same upstream)."""
from . import models, ops, registry  # noqa: F401
from . import datasets  # noqa: F401  (after models: the built pipelines replace their placeholders)
from .registry import build_detector, build_model

__all__ = ["models", "ops", "registry", "build_detector", "build_model"]
