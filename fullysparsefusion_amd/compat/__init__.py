"""Minimal stand-ins for the parts of mmcv 1.3.9 the reference's config/registry surface relies on
(`Config.fromfile`, `Registry`, `build_from_cfg`): mmcv is not installable on the target image, and the drop-in
boundary is exactly this surface."""
from .config import Config, ConfigDict
from .registry import Registry, build_from_cfg

__all__ = ["Config", "ConfigDict", "Registry", "build_from_cfg"]
