from .voxel_encoder import DynamicClusterVFE, DynamicScatterVFE, DynamicVFELayer, SIRLayer

__all__ = ["DynamicClusterVFE", "DynamicScatterVFE", "DynamicVFELayer", "SIRLayer"]
