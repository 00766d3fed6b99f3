from .voxel_encoder import DynamicScatterVFE, DynamicVFELayer, SIRLayer

__all__ = ["DynamicScatterVFE", "DynamicVFELayer", "SIRLayer"]
