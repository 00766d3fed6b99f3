from .voxel2point_neck import Voxel2PointScatterNeck

__all__ = ["Voxel2PointScatterNeck"]
