from . import placeholders  # noqa: F401  (registers the out-of-scope type names)
from .backbones import SIR, SimpleSparseUNet
from .decode_heads import VoteSegHead
from .detectors import FSF, ClusterAssigner, SingleStageFSD, VoteSegmentor
from .middle_encoders import PseudoMiddleEncoderForSpconvFSD
from .necks import Voxel2PointScatterNeck
from .voxel_encoders import DynamicScatterVFE, SIRLayer

__all__ = ["SIR", "SimpleSparseUNet", "VoteSegHead", "FSF", "SingleStageFSD", "VoteSegmentor", "ClusterAssigner",
           "PseudoMiddleEncoderForSpconvFSD", "Voxel2PointScatterNeck", "DynamicScatterVFE", "SIRLayer"]
