from . import placeholders  # noqa: F401  (registers the out-of-scope type names)
from .backbones import SIR, SimpleSparseUNet
from .decode_heads import VoteSegHead
from .dense_heads import FrustumClusterHead, FSDSeparateHead, SparseClusterHead, SparseClusterHeadV2
from .roi_heads import DynamicPointROIExtractor, FullySparseBboxHead
from .detectors import FSF, ClusterAssigner, SingleStageFSD, VoteSegmentor
from .middle_encoders import PseudoMiddleEncoderForSpconvFSD
from .necks import Voxel2PointScatterNeck
from .voxel_encoders import DynamicClusterVFE, DynamicScatterVFE, SIRLayer

__all__ = ["SIR", "SimpleSparseUNet", "VoteSegHead", "FSF", "SingleStageFSD", "VoteSegmentor", "ClusterAssigner",
           "PseudoMiddleEncoderForSpconvFSD", "Voxel2PointScatterNeck", "DynamicScatterVFE", "SIRLayer", "DynamicClusterVFE", "FrustumClusterHead", "FSDSeparateHead", "SparseClusterHead",
           "SparseClusterHeadV2", "DynamicPointROIExtractor", "FullySparseBboxHead"]
