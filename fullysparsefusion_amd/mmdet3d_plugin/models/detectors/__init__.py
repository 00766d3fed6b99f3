from .fsf import FSF
from .single_stage_fsd import ClusterAssigner, SingleStageFSD, VoteSegmentor

__all__ = ["FSF", "SingleStageFSD", "VoteSegmentor", "ClusterAssigner"]
