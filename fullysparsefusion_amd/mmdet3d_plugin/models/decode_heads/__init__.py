from .segmentation_head import VoteSegHead

__all__ = ["VoteSegHead"]
