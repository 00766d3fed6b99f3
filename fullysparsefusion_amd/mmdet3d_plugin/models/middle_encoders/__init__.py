from .pseudo_middle_encoder import PseudoMiddleEncoderForSpconvFSD

__all__ = ["PseudoMiddleEncoderForSpconvFSD"]
