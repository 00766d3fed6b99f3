from .simple_sparse_unet import SimpleSparseUNet
from .sir import SIR

__all__ = ["SIR", "SimpleSparseUNet"]
