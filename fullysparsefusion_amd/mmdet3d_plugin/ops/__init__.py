"""Public op API, same names as projects/mmdet3d_plugin/ops/__init__.py:1-11 (the SST window helpers that no
FSF config uses are out of scope) plus the mmdet3d-fork ops the path needs."""
from .dynamic_point_pool_op import dynamic_point_pool
from .norm import NaiveSyncBatchNorm1d
from .spconv import (SparseBasicBlock, SparseConv3d, SparseConvTensor, SparseInverseConv3d, SparseSequential,
                     SubMConv3d, make_sparse_convmodule)
from .sst_ops import (build_mlp, gather_by_inverse, get_activation, get_activation_layer, get_inner_win_inds,
                      scatter_v2, unique_with_plan)
from .voxel import Voxelization

__all__ = [
    "dynamic_point_pool", "scatter_v2", "get_inner_win_inds", "build_mlp", "get_activation", "get_activation_layer", "unique_with_plan",
    "gather_by_inverse", "Voxelization", "SparseConvTensor", "SparseConv3d", "SubMConv3d", "SparseInverseConv3d",
    "SparseSequential", "SparseBasicBlock", "make_sparse_convmodule", "NaiveSyncBatchNorm1d",
]
