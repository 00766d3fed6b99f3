from .cluster_heads import FrustumClusterHead, FSDSeparateHead, SparseClusterHead, SparseClusterHeadV2  # noqa: F401
