from .refine import DynamicPointROIExtractor, FullySparseBboxHead  # noqa: F401
