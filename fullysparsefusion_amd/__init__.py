"""fullysparsefusion_amd — MI355X-native hot path of FullySparseFusion."""
__version__ = "0.1.0"
