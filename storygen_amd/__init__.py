"""storygen_amd — MI355X-native implementation of StoryGen's denoising hot path.

The product path is: Python host (this package) -> thin C ABI (include/storygen_hip.h) -> hand-written HIP
kernels for gfx950 (storygen_amd/csrc).  There is no CPU or eager-PyTorch fallback: importing
`storygen_amd.ops` without the built library raises.
"""
__version__ = "0.1.0"
