"""stein_amd -- MI355X-native SVGD particle-update engine behind the stein.{kernels,samplers,optimizers} API.

The arithmetic is hand-written HIP for gfx950 in libsteinhip.so (stein_amd/csrc), bound through
ctypes; PyTorch-ROCm supplies device memory, streams and torch.distributed (RCCL).  There is no
CPU implementation in this package: without the built library or without a GPU, calls raise.
"""
from .version import __version__  # noqa: F401
from . import kernels, optimizers, samplers, utilities  # noqa: F401
