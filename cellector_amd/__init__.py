"""cellector_amd — MI355X (gfx950) implementation of cellector's genotype-likelihood / EM scoring path.

The product is libcellector_hip.so (hand-written HIP kernels behind the C ABI of include/cellector_ffi.h) and
the `cellector` host binary; this package is the Python-side plumbing (ctypes binding, multi-GPU driver over
torch.distributed, synthetic inputs).  There is no CPU compute path here.
"""
from .ffi import Cellector, CellectorError, load_library, assignments  # noqa: F401

__all__ = ["Cellector", "CellectorError", "load_library", "assignments"]
