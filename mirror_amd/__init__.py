"""mirror_amd — MI355X-native (gfx950) implementation of the MIRROR pre-training hot path.

Host side mirrors the reference's Python surface (`models.mirror`, `losses.MIRRORLoss`,
`losses.InfoNCE`); all arithmetic runs in hand-written HIP kernels behind the C ABI of
`include/mirror_hip.h` (`mirror_amd/lib/libmirror_hip.so`).  There is no CPU fallback.
"""
from ._lib import MirrorHipError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
