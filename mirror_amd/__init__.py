"""mirror_amd — MI355X-native (gfx950) implementation of the MIRROR pre-training hot path.

Host side mirrors the reference's Python surface (`models.mirror`, `losses.MIRRORLoss`,
`losses.InfoNCE`); all arithmetic runs in hand-written HIP kernels behind the C ABI of
`include/mirror_hip.h` (`mirror_amd/lib/libmirror_hip.so`).  There is no CPU fallback.
"""
from ._lib import MirrorHipError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"


def install_aliases() -> None:
    """Make the reference trainers' own imports resolve to this build: `import models`
    (train_mirror.py:43), `from losses import MIRRORLoss` (:889-891), `from losses import InfoNCE`
    (train_pretrain.py:43) and the sub-module paths `models.mirror`, `losses.mirror_loss`, `losses.info_nce`.
    Call it before the trainer's imports run (sitecustomize.py or the first lines of the script)."""
    import importlib
    import sys
    from . import losses, models
    # by module path: `models.mirror` the attribute is the registry function (models/__init__.py:1), not the module
    mirror_mod = importlib.import_module(__name__ + ".models.mirror")
    mirror_loss = importlib.import_module(__name__ + ".losses.mirror_loss")
    info_nce = importlib.import_module(__name__ + ".losses.info_nce")
    sys.modules["models"] = models
    sys.modules["models.mirror"] = mirror_mod
    sys.modules["losses"] = losses
    sys.modules["losses.mirror_loss"] = mirror_loss
    sys.modules["losses.info_nce"] = info_nce
