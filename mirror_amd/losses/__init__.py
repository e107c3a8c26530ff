"""Drop-in for the reference's `losses` package (losses/__init__.py:1-12) for the pre-training path.

`CrossEntropySurvLoss` / `NLLSurvLoss` belong to the downstream survival trainer (out of scope);
the names are exported so `from losses import ...` keeps working, and raise on use.
"""
from .info_nce import InfoNCE
from .mirror_loss import ClipLoss, MIRRORLoss


class _OutOfScope:
    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} is a downstream (survival) loss: outside the pre-training hot path")


class CrossEntropySurvLoss(_OutOfScope):
    pass


class NLLSurvLoss(_OutOfScope):
    pass


__all__ = ["CrossEntropySurvLoss", "InfoNCE", "MIRRORLoss", "NLLSurvLoss"]
