"""InfoNCE on HIP kernels — host-side mirror of the reference's `losses/info_nce.py` (same kwargs and checks).

Only the `negative_keys=None` branch is functional in the reference (losses/info_nce.py:144-164): its
explicit-negatives branch builds logits/labels but never assigns `loss`, so `return loss` raises
UnboundLocalError.  That branch has no oracle; it raises NotImplementedError here.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import functional as Fn

__all__ = ["InfoNCE"]
f32 = torch.float32


class InfoNCE(nn.Module):
    def __init__(self, temperature=0.1, reduction="mean", negative_mode="unpaired", symmetric=False):
        super().__init__()
        self.temperature = temperature
        self.reduction = reduction
        self.negative_mode = negative_mode
        self.symmetric = symmetric

    def forward(self, query, positive_key, negative_keys=None):
        return self.info_nce(query, positive_key, negative_keys, temperature=self.temperature,
                             reduction=self.reduction, negative_mode=self.negative_mode, symmetric=self.symmetric)

    def info_nce(self, query, positive_key, negative_keys=None, temperature=0.1, reduction="mean",
                 negative_mode="unpaired", symmetric=False):
        # input checks: same conditions and messages as losses/info_nce.py:85-120
        if query.dim() != 2:
            raise ValueError("<query> must have 2 dimensions.")
        if positive_key.dim() != 2:
            raise ValueError("<positive_key> must have 2 dimensions.")
        if negative_keys is not None:
            if negative_mode == "unpaired" and negative_keys.dim() != 2:
                raise ValueError("<negative_keys> must have 2 dimensions if <negative_mode> == 'unpaired'.")
            if negative_mode == "paired" and negative_keys.dim() != 3:
                raise ValueError("<negative_keys> must have 3 dimensions if <negative_mode> == 'paired'.")
        if len(query) != len(positive_key):
            raise ValueError("<query> and <positive_key> must must have the same number of samples.")
        if negative_keys is not None:
            if negative_mode == "paired" and len(query) != len(negative_keys):
                raise ValueError("If negative_mode == 'paired', then <negative_keys> must have the same number of samples as <query>.")
        if query.shape[-1] != positive_key.shape[-1]:
            raise ValueError("Vectors of <query> and <positive_key> should have the same number of components.")
        if negative_keys is not None:
            if query.shape[-1] != negative_keys.shape[-1]:
                raise ValueError("Vectors of <query> and <negative_keys> should have the same number of components.")
            raise NotImplementedError("explicit negative_keys: the reference's own branch is non-functional "
                                      "(UnboundLocalError at losses/info_nce.py:166); no oracle to match")
        if reduction not in ("mean", "sum", "none"):
            raise ValueError(f"{reduction} is not a valid value for reduction")
        n = query.shape[0]
        q = Fn.L2NormRowFn.apply(query.float(), 1e-12, f32)          # F.normalize(dim=-1), losses/info_nce.py:171-172
        k = Fn.L2NormRowFn.apply(positive_key.float(), 1e-12, f32)
        coef = (1.0 / n) if reduction == "mean" else 1.0
        per_row = reduction == "none"
        half = 0.5 if symmetric else 1.0
        loss = Fn.CERowsFn.apply(Fn.MatmulNTFn.apply(q, k), None, 1.0 / temperature, 0, coef * half, per_row)
        if symmetric:
            loss = loss + Fn.CERowsFn.apply(Fn.MatmulNTFn.apply(k, q), None, 1.0 / temperature, 0, coef * half, per_row)
        return loss if per_row else loss.reshape(())

    def transpose(self, x):
        return x.transpose(-2, -1)

    def normalize(self, *xs):
        return [None if x is None else Fn.L2NormRowFn.apply(x.float(), 1e-12, f32) for x in xs]
