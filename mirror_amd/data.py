"""On-device data feed for the pre-training step: the work of `TCGAWSIRNAPretrainDataset.__getitem__`
(datasets/dataset_pretrain.py:150-167) for a whole batch, without a host round trip.

The reference loads one slide's patch features [n_i, F] per item, draws `num_wsi_feature_tokens` row indices with
`np.random.choice(n_i, N, replace=n_i < N)` (:157-161), gathers them (:162) and looks the RNA vector up by slide id
(:164-166).  Here every slide of the split sits in ONE HBM bank ([sum n_i, F], 288 GB is room for thousands of slides in
bf16), the index draw runs on the device (a random permutation prefix when the slide is long enough, i.i.d. uniform
draws — i.e. sampling WITH replacement — when it is short: the two branches of :157) and the gather is one HIP kernel
launch per batch (mh_gather_rows, HBM bound: N * F * 2 bytes per sample).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import kernels as K
from .kernels import MirrorHipError


def sample_indices(n: int, num_tokens: int, generator: Optional[torch.Generator] = None, device="cpu") -> torch.Tensor:
    """The index draw of dataset_pretrain.py:157-161 for one slide: without replacement when n >= num_tokens (a uniformly
    random ordered subset), with replacement otherwise."""
    if n <= 0:
        raise ValueError("a slide needs at least one patch")
    if n >= num_tokens:
        return torch.randperm(n, generator=generator, device=device)[:num_tokens]
    return torch.randint(0, n, (num_tokens,), generator=generator, device=device)


class DeviceSlideBank:
    """All slides of a split resident in HBM + the RNA table; `batch(ids)` returns what a DataLoader over the reference
    dataset would collate: (wsi [B, N, F], rna [B, G] float32)."""

    def __init__(self, slides: Sequence[torch.Tensor], rna: torch.Tensor, num_wsi_feature_tokens: int, device="cuda",
                 dtype: Optional[torch.dtype] = None):
        if len(slides) == 0 or len(slides) != rna.shape[0]:
            raise ValueError("need one RNA row per slide")
        Fd = slides[0].shape[1]
        for sl in slides:
            if sl.dim() != 2 or sl.shape[1] != Fd or sl.shape[0] == 0:
                raise ValueError("every slide must be a non-empty [n_i, F] tensor with the same F")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise MirrorHipError("DeviceSlideBank lives in MI355X HBM (no CPU path); use the reference dataset on the host")
        dtype = dtype or slides[0].dtype
        self.num_tokens = int(num_wsi_feature_tokens)
        self.lengths = torch.tensor([int(sl.shape[0]) for sl in slides], dtype=torch.int64)
        self.offsets = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(self.lengths, 0)[:-1]])
        self.bank = torch.cat([sl.to(dtype) for sl in slides], dim=0).to(dev).contiguous()
        self.rna = rna.to(dev, torch.float32).contiguous()
        self.device = dev

    def __len__(self) -> int:
        return int(self.lengths.numel())

    def draw(self, ids: Sequence[int], generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """[B, N] global row indices into the bank for the slides `ids` (device tensor)."""
        rows = [sample_indices(int(self.lengths[i]), self.num_tokens, generator, self.device) + int(self.offsets[i]) for i in ids]
        return torch.stack(rows).contiguous()

    def batch(self, ids: Sequence[int], generator: Optional[torch.Generator] = None, rows: Optional[torch.Tensor] = None):
        """`rows` ([B, N] global indices, e.g. from the oracle's draw) overrides the device draw — parity tests."""
        if rows is None:
            rows = self.draw(ids, generator)
        rows = rows.to(self.device, torch.int64).contiguous()
        wsi = K.gather_rows(self.bank, rows)
        idx = torch.as_tensor(list(ids), device=self.device, dtype=torch.int64)
        return wsi, K.gather_rows(self.rna, idx)
