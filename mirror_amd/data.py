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


class HostFeeder:
    """Pipelined host -> HBM hand-off of DataLoader batches: what `train_mirror.py:1138-1139` does synchronously
    (`wsi_features.to(device)`, `rna_features.to(device)` at the top of every step) moved onto a copy stream, double
    buffered, so that batch k + 1 crosses PCIe (B * N * F * 4 bytes as f32: 268 MB at c2, ~5 ms at PCIe Gen5 rates; half that
    when the loader already holds bf16) while the GPU runs step k.

        feeder = HostFeeder(loader, device, wsi_dtype=torch.bfloat16)
        for wsi, rna in feeder:          # device tensors; valid until the NEXT iteration step
            engine.step(wsi, rna)

    Three slots by default (a slot is reused when the step that read it lies two steps back: the host-side wait for it never blocks).
    Host batches are staged in pinned buffers (allocated once, sized by the first batch); the f32 -> bf16 cast, when asked for,
    happens on the device on the copy stream (a host-side cast would cost more CPU time than the copy saves).  The consumer
    stream waits on an event, never on the host."""

    def __init__(self, loader, device, wsi_dtype: Optional[torch.dtype] = None, depth: int = 3):
        self.loader, self.device, self.wsi_dtype = loader, torch.device(device), wsi_dtype
        if self.device.type != "cuda":
            raise MirrorHipError("HostFeeder feeds an MI355X device")
        self.depth = max(3, int(depth))       # slots; batches in flight ahead of the consumer = depth - 2 (see _issue: the slot that is
                                              # refilled was last read TWO steps back, so the host-side wait for it does not block)
        self.stream = torch.cuda.Stream(device=self.device)
        self._slots = []          # per slot: (pinned wsi, pinned rna, device wsi raw, device wsi, device rna, ready event, free event)
        self._k = 0               # running slot counter: NOT reset per epoch, so two consecutive batches never share a slot

    def _slot(self, k: int, wsi: torch.Tensor, rna: torch.Tensor):
        while len(self._slots) <= k:
            self._slots.append(None)
        sl = self._slots[k]
        if sl is None or sl[0].shape != wsi.shape or sl[0].dtype != wsi.dtype or sl[1].shape != rna.shape:
            pw = torch.empty(wsi.shape, dtype=wsi.dtype, pin_memory=True)
            pr = torch.empty(rna.shape, dtype=torch.float32, pin_memory=True)
            # the slot's device buffers are first written on the copy stream: allocate them there (after the copy stream has
            # caught up with the consumer's stream), so that a block the caching allocator just took back from the consumer's
            # stream with kernels still pending cannot be handed out and overwritten
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                dw_raw = torch.empty(wsi.shape, dtype=wsi.dtype, device=self.device)
                dw = dw_raw if self.wsi_dtype in (None, wsi.dtype) else torch.empty(wsi.shape, dtype=self.wsi_dtype, device=self.device)
                dr = torch.empty(rna.shape, dtype=torch.float32, device=self.device)
            sl = self._slots[k] = [pw, pr, dw_raw, dw, dr, torch.cuda.Event(), None, False]
        return sl

    def _issue(self, k: int, batch):
        wsi, rna = batch[0], batch[1]
        sl = self._slot(k, wsi, rna)
        pw, pr, dw_raw, dw, dr, ready, free, used = sl
        if used:
            ready.synchronize()                   # the slot's previous host -> device copy has left the pinned buffer (depth steps ago)
        if free is not None:
            # the step that consumed this slot's device tensors has finished: waited for on the HOST.  With three slots that step lies two
            # steps back and the wait never blocks — and the copy stream must not carry a device-side wait: on this platform ANY
            # hipStreamWaitEvent on the stream that issues the pinned -> device copy (even on an event that completed long ago) takes the
            # copy out from under the replayed step, which then runs one whole PCIe transfer late (measured, tools/exp/h2d_overlap3.py:
            # replay 7.81 ms; + copy behind a host wait 8.12; + copy behind a stream wait on the same, complete, event 9.52; behind the
            # previous replay's event 10.24 — rounds 2-4 reported 22-37 % for --feed host-bf16 / host without finding this)
            free.synchronize()
        sl[7] = True
        # a pinned batch (DataLoader(pin_memory=True), what train_mirror.py builds with --pin-mem) goes to the device as it is;
        # a pageable one is staged through the slot's pinned buffer first (one host memcpy: ~10 GB/s on one core, i.e. the
        # bottleneck for 268 MB batches — pin in the loader's workers)
        src_w = wsi if wsi.is_pinned() else pw.copy_(wsi)
        src_r = rna if (rna.is_pinned() and rna.dtype == torch.float32) else pr.copy_(rna)
        with torch.cuda.stream(self.stream):
            dw_raw.copy_(src_w, non_blocking=True)
            dr.copy_(src_r, non_blocking=True)
            if dw is not dw_raw:
                K.cast(dw_raw, self.wsi_dtype, out=dw)
                # the raw kernel wrote through a pointer: tell torch the tensor changed (TrainEngine.step's graph replay skips the
                # copy into its static input when it sees the same tensor object at the same version)
                torch.autograd.graph.increment_version(dw)
            ready.record(self.stream)
        return sl

    def __iter__(self):
        it = iter(self.loader)
        pending = []
        k = self._k
        for _ in range(self.depth - 2):
            try:
                pending.append(self._issue(k % self.depth, next(it)))
                k += 1
            except StopIteration:
                break
        while pending:
            sl = pending.pop(0)
            try:                                   # request batch k + 1 before handing batch k to the consumer
                pending.append(self._issue(k % self.depth, next(it)))
                k += 1
            except StopIteration:
                pass
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(sl[5])
            try:
                yield sl[3], sl[4]
            finally:
                # also when the consumer breaks out of its loop (the generator is closed at the yield): the slot's next
                # user must wait for everything queued on this batch, and the slot counter must keep running
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                sl[6] = ev
                self._k = k
