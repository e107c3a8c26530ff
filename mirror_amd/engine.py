"""Training step engine for the MIRROR hot path: the per-step semantics of `train_one_epoch`
(train_mirror.py:1126-1284) on MI355X, one process per GPU.

  * flat arenas: f32 master params, f32 grads, Adam m/v and a bf16 shadow live in five contiguous HBM
    buffers; module parameters are views into them.  Adam (+ the bf16 shadow refresh) is ONE kernel over
    the arena, zero-grad is one memset, and gradient all-reduce works on arena slices (no flatten copies).
  * data parallel = DDP semantics (gradient AVG across ranks, train_mirror.py:811-813): arena buckets are
    all-reduced (SUM; the 1/world factor is folded into the Adam kernel) over RCCL on a side HIP stream as soon as autograd has finished the parameters they hold,
    overlapping the WSI backward; the arena is laid out in reverse registration order so buckets complete
    roughly front to back.
  * step glue kept from the reference: prototype rows L2-normalised before every batch (:1133-1136),
    logit_scale clamped to [0, ln 100] after the update (:1254-1255) — both as kernels, no host sync.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import functional as Fn
from . import kernels as K
from .functional import POLICIES

f32, bf16 = torch.float32, torch.bfloat16
_ALIGN = 8  # elements: keeps every parameter 32-B aligned in f32 and 16-B aligned in the bf16 shadow


def plan_buckets(sizes: Sequence[int], cap_elems: int, align: int = _ALIGN):
    """Cut the arena (parameters laid out back to back, each padded to `align` elements) into all-reduce
    buckets of at least `cap_elems` elements (the last one takes the remainder).
    Returns ([[start, end, n_params], ...], owner) with owner[i] = bucket index of parameter i."""
    buckets: List[List[int]] = []
    owner: List[int] = []
    start = off = count = 0
    for i, n in enumerate(sizes):
        off += (n + align - 1) // align * align
        owner.append(len(buckets))
        count += 1
        if off - start >= cap_elems or i == len(sizes) - 1:
            buckets.append([start, off, count])
            start, count = off, 0
    return buckets, owner


# A/B switch: 1 (default) = the transposed weight copies are rebuilt at the start of a step on the RNA stream, 0 = behind Adam
_ASYNC_GATHER = True      # alignment all-gather issued behind the heads
_DEFER_SKINNY = True      # one multi-tensor launch for the skinny weight gradients
_TRANSPOSE_AT_START = True      # (test hook)
_EARLY_ADAM = True      # (test hook, round 5) the RNA encoder's share of the optimizer step on its branch's stream, beside the WSI backward
_KERNEL_D2D = True      # (test hook, round 5) the static-input refresh of a replayed step as a kernel, not a copy-engine transfer


class TrainEngine:
    def __init__(self, model: torch.nn.Module, loss_fn, *, lr: float = 2e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 precision: str = "bf16", wsi_mask_ratio: float = 0.75, rna_mask_ratio: float = 0.75,
                 bucket_mb: float = 25.0, process_group=None, graph: Optional[bool] = None,
                 clip_grad: Optional[float] = None, clip_mode: str = "norm", accum_steps: int = 1, seed: Optional[int] = None,
                 snapshot_grads: bool = False, grad_reduce_dtype: str = "f32"):
        """grad_reduce_dtype: "f32" (default: the f32 arena slices are all-reduced in place) or "bf16" (BASELINE config 5 /
        SURVEY.md §8e "bf16 grads optional": each bucket is rounded to bf16 for the wire — half the xGMI bytes — summed by
        RCCL in bf16 and widened back into the f32 arena; Adam still reads f32).
        seed: dropout (Philox) seed of this process; rank is added to it, as the reference's
        `utils.random_seed(args.seed, args.rank)` does (train_mirror.py:682).  Without it, a multi-rank engine folds its rank
        into whatever seed `Fn.manual_seed` last set, so that ranks never draw identical dropout masks.
        snapshot_grads: keep a copy of the (reduced) gradient arena of the last update in `self.grad_snap` (tests)."""
        if precision not in POLICIES:
            raise ValueError(f"unknown precision {precision!r}")
        # timm's dispatch_clip_grad modes (train_mirror.py:1219-1229, --clip-mode): "norm" (the template's default: one fused
        # kernel, the factor stays on the device) and "value" (element-wise clamp of the averaged gradient); "agc" is not built
        if clip_mode not in ("norm", "value"):
            raise NotImplementedError(f"clip_mode {clip_mode!r}: only 'norm' and 'value' are implemented (timm's 'agc' is not)")
        self.clip_mode = clip_mode
        if grad_reduce_dtype not in ("f32", "bf16"):
            raise ValueError(f"grad_reduce_dtype must be 'f32' or 'bf16', got {grad_reduce_dtype!r}")
        self.grad_reduce_dtype = grad_reduce_dtype
        self.model, self.loss_fn = model, loss_fn
        # the fp8 delayed-scaling sites are keyed by weight-shadow addresses: a new engine's shadows may land where a freed
        # engine's lived, and must not inherit its amax rings — each engine owns its site table (installed around its steps)
        self._fp8_sites: dict = {}
        self.lr, self.betas, self.eps = lr, betas, eps
        self.precision = precision
        self.wsi_mask_ratio, self.rna_mask_ratio = wsi_mask_ratio, rna_mask_ratio
        self.step_count = 0
        model.precision = precision
        self.pg = process_group
        self.world = dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(self.pg) if self.world > 1 else 0
        if seed is not None:
            Fn.manual_seed(int(seed) + self.rank)
        elif self.world > 1:
            Fn.manual_seed(Fn._dropout_state["seed"] + self.rank)
        clip = getattr(loss_fn, "clip_loss", None)
        if process_group is not None and clip is not None and hasattr(clip, "process_group"):
            clip.process_group = process_group       # gather size / label offset / reduce-scatter follow the gradient group
        # global-batch InfoNCE: the model issues the alignment all-gather itself, right behind the heads (asynchronous, on a
        # communication stream: losses.mirror_loss.prefetch_alignment_gather); the loss only awaits it
        # (armed around the engine's own model calls only: a rank-local forward elsewhere must not issue an unmatched collective)
        self._align_gather = ((clip.process_group,) if (clip is not None and getattr(clip, "gather_distributed", False) and self.world > 1
                                                        and _ASYNC_GATHER) else None)
        model._align_gather = None
        params = [p for p in model.parameters() if p.requires_grad]
        if not params or not params[0].is_cuda:
            raise Fn.K.MirrorHipError("TrainEngine needs the model on an MI355X device (model.to('cuda') first)")
        self.device = params[0].device
        # arena order: reverse registration order ~ the order in which autograd finishes gradients
        order = list(reversed(params))
        offs, total = [], 0
        for p in order:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = total
        self.master = torch.zeros(total, device=self.device, dtype=f32)
        self.grad = torch.zeros(total, device=self.device, dtype=f32)
        self.m = torch.zeros(total, device=self.device, dtype=f32)
        self.v = torch.zeros(total, device=self.device, dtype=f32)
        self.shadow = torch.zeros(total, device=self.device, dtype=bf16) if POLICIES[precision].act == bf16 else None
        self.params, self.offsets = order, offs
        # the RNA encoder's parameters (80 % of the arena at c2) are one contiguous range: their gradients are complete when the RNA
        # branch's stream has flushed its deferred weight gradients, ~2 ms before the WSI encoder's last one (see _EARLY_ADAM)
        self._early_range = None
        pname = {id(p): n for n, p in model.named_parameters()}
        idx = [i for i, p in enumerate(order) if pname.get(id(p), "").startswith("rna_encoder.")]
        if idx and idx == list(range(idx[0], idx[-1] + 1)):
            last = idx[-1]
            self._early_range = (offs[idx[0]], offs[last] + (order[last].numel() + _ALIGN - 1) // _ALIGN * _ALIGN)
        with torch.no_grad():
            for p, o in zip(order, offs):
                n = p.numel()
                self.master[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.master[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)
        # W^T shadows of the 2-D weights (skinny data-gradient kernels stream them like forward weights)
        self.shadow_t = None
        self._zero_pending = False
        self._t_stale = False
        if self.shadow is not None:
            two_d = [(p, o) for p, o in zip(order, offs) if p.dim() == 2 and p.shape[1] % 32 == 0 and p.shape[0] % 32 == 0]
            if two_d:
                self.shadow_t = torch.zeros(total, device=self.device, dtype=bf16)
                tab = []
                for p, o in two_d:
                    tab += [o, o, p.shape[0], p.shape[1]]
                self._t_table = torch.tensor(tab, device=self.device, dtype=torch.int64)
                self._t_n = len(two_d)
                self._t_max = (max(p.shape[0] for p, _ in two_d), max(p.shape[1] for p, _ in two_d))
                self._t_params = two_d
        self.sync_shadows()
        # gradient sink: Functions accumulate weight gradients straight into the arena (mirror_amd.functional._gbuf)
        self._slot_of = {p.data_ptr(): i for i, p in enumerate(order)}
        self._uses = [0] * len(order)        # sink writes per parameter per step, learnt during the first step
        self._seen = [0] * len(order)
        self._counting = True
        self._capturing = None               # list of parameter indices while graphed.GraphedBranch records a backward
        self._index_of = {id(p): i for i, p in enumerate(order)}
        self._proto = getattr(model, "prototypes", None)
        self._logit = getattr(model, "logit_scale", None)
        # step state on the device {t, 1 - b1^t, 1 - b2^t, lr}: advanced by mh_adam itself, so nothing that changes from
        # step to step is a launch argument and the whole step can be replayed as one HIP graph
        self._state = torch.tensor([0.0, 0.0, 0.0, float(lr), 1.0, 0.0], device=self.device, dtype=f32)
        # timm's --clip-grad (mode "norm") and --grad-accum-steps (train_mirror.py:1192-1230): both stay on the device
        self.clip_grad = clip_grad
        self.accum_steps = max(1, int(accum_steps))
        self._one: Optional[torch.Tensor] = None      # root gradient of loss.backward()
        self._micro = 0
        self._force = False
        self._state_lr = float(lr)
        self._snapshot_grads = bool(snapshot_grads)
        self.grad_snap = torch.empty_like(self.grad) if snapshot_grads else None
        # HIP graph of the step (~700 launches: the host needs ~10 ms to enqueue what the GPU runs in ~14 ms).  Single-GPU
        # only by default: with RCCL buckets in flight the eager path stays (MIRROR_GRAPH=1 forces, =0 disables).
        env = os.environ.get("MIRROR_GRAPH")
        self._use_graph = (self.world == 1 if graph is None else bool(graph)) if env is None else env not in ("0", "")
        if self.accum_steps > 1:
            self._use_graph = False      # micro-steps and update steps are different launch sequences
        self._graph = None
        self._graph_warm = 0
        self._zarena = Fn.ZeroArena()          # per engine; frozen once a captured step holds its address
        # eager launch mode: the launch-bound RNA branch is replayed from two HIP graphs (MIRROR_RNA_GRAPH=0 disables)
        self._rna_branch_state = "pending" if os.environ.get("MIRROR_RNA_GRAPH", "1") not in ("0", "") else "off"
        self._rna_warm = 0
        self._g_in = None
        self._g_src = self._g_ver = None      # the tensors last copied into the static inputs and their versions
        self._g_out = None
        if self.world > 1:
            dist.broadcast(self.master, src=0, group=self.pg)  # DDP's parameter broadcast at wrap time
            self.sync_shadows()
            self._build_buckets(bucket_mb)

    # ------------------------------------------------------------------ shadows
    def sync_shadows(self) -> None:
        """(Re)publish the bf16 copies after the master arena was written by anything but step()."""
        if self.shadow is None:
            return
        K.cast(self.master, bf16, out=self.shadow)
        for p, o in zip(self.params, self.offsets):
            Fn.register_shadow(p, self.shadow[o:o + p.numel()].view(p.shape))
        self._refresh_transposes()
        if self.shadow_t is not None:
            for p, o in self._t_params:
                Fn.register_shadow_t(p, self.shadow_t[o:o + p.numel()].view(p.shape[1], p.shape[0]), owner=self)

    def refresh_transposes_now(self) -> None:
        """The transposed copies trail an update until the next step starts (see _step_body); anything that differentiates
        through the model outside step() gets them rebuilt on its own stream first (functional.shadow_t calls this)."""
        self._refresh_transposes()
        if self._zero_pending and not torch.cuda.is_current_stream_capturing():
            # an out-of-step backward accumulates into p.grad = arena views: start from zero.  `_zero_pending` stays set — the
            # next step (and a step captured next) must still clear what that backward leaves behind
            self.grad.zero_()

    def _refresh_transposes(self) -> None:
        self._t_stale = False
        if self.shadow_t is not None:
            K.transpose_bf16_many(self.shadow, self.shadow_t, self._t_table, self._t_n, self._t_max[0], self._t_max[1],
                                  vec_ok=True)   # 2-D weights with both dims % 32 == 0 at offsets that are multiples of _ALIGN = 8

    # ------------------------------------------------------------------ gradient sink protocol
    def slot(self, t: torch.Tensor):
        i = self._slot_of.get(t.data_ptr())
        return None if i is None else self.params[i].grad

    def done(self, t: torch.Tensor) -> None:
        i = self._slot_of[t.data_ptr()]
        if self._capturing is not None:          # dry run that records a branch's backward graph (graphed.py)
            self._capturing.append(i)
            return
        self._seen[i] += 1
        if self._counting:
            self._uses[i] += 1
        elif self.world > 1 and self._seen[i] == self._uses[i]:
            self._on_grad(self.params[i])

    def _branch_done(self, idx) -> None:
        """A replayed backward graph has finished these parameters' gradients (they never pass done() / autograd hooks)."""
        if self.world > 1:
            for i in idx:
                self._on_grad(self.params[i])

    # ------------------------------------------------------------------ graphed RNA branch of the eager step
    def _maybe_graph_rna(self, rna: torch.Tensor) -> None:
        """Eager launch mode only (N > 1, masks, accumulation): after two warm steps record the RNA branch's forward and
        backward as HIP graphs (graphed.py) and let the model replay them.  Any failure keeps the eager branch."""
        if self._rna_branch_state != "pending" or self._rna_warm < 2:
            self._rna_warm += 1
            return
        self._rna_branch_state = "off"
        m = self.model
        if not hasattr(m, "rna_branch") or not m.training or self.precision == "fp32":
            return
        try:
            from .graphed import GraphedBranch
            noise = torch.rand(rna.shape[0], m.embed_dim, device=rna.device)
            ratio = self.rna_mask_ratio
            anchor = next((p for p in m.rna_encoder.parameters() if p.requires_grad), None)
            if anchor is None:                               # a frozen RNA encoder has no backward to replay
                return
            br = GraphedBranch(lambda x, nz: m.rna_branch(x, nz, ratio), (rna, noise), anchor, self)
            m._rna_graph = (br, ratio)
            self._rna_branch_state = "on"
        except Exception as e:                               # noqa: BLE001  (an optimisation: fall back loudly)
            import warnings
            warnings.warn(f"mirror_amd: HIP graph capture of the RNA branch failed ({e!r}); it stays eager")
            m._rna_graph = None
            self._capturing = None
            Fn.set_grad_sink(None)
            torch.cuda.synchronize()
        if self.world > 1:
            # the graphed branch reports its parameters to the buckets at another point of the backward than the eager
            # one: every rank has to run the same variant or their bucket all-reduces pair up in different orders
            ok = torch.tensor([1 if self._rna_branch_state == "on" else 0], device=rna.device, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.pg)
            if int(ok) == 0 and self._rna_branch_state == "on":
                m._rna_graph = None
                self._rna_branch_state = "off"

    # ------------------------------------------------------------------ gradient buckets (data parallel)
    def _build_buckets(self, bucket_mb: float) -> None:
        self.buckets, owner = plan_buckets([p.numel() for p in self.params], int(bucket_mb * (1 << 20) / 4))
        self._bucket_of: Dict[int, int] = {id(p): b for p, b in zip(self.params, owner)}
        self._pending = [b[2] for b in self.buckets]
        self._reported = [False] * len(self.params)
        self._works = []
        self._widen = []
        self.comm_stream = torch.cuda.Stream(device=self.device)
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def _on_grad(self, p: torch.Tensor) -> None:
        if self._capturing is not None:             # autograd-accumulated parameter inside a branch capture
            self._capturing.append(self._index_of[id(p)])
            return
        if self._micro + 1 < self.accum_steps and not self._force:   # accumulation micro-step: no reduction yet (DDP no_sync)
            return
        i = self._index_of[id(p)]
        if self._reported[i]:       # a parameter reports ONCE per step: autograd still runs the AccumulateGrad node (and its
            return                  # post hook) of a parameter whose Functions returned None because they used the sink
        self._reported[i] = True
        b = self._bucket_of[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            s, e, _ = self.buckets[b]
            # a bucket mixes parameters whose gradient kernels were queued on different streams (main, RNA side stream):
            # the reduction has to wait for all of them, not only for the stream of the parameter that completed it
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            Fn.join_side_streams(self.device, self.comm_stream)
            with torch.cuda.stream(self.comm_stream):
                self._works.append(self._reduce_slice(s, e))

    def _reduce_slice(self, s: int, e: int):
        """SUM all-reduce of arena slice [s, e) (called on the communication stream); returns the work handle."""
        if self.grad_reduce_dtype == "bf16":
            tmp = self.grad[s:e].to(bf16)
            self._widen.append((s, e, tmp))
            return dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        return dist.all_reduce(self.grad[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _finish_reduce(self) -> None:
        if self.world == 1:
            return
        if any(c != 0 for c in self._pending):  # a parameter got no gradient this step: reduce what is left
            for b, c in enumerate(self._pending):
                if c != 0:
                    s, e, _ = self.buckets[b]
                    self.comm_stream.wait_stream(torch.cuda.current_stream())
                    Fn.join_side_streams(self.device, self.comm_stream)
                    with torch.cuda.stream(self.comm_stream):
                        self._works.append(self._reduce_slice(s, e))
        for w in self._works:
            w.wait()
        if self._widen:            # bf16 wire format: widen the summed buckets back into the f32 arena (on the comm stream, in order)
            with torch.cuda.stream(self.comm_stream):
                for w, (s, e, tmp) in zip(self._works, self._widen):
                    w.wait()
                    self.grad[s:e].copy_(tmp)
            self._widen = []
        torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._works = []
        self._pending = [b[2] for b in self.buckets]
        self._reported = [False] * len(self.params)

    # ------------------------------------------------------------------ one optimizer step
    def step(self, wsi: torch.Tensor, rna: torch.Tensor, noise: Optional[dict] = None,
             wsi_key_padding_mask: Optional[torch.Tensor] = None, force_update: bool = False):
        """prototype renorm -> forward -> MIRRORLoss -> backward (+ overlapped all-reduce) -> Adam -> clamp.
        force_update: with accum_steps > 1, make this micro-step the last of its window whatever its index (the reference's
        `need_update = last_batch or (batch_idx + 1) % accum_steps == 0`, train_mirror.py:1128-1131): the gradients summed so
        far are reduced and applied instead of leaking into the next epoch's first update.
        Returns the 6 loss tensors (device scalars; nothing is synchronised here; in the eager path they are copies that
        stay valid across later steps).  Without injected `noise` the step is
        captured into a HIP graph after two eager steps and replayed from then on (the returned tensors are then the
        graph's static outputs: read them before the next step).  A batch tensor that is the same object, with the same
        version counter, as the one passed last time is taken to hold the same data (its copy into the graph's static input
        is skipped): refill a reused batch buffer with torch ops — or pass a fresh view — not through a raw data_ptr."""
        if self.lr != self._state_lr:                       # lr schedulers write engine.lr: publish it to the device state
            self._state[3:4].fill_(float(self.lr))
            self._state_lr = float(self.lr)
        self._force = bool(force_update) and self.accum_steps > 1
        mask = wsi_key_padding_mask
        if not self._use_graph or noise is not None:
            return self._step_eager(wsi, rna, noise, mask)
        if mask is not None and (mask.dtype != torch.bool or mask.device != wsi.device or tuple(mask.shape) != tuple(wsi.shape[:2])):
            return self._step_eager(wsi, rna, None, mask)       # a mask the capture cannot hold as a static input: eager
        if self._graph is not None:
            gm = self._g_in[2]
            if (wsi.shape != self._g_in[0].shape or rna.shape != self._g_in[1].shape or wsi.dtype != self._g_in[0].dtype
                    or (mask is None) != (gm is None)):
                return self._step_eager(wsi, rna, None, mask)     # a ragged last batch / a batch of the other kind runs eagerly
            # the replay reads static buffers; a batch that is the very tensor copied last time (same object, not written
            # since: a resident benchmark batch, a repeated validation batch) needs no second 134 MB copy.  BASELINE config 4: the
            # key-padding mask is one more static input of the captured step, refreshed like the batch
            for k, t in enumerate((wsi, rna) if mask is None else (wsi, rna, mask)):
                if self._g_src[k] is not t or self._g_ver[k] != t._version:
                    dst = self._g_in[k]
                    if (_KERNEL_D2D and t.dtype in (f32, bf16) and t.dtype == dst.dtype and t.is_contiguous() and t.numel() >= (1 << 20)
                            and t.numel() % 4 == 0 and t.data_ptr() % 16 == 0):
                        # a KERNEL, not the copy engine: a device-to-device hipMemcpyAsync is queued on the same in-order SDMA engine
                        # as the feeder's host -> device copy of the NEXT batch, which the host enqueued first and which waits for the
                        # previous step — the replay then started one PCIe copy (2.4 ms for 134 MB) late, every step (row f2:
                        # --feed host-bf16 ran 22-37 % behind the resident batch for three rounds)
                        K.cast(t, t.dtype, out=dst)
                    else:
                        dst.copy_(t, non_blocking=True)
                    self._g_src[k], self._g_ver[k] = t, t._version
            self._graph.replay()
            self.step_count += 1
            # the replay ran Adam: the transposed copies trail it and the arena holds this step's gradients until the next
            # step's start — the Python flags of _step_body have to say so after a replay too (out-of-step backward)
            self._t_stale = self.shadow_t is not None and _TRANSPOSE_AT_START
            self._zero_pending = self._zero_pending or _TRANSPOSE_AT_START
            return self._g_out
        if self._graph_warm < 2:                            # allocator, shadows, sink counts and lazy inits settle first
            self._graph_warm += 1
            return self._step_eager(wsi, rna, None, mask)
        self._g_in = (wsi.clone(), rna.clone(), None if mask is None else mask.clone())
        self._g_src = [wsi, rna, mask]
        self._g_ver = [wsi._version, rna._version, None if mask is None else mask._version]
        g = torch.cuda.CUDAGraph()
        count = self.step_count
        self._zarena.freeze()       # the captured memset and every carved slice hold this buffer's address from here on
        try:
            with torch.cuda.graph(g):
                self._g_out = self._step_eager(self._g_in[0], self._g_in[1], None, self._g_in[2])
        except Exception as e:                              # noqa: BLE001  (capture is an optimisation: fall back loudly)
            import warnings
            warnings.warn(f"mirror_amd: HIP graph capture of the training step failed ({e!r}); running eagerly")
            self._use_graph, self._graph, self._g_in, self._g_out = False, None, None, None
            self._g_src = self._g_ver = None
            torch.cuda.synchronize()
            return self._step_eager(wsi, rna, None, mask)
        self.step_count = count                             # capture enqueues nothing: the step runs by replay
        self._graph = g
        return self.step(wsi, rna, wsi_key_padding_mask=mask)

    def _step_eager(self, wsi: torch.Tensor, rna: torch.Tensor, noise: Optional[dict],
                    wsi_key_padding_mask: Optional[torch.Tensor] = None):
        if not torch.cuda.is_current_stream_capturing() and not self._use_graph:
            self._maybe_graph_rna(rna)       # steps that are not replayed as one graph (N > 1)
        Fn.zero_arena_begin(self.device, self._zarena)
        # the per-engine fp8 call-site table (amax rings) is installed for the duration of THIS engine's step only: another engine's
        # validate() or a bare model call must neither read nor advance these rings, and a freed engine's table must not stay
        # reachable from the module global
        sites_before = Fn._fp8_state["sites"]
        Fn.pending_lm_merge_reset("TrainEngine.step (start)")
        try:
            return self._step_body(wsi, rna, noise, wsi_key_padding_mask)
        finally:
            Fn.zero_arena_end()      # also after an exception: nothing outside a step may carve from an arena that is not re-zeroed
            Fn.fp8_delayed_scaling(None)      # delayed fp8 scaling belongs to training steps only
            Fn._fp8_state["sites"] = sites_before
            Fn.pending_lm_merge_reset("TrainEngine.step (end)")      # an aborted backward must not pin its buffers

    def _step_body(self, wsi: torch.Tensor, rna: torch.Tensor, noise: Optional[dict],
                   wsi_key_padding_mask: Optional[torch.Tensor] = None):
        Fn.dropout_step_begin(self.device)
        Fn._res_grads.clear()
        Fn.probe("step_start")
        if POLICIES[self.precision].fp8_fwd:       # delayed fp8 scaling keys its amax rings on the device-side step counter
            Fn._fp8_state["sites"] = self._fp8_sites
            Fn.fp8_delayed_scaling(self._state[0:1], self.step_count)
        def renorm_prototypes():
            if self._proto is not None:
                w = self._proto.weight
                K.rownorm_(w.data, shadow=Fn.shadow(w, POLICIES[self.precision]) if self.shadow is not None else None)
        t_done = None
        if not _TRANSPOSE_AT_START:
            renorm_prototypes()
        if _TRANSPOSE_AT_START:
            # The transposed bf16 weight copies are read by BACKWARD kernels only (data gradients of the [B, D]-row linears): instead
            # of 50 us behind Adam at the end of every step they are rebuilt at the start of the next one, on the RNA branch's
            # stream, beside the WSI forward; the backward below waits for them.
            main, side = torch.cuda.current_stream(), Fn._side_stream(self.device, 1)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                renorm_prototypes()          # read by the prototype head only, which runs on this stream (forward and backward)
                self._refresh_transposes()
                if self._zero_pending:           # the gradient arena of the update that ended the previous step
                    self.grad.zero_()
                    self._zero_pending = False
                t_done = side.record_event()
        kw = {} if wsi_key_padding_mask is None else {"wsi_key_padding_mask": wsi_key_padding_mask}
        self.model._align_gather = self._align_gather
        try:
            outs = self.model(wsi, rna, wsi_mask_ratio=self.wsi_mask_ratio, rna_mask_ratio=self.rna_mask_ratio, noise=noise, **kw)
        finally:
            self.model._align_gather = None
        losses = self.loss_fn(*outs)
        if t_done is not None:
            torch.cuda.current_stream().wait_event(t_done)
        Fn.set_grad_sink(self)
        # one process, no graphed RNA branch: the ~18 weight gradients of the [B, D]-row linears (RNA branch, heads) are queued during
        # the backward and run as ONE launch on the branch's stream behind it (with data parallelism they stay where they are: their
        # buckets should be reduced as early as possible)
        defer = _DEFER_SKINNY and self.world == 1 and self._rna_branch_state != "on" and self.shadow is not None
        # two-launch optimizer step: only where nothing couples the ranges (no gradient reduction, no accumulation window, no global
        # clipping norm) — otherwise the one launch behind the backward, as before
        early = self._early_range if (_EARLY_ADAM and defer and self.accum_steps == 1 and self.clip_grad is None
                                      and not self._force) else None
        if defer:
            Fn.skinny_wgrads_begin()
        try:
            if self._one is None or self._one.device != losses[0].device:
                self._one = torch.ones((), device=losses[0].device, dtype=losses[0].dtype)
            Fn.probe("loss_done")
            losses[0].backward(self._one)        # a persistent root gradient: no ones_like fill launch per step
            Fn.pending_lm_merge_reset("TrainEngine.step (after backward)", strict=True)
            Fn.flush_deferred_bwd()
            if defer:
                with torch.cuda.stream(Fn._side_stream(self.device, 1)):
                    Fn.probe("side_bwd_end")
                    Fn.flush_skinny_wgrads()
                    Fn.probe("flush_end")
                    if early is not None:
                        # every gradient of rna_encoder.* was written on this stream (the branch's forward ran here, so did its backward
                        # nodes and the flush above): its share of the optimizer step runs now, beside the WSI encoder's backward,
                        # instead of behind it — the step's tail keeps only the WSI / heads share of mh_adam's 28 B per parameter
                        b1, b2 = self.betas
                        lo, hi = early
                        K.adam(self.master[lo:hi], self.grad[lo:hi], self.m[lo:hi], self.v[lo:hi],
                               None if self.shadow is None else self.shadow[lo:hi], self.lr, b1, b2, self.eps, 1.0, 1.0,
                               grad_scale=1.0, dev_state=self._state, tick="early")
        finally:
            Fn._wgrad_queue = None
            Fn.set_grad_sink(None)
        Fn.join_side_streams(self.device)       # sink-written gradients of the side-stream branches (see join_side_streams)
        if self._counting:
            # first step: sink-written parameters were only counted; their buckets are reduced below (_finish_reduce
            # reduces every bucket that is still pending)
            self._counting = False
        self._seen = [0] * len(self.params)
        self._micro += 1
        if self._micro < self.accum_steps and not self._force:   # gradient accumulation: keep summing into the grad arena (the
            Fn.dropout_step_end()                    # reference's no_sync micro-steps), no reduction, no update yet
            return self._loss_out(losses)
        n_micro = self._micro            # micro-steps actually summed in this window (< accum_steps when force_update closed it)
        self._micro = 0
        self._force = False
        self._finish_reduce()
        if self._snapshot_grads:
            self.grad_snap.copy_(self.grad)
        self.step_count += 1
        b1, b2 = self.betas
        # buckets are SUM-reduced, micro-batch losses are means.  A window that force_update closed early is averaged over the
        # micro-steps it holds: the reference switches its divisor to `last_accum_steps` for the tail batches of an epoch
        # (train_mirror.py:1117-1131, :1192)
        gs = 1.0 / (self.world * n_micro)
        self.last_grad_scale = gs
        if self.clip_grad is not None and self.clip_mode == "value":
            lim = float(self.clip_grad) / gs          # clamp(gs * g, -c, c) == gs * clamp(g, -c / gs, c / gs): Adam applies gs
            self.grad.clamp_(-lim, lim)
        elif self.clip_grad is not None:
            K.grad_clip(self.grad, gs, float(self.clip_grad), self._state)
        Fn.probe("adam_start")
        # step glue that rides on Adam's two launches instead of three of its own: the logit_scale clamp behind the update
        # (train_mirror.py:1255; master and shadow) and the dropout streams' device-side base (functional.dropout_step_end)
        clamp = None
        if self._logit is not None and self._logit.requires_grad:
            clamp = (self.offsets[self._index_of[id(self._logit)]], 0.0, math.log(100.0))
        base, used = Fn.dropout_step_take()
        K.adam(self.master, self.grad, self.m, self.v, self.shadow, self.lr, b1, b2, self.eps, 1.0, 1.0,
               grad_scale=gs,                  # the DDP / accumulation average is folded into Adam
               dev_state=self._state,          # t, bias corrections, lr and the clip factor live on the device
               clamp=clamp, counter=base, counter_add=used,
               tick=early is None, hole=early)     # the RNA encoder's range was updated (and t advanced) beside the WSI backward
        if _TRANSPOSE_AT_START:
            self._t_stale = self.shadow_t is not None
        else:
            self._refresh_transposes()
        if self._logit is not None and clamp is None:          # a frozen logit_scale is not in the arena
            K.clamp_(self._logit.data.reshape(1), 0.0, math.log(100.0))
            if self.shadow is not None:
                K.cast(self._logit.data.reshape(1), bf16, out=Fn.shadow(self._logit, POLICIES[self.precision]).reshape(1))
        if _TRANSPOSE_AT_START:
            self._zero_pending = True        # cleared by the next step, beside its forward (nobody reads the arena in between)
        else:
            self.grad.zero_()
        return self._loss_out(losses)

    @staticmethod
    def _loss_out(losses):
        """The six scalars as views of ONE freshly allocated tensor: some loss terms are slices of the step's zero arena,
        which the next step's memset clears — a caller that keeps them (deferred logging) must not read zeros later.
        MirrorLossTermsFn already hands them over that way (six f32 views of its own 8-float result): no copy then."""
        ls = [x.detach() for x in losses]
        st = ls[0].untyped_storage()
        if (all(x.dtype == f32 and x.dim() == 0 and x.untyped_storage().data_ptr() == st.data_ptr() for x in ls)
                and st.nbytes() <= 64):
            return tuple(ls)
        return tuple(torch.stack([x.reshape(()).float() for x in ls]).unbind(0))

    # ------------------------------------------------------------------ validation (train_mirror.py:1382-1526)
    LOSS_NAMES = ("loss", "alignment_loss", "wsi_retention_loss", "rna_retention_loss", "style_loss", "cluster_loss")

    def validate(self, loader: Iterable[Tuple[torch.Tensor, torch.Tensor]], noise: Optional[Sequence[dict]] = None) -> "OrderedDict[str, float]":
        """The reference's `validate()`: eval mode (dropout off, masking still on), no autograd, the six losses averaged
        over the loader weighted by batch size (utils.AverageMeter.update(loss, B)) and, under DDP, averaged over ranks
        (utils.reduce_tensor).  The running sums stay on the device: ONE host sync at the end instead of six `.item()`s
        per batch.  `noise[i]` optionally pins the random draws of batch i (parity tests).  The module's train / eval
        mode is restored on return (the reference leaves it in eval and flips it back in train_one_epoch)."""
        was_training = self.model.training
        self.model.eval()
        acc = torch.zeros(7, device=self.device, dtype=torch.float64)        # 6 weighted sums + the sample count
        sites_before = Fn._fp8_state["sites"]
        if POLICIES[self.precision].fp8_fwd:
            Fn._fp8_state["sites"] = self._fp8_sites      # this engine's call sites (no tick: the exact two-pass quantisation runs)
        try:
            with torch.no_grad():
                for i, (wsi, rna) in enumerate(loader):
                    wsi = wsi.to(self.device, non_blocking=True)
                    rna = rna.to(self.device, non_blocking=True)
                    self.model._align_gather = self._align_gather
                    try:
                        outs = self.model(wsi, rna, wsi_mask_ratio=self.wsi_mask_ratio, rna_mask_ratio=self.rna_mask_ratio,
                                          noise=None if noise is None else noise[i])
                    finally:
                        self.model._align_gather = None
                    losses = self.loss_fn(*outs)
                    b = float(wsi.shape[0])
                    acc[:6] += torch.stack([x.detach().reshape(()) for x in losses]).double() * b
                    acc[6] += b
        finally:
            self.model.train(was_training)
            Fn._fp8_state["sites"] = sites_before
        if self.world > 1:      # mean over ranks of every batch's loss == summed weighted sums / summed counts for equal batch sizes
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.pg)
        vals = acc.cpu()
        n = max(float(vals[6]), 1.0)
        return OrderedDict((k, float(vals[j]) / n) for j, k in enumerate(self.LOSS_NAMES))

    # ------------------------------------------------------------------ optimizer state (resume_checkpoint, train_mirror.py:772-780)
    def _opt_order(self) -> List[Tuple[torch.nn.Parameter, int]]:
        off = {id(p): o for p, o in zip(self.params, self.offsets)}
        return [(p, off[id(p)]) for p in self.model.parameters() if id(p) in off]

    def state_dict(self) -> dict:
        """torch.optim.Adam-shaped state (what the reference checkpoints and `resume_checkpoint` reloads): per parameter, in
        model.parameters() order, {step, exp_avg, exp_avg_sq} on the CPU."""
        t = float(self._state[0].item())
        state = {}
        for i, (p, o) in enumerate(self._opt_order()):
            n = p.numel()
            state[i] = {"step": torch.tensor(t), "exp_avg": self.m[o:o + n].view(p.shape).cpu().clone(),
                        "exp_avg_sq": self.v[o:o + n].view(p.shape).cpu().clone()}
        group = {"lr": float(self.lr), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(state)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        order = self._opt_order()
        if len(sd["state"]) not in (0, len(order)):
            raise ValueError(f"optimizer state has {len(sd['state'])} entries, the model has {len(order)} parameters")
        t = 0.0
        for i, (p, o) in enumerate(order):
            st = sd["state"].get(i)
            if st is None:
                continue
            n = p.numel()
            self.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            t = float(st["step"])
        if sd.get("param_groups"):
            self.lr = float(sd["param_groups"][0].get("lr", self.lr))
        b1, b2 = self.betas
        self._state[:4].copy_(torch.tensor([t, 1.0 - b1 ** t, 1.0 - b2 ** t, float(self.lr)]))
        self._state_lr = float(self.lr)
        self.step_count = int(t)
        self._graph, self._graph_warm = None, 0          # a captured step is still valid, but re-capture keeps this simple
        self._fp8_sites.clear()
        self.sync_shadows()

