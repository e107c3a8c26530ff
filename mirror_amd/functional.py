"""torch.autograd.Function wrappers around the HIP kernels (forward AND hand-derived backward).

torch is used here for what the task calls plumbing: device memory (empty/zeros/views), the
autograd tape, streams.  Every arithmetic step is a kernel launch from kernels.py.

Precision policies
  FP32  : f32 storage, v_mfma_f32_32x32x2_f32 everywhere  -> parity mode (vs the CPU oracle)
  BF16  : bf16 GEMM operands / f32 accumulate, f32 residual stream, LN/softmax/loss maths in f32; the
          pseudo-inverse GEMMs keep f32 storage and round operands to bf16 in the LDS staging pass (what the
          reference's own fp16 autocast does to them; measured: same loss/gradient error band as f32 pinv)
  BF16_PINV32 : as BF16 with the pseudo-inverse iterations in exact f32 MFMA (conservative, ~1.6x slower step)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import os

import torch
from torch.autograd import Function

from . import kernels as K
from ._lib import ACT_NONE, ACT_RELU, MH_BF16, MH_F32

f32, bf16 = torch.float32, torch.bfloat16


@dataclass(frozen=True)
class Precision:
    name: str
    mma: int
    act: torch.dtype
    pinv_mma: int
    fp8_fwd: bool = False       # forward of the big WSI projections with e4m3 MFMA operands (BASELINE config 5)


FP32 = Precision("fp32", MH_F32, f32, MH_F32)
BF16 = Precision("bf16", MH_BF16, bf16, MH_BF16)
BF16_PINV32 = Precision("bf16_pinv32", MH_BF16, bf16, MH_F32)
FP8 = Precision("fp8", MH_BF16, bf16, MH_BF16, True)      # bf16 policy + fp8 forward projections; backward stays bf16
POLICIES = {p.name: p for p in (FP32, BF16, BF16_PINV32, FP8)}

# ------------------------------------------------------------------ bf16 shadows of f32 master weights
# Entries are keyed by (address, shape) and carry a weak reference to the tensor they were made for: an entry is valid only
# while that tensor is alive — then nothing else can own the address.  (Without it a model built after another one was freed
# could land on the same addresses and silently pick up the old model's bf16 weights.)
import weakref

_shadow_cache: dict = {}


def _alive(ref) -> bool:
    return ref is not None and ref() is not None


def shadow(w: torch.Tensor, prec: Precision) -> torch.Tensor:
    """The weight in the policy's GEMM operand dtype; bf16 copies are cached per (tensor, version)."""
    wd = w.detach()
    if wd.dtype == prec.act:
        return wd
    key = (wd.data_ptr(), tuple(wd.shape))
    man = _managed_shadows.get(key)
    if man is not None:
        if _alive(man[1]) and man[0].dtype == prec.act:
            return man[0]
        if not _alive(man[1]):
            _managed_shadows.pop(key, None)
    hit = _shadow_cache.get(key)
    if hit is not None and _alive(hit[2]) and hit[0] == w._version and hit[1].dtype == prec.act:
        return hit[1]
    s = K.cast(wd.contiguous(), prec.act)
    _shadow_cache[key] = (w._version, s, weakref.ref(w))
    return s


_managed_shadows: dict = {}
_managed_shadows_t: dict = {}
_shadow_t_cache: dict = {}


def shadow_t(w: torch.Tensor, prec: Precision) -> torch.Tensor:
    """bf16 W^T [K, N] of a 2-D master weight (skinny data-gradient path streams it like a forward weight)."""
    key = (w.data_ptr(), tuple(w.shape))
    man = _managed_shadows_t.get(key)
    if man is not None:
        if _alive(man[1]):
            owner = man[2]() if len(man) > 2 and man[2] is not None else None
            if owner is not None and getattr(owner, "_t_stale", False):
                owner.refresh_transposes_now()      # a backward pass outside TrainEngine.step(): its transposes trail the update
            return man[0]
        _managed_shadows_t.pop(key, None)
    # An optimizer that rewrites the master through raw kernels (TrainEngine: mh_adam) does not bump the version counter:
    # for a weight whose bf16 shadow it manages but whose transpose it does not (a dimension that is not a multiple of 32),
    # the cached W^T would be the first step's for ever — transpose the live shadow on every call instead (small weights).
    engine_managed = key in _managed_shadows
    hit = _shadow_t_cache.get(key)
    if not engine_managed and hit is not None and _alive(hit[2]) and hit[0] == w._version:
        return hit[1]
    t = K.transpose_bf16(shadow(w, prec).contiguous())
    if not engine_managed:
        _shadow_t_cache[key] = (w._version, t, weakref.ref(w))
    return t


def register_shadow_t(w: torch.Tensor, t: Optional[torch.Tensor], owner=None) -> None:
    """owner: the engine that keeps `t` current.  It may rebuild the transposes lazily (TrainEngine does it at the start of its
    next step); while its `_t_stale` is set, a lookup from outside one of its steps triggers `owner.refresh_transposes_now()`."""
    key = (w.data_ptr(), tuple(w.shape))
    if t is None:
        _managed_shadows_t.pop(key, None)
    else:
        _managed_shadows_t[key] = (t, weakref.ref(w), None if owner is None else weakref.ref(owner))


def register_shadow(w: torch.Tensor, s: Optional[torch.Tensor]) -> None:
    """An optimizer that maintains the bf16 copy itself (mh_adam writes master + shadow in one pass) publishes it
    here; whoever rewrites the master outside that optimizer must refresh the shadow (TrainEngine.sync_shadows).
    `w` must be the long-lived parameter object: the entry dies with it."""
    key = (w.data_ptr(), tuple(w.shape))
    if s is None:
        _managed_shadows.pop(key, None)
    else:
        _managed_shadows[key] = (s, weakref.ref(w))


# ------------------------------------------------------------------ per-step zero arena
# ~30 small / medium f32 buffers per step start as zeros (atomic targets, loss accumulators, zero-initialised gradients): inside
# a TrainEngine step they are carved from ONE buffer that is cleared by a single memset at the start of the step (30 fill
# launches less; the eager multi-GPU step is host-bound).  Outside a step `zeros` is plain torch.zeros.
# Every engine owns its arena (ZeroArena).  A captured HIP graph bakes the buffer's address into its memset and into every
# carved slice, so a buffer that a graph has seen is never handed back to the allocator: the arena is FROZEN once its engine
# captures a step that carves from it (no regrowth — a later step that needs more falls back to torch.zeros for the excess).
class ZeroArena:
    __slots__ = ("buf", "off", "demand", "peak", "frozen")

    def __init__(self):
        self.buf, self.off, self.demand, self.peak, self.frozen = None, 0, 0, 0, False

    def freeze(self) -> None:
        """Called by the owner right before it captures a graph that carves from this arena."""
        self.frozen = True


_default_arena = ZeroArena()
_active_arena: Optional[ZeroArena] = None


def zero_arena_begin(device, arena: Optional[ZeroArena] = None) -> None:
    global _active_arena
    st = arena if arena is not None else _default_arena
    want = int(st.peak * 1.25) + 4096
    if not st.frozen and st.peak > 0 and (st.buf is None or st.buf.numel() < want or st.buf.device != torch.device(device)):
        st.buf = torch.empty((want,), device=device, dtype=f32)
    if st.buf is not None:
        st.buf.zero_()
    st.off, st.demand = 0, 0
    _active_arena = st


def zero_arena_end() -> None:
    global _active_arena
    st = _active_arena
    if st is not None:
        st.peak = max(st.peak, st.demand)
    _active_arena = None


def zeros(shape, device) -> torch.Tensor:
    """f32 zeros of `shape`: a slice of the step's zero arena when one is active and has room, torch.zeros otherwise."""
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    n = 1
    for d in shape:
        n *= int(d)
    st = _active_arena
    if st is not None:
        pad = (n + 63) // 64 * 64
        st.demand += pad
        buf = st.buf
        if buf is not None and st.off + pad <= buf.numel() and buf.device == torch.device(device):
            out = buf[st.off:st.off + n].view(shape)
            st.off += pad
            return out
    return torch.zeros(shape, device=device, dtype=f32)


def _zeroed1(device):
    """One zeroed f32 from the step's arena for an entry point that would otherwise memset its scratch itself; None outside a step."""
    return zeros((1,), device) if _active_arena is not None else None


# ------------------------------------------------------------------ gradient sink (TrainEngine's flat grad arena)
# With a sink installed, weight/bias/LayerNorm gradients are accumulated straight into the arena view of the parameter
# (no zeros() + autograd "grad += new" pass per parameter) and the Function returns None for them; the sink is told
# when a parameter's gradient is complete so that the data-parallel bucket logic keeps working without autograd hooks.
_sink = None


def set_grad_sink(sink) -> None:
    global _sink
    _sink = sink


def _gbuf(param: torch.Tensor, shape):
    """(f32 buffer to ACCUMULATE into, came_from_sink)."""
    if _sink is not None:
        v = _sink.slot(param)
        if v is not None:
            return v.view(shape), True
    return zeros(shape, param.device), False


def _gbuf_n(param: torch.Tensor, shape):
    """_gbuf for a parameter that reaches its Function reshaped (tokens, positional tables): the sink's slot only when it has exactly
    the elements of `shape` (a slice of a longer table falls back to a buffer of its own, returned through autograd)."""
    if _sink is not None:
        v = _sink.slot(param)
        n = 1
        for d in shape:
            n *= int(d)
        if v is not None and v.numel() == n:
            return v.view(shape), True
    return zeros(shape, param.device), False


def _gret(param: torch.Tensor, buf: torch.Tensor, via_sink: bool):
    if via_sink:
        _sink.done(param)
        return None
    return buf


def _split_k_for(rows: int, n: int, k: int, batch: int = 1) -> int:
    """Weight-gradient GEMMs have a huge contraction (rows) and few output tiles: split K to fill 256 CUs.  A batch
    that reduces into the same dW already multiplies the workgroups (and the f32 atomics: 64 KiB per tile each)."""
    tiles = ((n + 127) // 128) * ((k + 127) // 128)
    want = max(1, 1024 // max(tiles * batch, 1))
    return int(max(1, min(want, rows // 256)))


_GEMM_SPLIT = True      # for the row split below
_CUS = 256       # MI355X compute units = workgroup slots of the one-workgroup-per-CU 256 x 256 GEMM tile


_TAIL_SKINNY = True      # (test hook)


_TAIL_ASIDE = True      # (test hook) the ragged-row launches of a captured step run on a parallel branch beside their tiled launch
import contextlib


def _tail_fork():
    """Call BEFORE a tiled launch whose few ragged rows follow as tiny launches (_tail_rows): inside a HIP-graph capture it marks the
    point from which those launches may run in parallel (an event on the capturing stream); None otherwise (eager: same stream)."""
    if not (_TAIL_ASIDE and torch.cuda.is_current_stream_capturing()):
        return None
    cur = torch.cuda.current_stream()
    if any(st == cur for st in _side_streams.values()):
        # issued from a helper stream (the chain's, the RNA / heads branch): ONE tail stream serving two originating branches would wait on
        # forks of both, every join would then pick up the other branch's work (false serialisation), and a stream whose first node joins
        # two branches is the pattern that makes hipStreamEndCapture segfault on ROCm 7.2 (models/mirror.py) — run those rows inline
        return None
    return cur.record_event()


@contextlib.contextmanager
def _tail_branch(fork, device):
    """The ragged rows of a tiled launch (16 cls rows of B x 4097) are a dependent ~5 us launch that costs the replayed step 6-9 us of
    serial time wherever it stands: ~25 of them per step.  Inside a capture they go to a helper stream forked at
    `fork` (in front of the tiled launch, which they do not depend on) and the capturing stream joins behind the tiled launch: a
    parallel branch of the graph instead of a link of the chain."""
    if fork is None:
        yield
        return
    main, side = torch.cuda.current_stream(), _side_stream(device, 2)
    side.wait_event(fork)
    with torch.cuda.stream(side):
        yield
    main.wait_stream(side)


def _tail_rows(a2, b, out2, *, bias=None, act=ACT_NONE, mma, wt=None):
    """out2 [m, N] = act(a2 [m, K] @ b [K, N] + bias) for the few ragged rows a tiled launch leaves over (16 of B x 4097).  A 16-row
    product is the [B, D]-row shape of the RNA linears: the weight-streaming kernel (mh_skinny_fwd, ~6 us) instead of a tiled GEMM
    launch (13-22 us for one MFMA row block per workgroup).  b is a weight VIEW: W^T of a forward ([N, K] row-major underneath),
    or the weight itself in a data gradient, where `wt` is its transposed bf16 shadow."""
    if _TAIL_SKINNY and mma == MH_BF16 and a2.dim() == 2:
        W = b.t() if b.stride(0) == 1 else wt
        if W is not None and W.shape[0] == b.shape[1] and K.skinny_rows_ok(a2, W, out2):
            return K.skinny_fwd(a2, W, bias, act, out2.dtype, out=out2)
    return K.gemm(a2, b, out=out2, bias=bias, act=act, mma=mma)


def _gemm_rows(a, b, *, bias=None, act=ACT_NONE, mma, out_dtype, out=None, wt=None):
    """a @ b for activations a [..., R, K] against a 2-D weight view b [K, N].  The 256 x 256-tile kernel runs one
    workgroup per CU, so a launch costs ceil(tiles / 256) rounds: the to_qkv data gradient (544 tiles) pays 3 rounds for
    2.1 rounds of work.  When the last round would be less than half full, the rows that fill whole rounds go to one
    launch and the remaining rows to a second one, which is too small for the big tile and runs on the 128 x 128 kernel
    (4x smaller tiles, 2 workgroups per CU): 2.1 rounds cost ~2.3 instead of 3."""
    if (a.dim() >= 2 and a.is_contiguous() and b.dim() == 2 and a.dtype == bf16 and b.dtype == bf16 and mma == MH_BF16
            and (out_dtype or a.dtype) in (bf16, f32) and _GEMM_SPLIT
            and (out is None or (out.dim() == a.dim() and out.stride(-1) == 1 and all(
                out.stride(i) == out.shape[i + 1] * out.stride(i + 1) for i in range(out.dim() - 2))))):
        R, Kd, N = a.numel() // a.shape[-1], a.shape[-1], b.shape[1]
        rem = R % 256
        if 0 < rem <= 64 and R > 256 and N % 256 == 0 and Kd % 64 == 0:
            # a few rows past whole tiles ([16, 4097, D] = cls + patches: 65552 rows): as a batched problem every slide pays a
            # 17th row tile for one row; flat, the 16 extra rows go to their own small launch
            a2 = a.reshape(R, Kd)
            o2 = torch.empty((R, N), device=a.device, dtype=out_dtype or a.dtype) if out is None else out.view(R, N)
            fork = _tail_fork()
            _gemm_rows(a2[:R - rem], b, bias=bias, act=act, mma=mma, out_dtype=out_dtype, out=o2[:R - rem])
            with _tail_branch(fork, a.device):
                _tail_rows(a2[R - rem:], b, o2[R - rem:], bias=bias, act=act, mma=mma, wt=wt)
            return o2.reshape(*a.shape[:-1], N) if out is None else out
        if R % 256 == 0 and N % 256 == 0 and Kd % 64 == 0:
            tn = N // 256
            tiles = (R // 256) * tn
            full = tiles // _CUS
            if full >= 1 and 0 < tiles - full * _CUS <= _CUS // 2:
                rows_main = (full * _CUS // tn) * 256
                if 0 < rows_main < R:
                    a2 = a.reshape(R, Kd)
                    o2 = torch.empty((R, N), device=a.device, dtype=out_dtype or a.dtype) if out is None else out.view(R, N)
                    K.gemm(a2[:rows_main], b, out=o2[:rows_main], bias=bias, act=act, mma=mma)
                    K.gemm(a2[rows_main:], b, out=o2[rows_main:], bias=bias, act=act, mma=mma)
                    return o2.reshape(*a.shape[:-1], N) if out is None else out
    return K.gemm(a, b, out=out, bias=bias, act=act, mma=mma, out_dtype=out_dtype)


_PAD_SKIP = True      # to_qkv / its data gradient skip the front-pad rows


def _rows_window(a3, b2, out3, r0, R, *, mma, wt=None):
    """out3[:, r0:r0 + R] = a3[:, r0:r0 + R] @ b2 over the B * R real rows as one flat problem (K.gemm_rows_window) + the few rows
    past whole 256-row tiles through the weight-streaming kernel.  Rows outside the window are NOT written."""
    Bn = a3.shape[0]
    M = Bn * R
    tail = M % 256
    fork = _tail_fork() if tail else None
    K.gemm_rows_window(a3, b2, out3, r0, R, m_rows=M - tail)
    if tail:
        if tail > R:
            raise K.MirrorHipError("_rows_window: the ragged rows span more than one batch")
        with _tail_branch(fork, a3.device):
            _tail_rows(a3[Bn - 1, r0 + R - tail:r0 + R], b2, out3[Bn - 1, r0 + R - tail:r0 + R], mma=mma, wt=wt)


def _rows_scatter(dy, b2, dx, r0, R, *, mma, wt=None):
    """dx[:, r0:r0 + R] = dy @ b2 for a CONTIGUOUS dy [B, R, N] and a padded dx [B, T, K] (the data gradient of Nystrom's
    `to_out(out)[:, -n:]`): one flat problem over the B * R rows whose results are scattered into the row windows
    (mh_gemm_desc.c_rows_per_batch) — as a batched product every slide pays its own tile rounds (484 vs 790 TF/s)."""
    Bn = dy.shape[0]
    M = Bn * R
    tail = M % 256
    fork = _tail_fork() if tail else None
    K.gemm_rows_window(dy, b2, dx, r0, R, m_rows=M - tail, a_r0=0)
    if tail:
        with _tail_branch(fork, dy.device):
            _tail_rows(dy[Bn - 1, R - tail:R], b2, dx[Bn - 1, r0 + R - tail:r0 + R], mma=mma, wt=wt)


def _rows_window_ok(a3, out3, r0, R, N, prec) -> bool:
    return (_PAD_SKIP and r0 > 0 and prec.mma == MH_BF16 and prec.act == bf16 and not prec.fp8_fwd and (a3.shape[0] * R) % 256 <= 32
            and K.gemm_rows_window_ok(a3, out3, r0, R, N))


_GEMM_WINDOW = True      # (test hook)
_LM_FIRST = False     # (experiment hook, round 5) tile-path geometries: only the landmark rows of to_qkv in front of the pinv fork, the sequence rows'
                      # q | k | v as one launch beside the iteration: template +0.2 % +- 0.1 (20.137 vs 20.115 ms against the same base), off
_DEFER_QK = False     # (experiment hook, round 5) the sequence rows of q | k under the pinv chain with the v columns (one launch), only the
                      # landmark rows in front of the fork: measured +0.61 % +- 0.17 SLOWER (the window grows by more than the 93 us it
                      # takes out of the serial part: in-window GEMMs run on the non-persistent kernel beside the half-chip chain)


def _wt_of(w, prec, dy):
    """the transposed bf16 shadow of `w` when the rows of `dy` leave a ragged tail for _tail_rows (else None: nothing to look up)"""
    if not _TAIL_SKINNY or prec.mma != MH_BF16 or w is None or w.dim() != 2:
        return None
    rows = dy.shape[-2] if dy.dim() == 3 else dy.numel() // dy.shape[-1]
    flat = dy.numel() // dy.shape[-1]
    if not (0 < rows % 256 <= 32 or 0 < flat % 256 <= 32):
        return None
    return shadow_t(w, prec)


def _gemm_window(a3, b2, out3, *, bias=None, mma, wt=None):
    """out3[b] = a3[b] @ b2 (+ bias) for a row window a3 [B, R, K] of a larger buffer.  Nystrom's `to_out(out)[:, -n:]`
    has R = n = 4097 rows (cls + 4096 patches): 17 row tiles of the 256 x 256 kernel per slide, the 17th holding ONE row,
    and 16 x 17 x 2 = 544 workgroups = a third round on 256 CUs for 32 almost empty tiles.  The few ragged rows go to their
    own small launch instead and the rest is whole tiles (512 workgroups, two full rounds)."""
    R = a3.shape[1]
    hr = R % 256
    if a3.dim() == 3 and R > 256 and 0 < hr <= 32 and mma == MH_BF16 and _GEMM_WINDOW:
        K.gemm(a3[:, hr:], b2, out=out3[:, hr:], bias=bias, mma=mma)
        if hr == 1:      # one row per slide: [B, K] rows a batch stride apart
            _tail_rows(a3[:, 0], b2, out3[:, 0], bias=bias, mma=mma, wt=wt)
        else:
            K.gemm(a3[:, :hr], b2, out=out3[:, :hr], bias=bias, mma=mma)
    else:
        K.gemm(a3, b2, out=out3, bias=bias, mma=mma)
    return out3


# Delayed scaling for the fp8 forward (BASELINE config 5): per call site (keyed by the weight's address + the operand role) a
# 3-slot amax ring on the device; the engine publishes its device-side step counter here.  Without a counter (no engine,
# evaluation) or during a site's first two steps the exact two-pass quantisation runs.
_fp8_state = {"tick": None, "sites": {}, "host_step": 0}


_prequant: dict = {}        # data_ptr of a producer's bf16 output -> (e4m3 copy, scale) the producer wrote in the same pass


def fp8_delayed_scaling(tick: Optional[torch.Tensor], host_step: int = 0) -> None:
    """tick: device f32[1] step counter (TrainEngine._state[0:1]) for the duration of a training step; None: back to the exact
    two-pass quantisation (evaluation, module use without an engine).  The per-site amax rings survive a None."""
    _fp8_state["tick"] = tick
    _fp8_state["host_step"] = int(host_step)
    _prequant.clear()


def _quant_site(t: torch.Tensor, key):
    st = _fp8_state
    if st["tick"] is None:
        return K.quant_fp8(t)
    pq = _prequant.get(t.data_ptr())
    if pq is not None and pq[0].shape == t.shape:
        return pq                       # the LayerNorm that produced `t` already wrote its e4m3 copy (layernorm_fwd_q8)
    site = st["sites"].get(key)
    if site is None:
        site = st["sites"][key] = [torch.zeros(3, device=t.device, dtype=torch.int32), st["host_step"]]
    ring, born = site
    if st["host_step"] - born < 2:
        # the ring is still empty: exact scale now, and seed this step's slot so that the next step finds a history
        q, sc = K.quant_fp8(t)
        slot = st["host_step"] % 3
        ring[slot:slot + 1].copy_((sc * 448.0).view(torch.int32))
        return q, sc
    return K.quant_fp8_delayed(t, ring, st["tick"])


def _fp8_linear(xa, wa, bias, act, out):
    """out = act(xa @ wa^T + bias) with per-tensor-scaled e4m3 operands (config 5).  xa [.., R, K] bf16 (a row window of a
    contiguous parent is quantised through the parent), wa [N, K] bf16 contiguous.  Returns False when the shape does not
    qualify (the caller then runs the bf16 product)."""
    N = wa.shape[0]
    rows = xa.numel() // xa.shape[-1]
    if rows < 1024 or not wa.is_contiguous() or wa.numel() % 4:
        return False
    base = xa if xa.is_contiguous() else xa._base
    if base is None or not base.is_contiguous() or base.numel() % 4 or base.dtype != bf16:
        return False
    qb, sx = _quant_site(base, (wa.data_ptr(), "x"))
    xq = qb if base is xa else qb.as_strided(xa.shape, xa.stride(), xa.storage_offset() - base.storage_offset())
    if not K.gemm_fp8_ok(xq, N):
        return False
    wq, sw = _quant_site(wa, (wa.data_ptr(), "w"))
    K.gemm_fp8(xq, sx, wq, sw, out, bias=bias, act=act)
    return True


# ------------------------------------------------------------------ Linear (+ReLU)
class LinearFn(Function):
    """y = act(x @ W^T + b).  x: [R, K] or [B, R, K] (a row window of a larger buffer is fine);
    W: [N, K] f32 master.  Replaces nn.Linear (+ nn.ReLU for _fc1, models/mirror.py:346)."""

    @staticmethod
    def forward(ctx, x, w, b, act, prec, out_dtype, defer_from=None):
        wa = shadow(w, prec)
        # [B, D] activations: weight-streaming kernels, which take an f32 operand as it is (rounded to bf16 on load: no cast launch)
        ctx.skinny = prec.act == bf16 and _SKINNY_F32 and x.dtype == f32 and _skinny_ok(x, wa)
        xa = x if (x.dtype == prec.act or ctx.skinny) else K.cast(x.contiguous(), prec.act)
        bd = None if b is None else b.detach()
        ctx.skinny = ctx.skinny or (prec.act == bf16 and _skinny_ok(xa, wa))
        y = None
        if (defer_from and _DEFER_V and not ctx.skinny and not prec.fp8_fwd and b is None and act == ACT_NONE
                and xa.is_contiguous() and 0 < defer_from < wa.shape[0]):
            # to_qkv: the q | k columns now, the v columns when NystromCoreFn asks for them (under the pinv chain)
            y = torch.empty(tuple(xa.shape[:-1]) + (wa.shape[0],), device=xa.device, dtype=out_dtype or prec.act)
            c0, od = defer_from, out_dtype or prec.act
            _gemm_rows(xa, wa[:c0].t(), mma=prec.mma, out_dtype=od, out=y[..., :c0])
            _deferred[y.data_ptr()] = lambda: _gemm_rows(xa, wa[c0:].t(), mma=prec.mma, out_dtype=od, out=y[..., c0:])
        if ctx.skinny:
            y = K.skinny_fwd(xa, wa, bd, act, out_dtype or prec.act)
        elif prec.fp8_fwd:
            y = torch.empty(tuple(xa.shape[:-1]) + (wa.shape[0],), device=xa.device, dtype=out_dtype or prec.act)
            if not _fp8_linear(xa, wa, bd, act, y):
                y = None
        if y is None:
            y = _gemm_rows(xa, wa.t(), bias=bd, act=act, mma=prec.mma, out_dtype=out_dtype or prec.act)
        ctx.save_for_backward(xa, wa, y if act == ACT_RELU else None, w, b)
        ctx.act, ctx.prec, ctx.has_b, ctx.x_dtype = act, prec, b is not None, x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, wa, y, w, b = ctx.saved_tensors
        prec = ctx.prec
        if ctx.act == ACT_RELU:
            dy = K.relu_bwd(y, dy if _blk_ok(dy) else dy.contiguous(), out_dtype=prec.act)
        else:
            if not dy.is_contiguous():
                dy = dy.contiguous()
            if dy.dtype != prec.act and not (ctx.skinny and _SKINNY_F32 and dy.dtype == f32 and dy.dim() == 2):
                dy = K.cast(dy, prec.act)          # (the skinny kernels round an f32 gradient on load)
        N, Kd = wa.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            t_managed = N % 32 == 0 and Kd % 32 == 0      # the engine keeps W^T current for these (engine.py: two_d); others are transposed per call
            if ctx.skinny and (t_managed or (w.numel() <= (1 << 21) and (N % 32 == 0 or (_SKINNY_ANY and N < 1024)))):
                # (a short N that is no multiple of 32 runs the element-wise instance of the kernel on a W^T that shadow_t transposes per
                #  call: the engine keeps transposes for multiples of 32 only.  Long ones — the 3000 prototypes, the template's 1975-wide MLP
                #  — keep the split contraction below: with the per-call transpose the skinny form measured +0.34 % +- 0.22 on the c2 step;
                #  and a big weight whose transpose nobody keeps — the template's [1536, 10234] gene embedding, whose data gradient the mask
                #  token needs — is not transposed per call either: 31 MB each way, three times per step)
                dx = K.skinny_fwd(dy, shadow_t(w, prec), None, ACT_NONE, ctx.x_dtype)
            else:
                rows = dy.numel() // N
                if rows <= 128 and N >= 1024 and dy.dim() == 2 and dy.is_contiguous():
                    # few rows, long contraction (the prototype scores: [B, 3000] @ [3000, D]): Kd / 128 workgroups would walk
                    # all of N each (50 us); split the contraction instead and round once at the end
                    dx32 = zeros((rows, Kd), dy.device)
                    K.gemm(dy, wa, out=dx32, accumulate=True, split_k=max(2, min(32, N // 128)), mma=prec.mma)
                    dx = dx32 if ctx.x_dtype == f32 else K.cast(dx32, ctx.x_dtype)
                else:
                    dx = _gemm_rows(dy, wa, mma=prec.mma, out_dtype=ctx.x_dtype, wt=_wt_of(w, prec, dy))
        want_db = ctx.has_b and ctx.needs_input_grad[2]
        fused_db = False
        if ctx.needs_input_grad[1]:
            dw, sunk = _gbuf(w, (N, Kd))
            if ctx.skinny:
                dbuf = sunk_b = None
                if want_db:                      # the bias gradient rides on the weight-gradient launch
                    dbuf, sunk_b = _gbuf(b, (N,))
                    fused_db = True
                if _wgrad_queue is not None and sunk and (dbuf is None or sunk_b) and dy.dim() == 2:
                    # engine step: every [B, D]-row weight gradient of the backward goes into ONE launch at its end (flush_skinny_wgrads)
                    _wgrad_queue.append((dy, xa, dw, dbuf, w, b if fused_db else None, torch.cuda.current_stream()))
                    return dx, None, None, None, None, None, None
                K.skinny_wgrad(dy, xa, dw, accumulate=True, db=dbuf)
                if fused_db:
                    db = _gret(b, dbuf, sunk_b)
            else:
                _wgrad(dy, xa, N, Kd, prec, dw)
            dw = _gret(w, dw, sunk)
        if want_db and not fused_db:
            db, sunk = _gbuf(b, (N,))
            K.colsum(dy.reshape(-1, N), db)
            db = _gret(b, db, sunk)
        return dx, dw, db, None, None, None, None


# deferred weight gradients of the skinny linears: a list while an engine step's backward runs (TrainEngine sets it), else None
_wgrad_queue = None


def skinny_wgrads_begin() -> None:
    global _wgrad_queue
    _wgrad_queue = []


def flush_skinny_wgrads() -> None:
    """One launch for everything the backward queued (mh_skinny_wgrad_many), then the parameters report their gradients done."""
    global _wgrad_queue
    q, _wgrad_queue = _wgrad_queue, None
    if not q:
        return
    cur = torch.cuda.current_stream()
    for st in {e[6] for e in q}:            # operands queued from another stream than the one this launch goes to
        if st != cur:
            cur.wait_stream(st)
    # items of one launch run concurrently and read-modify-write their destination: a weight used twice in the step (the style /
    # prototype branch runs on both modalities) goes to a later launch
    rounds = []
    for e in q:
        for r in rounds:
            if e[2].data_ptr() not in r[0]:
                break
        else:
            r = (set(), [])
            rounds.append(r)
        r[0].add(e[2].data_ptr())
        r[1].append(e[:4])
    for _, items in rounds:
        K.skinny_wgrad_many(items)
    for _, _, _, _, w, b, _ in q:
        _sink.done(w)
        if b is not None:
            _sink.done(b)


def _blk_ok(t: torch.Tensor) -> bool:
    return t.is_contiguous() or (t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == t.shape[2])


def _wgrad(dy: torch.Tensor, x: torch.Tensor, N: int, Kd: int, prec: Precision, dw: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW[N, K] += sum over every row of dy^T x (f32, split-K atomics). dy contiguous; x may be a row window."""
    if dw is None:
        dw = torch.zeros((N, Kd), device=dy.device, dtype=f32)
    if dy.dim() > 2 and x.is_contiguous():
        dy, x = dy.reshape(-1, N), x.reshape(-1, Kd)
    if dy.dim() == 2:
        K.gemm(dy.t(), x, out=dw, accumulate=True, split_k=_split_k_for(dy.shape[0], N, Kd), mma=prec.mma)
    else:  # batched window: every batch reduces into the same dW (atomics)
        K.gemm(dy.transpose(-1, -2), x, out=dw.expand(*dy.shape[:-2], N, Kd), accumulate=True,
               split_k=_split_k_for(dy.shape[-2], N, Kd, batch=dy.numel() // (dy.shape[-2] * N)), mma=prec.mma)
    return dw


def _skinny_param_grads(dy, xa, w, b, need_w, need_b):
    """(dW, db) of a [B, D]-row Linear as LinearFn.backward forms them: into the gradient sink when there is one, queued for the
    step's one weight-gradient launch when the engine collects them (both then come back as None)."""
    N, Kd = w.shape
    dw = db = None
    if need_w:
        dw, sunk = _gbuf(w, (N, Kd))
        dbuf = sunk_b = None
        if need_b:
            dbuf, sunk_b = _gbuf(b, (N,))
        if _wgrad_queue is not None and sunk and (dbuf is None or sunk_b) and dy.dim() == 2:
            _wgrad_queue.append((dy, xa, dw, dbuf, w, b if need_b else None, torch.cuda.current_stream()))
            return None, None
        K.skinny_wgrad(dy, xa, dw, accumulate=True, db=dbuf)
        if need_b:
            db = _gret(b, dbuf, sunk_b)
        dw = _gret(w, dw, sunk)
    elif need_b:
        db, sunk = _gbuf(b, (N,))
        K.colsum(dy.reshape(-1, N), db)
        db = _gret(b, db, sunk)
    return dw, db


class LinearPairFn(Function):
    """(x @ W1^T + b1, x @ W2^T + b2) for [B, D] rows: style_mu and style_logstd read the same hidden vector
    (models/mirror.py:845-857).  As ONE node x has a single consumer: its data gradient is the second launch's result with the
    first one's as the addend, where two LinearFn nodes leave autograd an add launch."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, prec, out_dtype):
        ctx.set_materialize_grads(False)
        wa1, wa2 = shadow(w1, prec), shadow(w2, prec)
        od = out_dtype or prec.act
        y1 = K.skinny_fwd(x, wa1, None if b1 is None else b1.detach(), ACT_NONE, od)
        y2 = K.skinny_fwd(x, wa2, None if b2 is None else b2.detach(), ACT_NONE, od)
        ctx.save_for_backward(x, w1, b1, w2, b2)
        ctx.prec = prec
        return y1, y2

    @staticmethod
    def backward(ctx, dy1, dy2):
        x, w1, b1, w2, b2 = ctx.saved_tensors
        prec = ctx.prec
        dx, part = None, None
        grads = []
        for k, (dy, w, b) in enumerate(((dy1, w1, b1), (dy2, w2, b2))):
            if dy is None:
                grads += [None, None]
                continue
            dy = dy.contiguous()
            if dy.dtype not in (f32, bf16) or (dy.dtype == f32 and not _SKINNY_F32):
                dy = K.cast(dy, prec.act)
            if ctx.needs_input_grad[0]:
                last = k == 1 or dy2 is None
                part = K.skinny_fwd(dy, shadow_t(w, prec), None, ACT_NONE, x.dtype if last else f32, addend=part)
                if last:
                    dx = part
            grads += list(_skinny_param_grads(dy, x, w, b, ctx.needs_input_grad[1 + 2 * k], b is not None and ctx.needs_input_grad[2 + 2 * k]))
        return (dx, *grads, None, None)


def linear_pair(x, w1, b1, w2, b2, *, prec: Precision, out_dtype=None):
    """(linear(x, w1, b1), linear(x, w2, b2)); one autograd node on the [B, D]-row kernels, two plain linears otherwise."""
    if (prec.act == bf16 and not prec.fp8_fwd and x.is_cuda and x.dim() == 2 and w1.shape == w2.shape and w1.shape[0] % 32 == 0
            and (x.dtype == bf16 or (x.dtype == f32 and _SKINNY_F32))):
        wa1, wa2 = shadow(w1, prec), shadow(w2, prec)
        if _skinny_ok(x, wa1) and _skinny_ok(x, wa2):
            return LinearPairFn.apply(x, w1, b1, w2, b2, prec, out_dtype)
    return linear(x, w1, b1, prec=prec, out_dtype=out_dtype), linear(x, w2, b2, prec=prec, out_dtype=out_dtype)


def linear(x, w, b=None, *, act=ACT_NONE, prec: Precision, out_dtype=None, defer_from=None):
    """defer_from = c: only output columns [0, c) are computed here; the rest is a pending launch that the consumer runs
    with run_deferred(y) (NystromCoreFn does, right after it has forked the pinv chain)."""
    return LinearFn.apply(x, w, b, act, prec, out_dtype, defer_from)


_DEFER_V = True      # (test hook)
_SKINNY_F32 = True      # f32 operands straight into the skinny kernels
_SKINNY_ANY = True      # (test hook, round 5) [B, D]-row linears whose K / N is no multiple of 32 on the skinny kernels' element-wise instance


def _skinny_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    return K.skinny_ok(x, w) and (_SKINNY_ANY or K.skinny_vec_ok(x, w))
_deferred: dict = {}        # data_ptr of a partly computed linear output -> the launch that completes it


def run_deferred(y: torch.Tensor) -> None:
    t = _deferred.pop(y.data_ptr(), None)
    if t is not None:
        t()


class LinearRowsFn(Function):
    """y = x[:, r0:r0+R] @ W^T + b for a [B, T, K] buffer: only a row window is consumed (Nystrom's
    `to_out(out)[:, -n:]`: rows that are sliced away are never computed).  dx is zero outside the window."""

    @staticmethod
    def forward(ctx, x, w, b, r0, R, prec, out_dtype):
        wa = shadow(w, prec)
        xv = x[:, r0:r0 + R]
        y = None
        if prec.fp8_fwd and x.dtype == bf16:
            y = torch.empty((x.shape[0], R, wa.shape[0]), device=x.device, dtype=out_dtype or prec.act)
            if not _fp8_linear(xv, wa, None if b is None else b.detach(), ACT_NONE, y):
                y = None
        if y is None:
            y = torch.empty((x.shape[0], R, wa.shape[0]), device=x.device, dtype=out_dtype or prec.act)
            _gemm_window(xv, wa.t(), y, bias=None if b is None else b.detach(), mma=prec.mma)
        ctx.save_for_backward(x, wa, w, b)
        ctx.r0, ctx.R, ctx.prec, ctx.has_b = r0, R, prec, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wa, w, b = ctx.saved_tensors
        prec, r0, R = ctx.prec, ctx.r0, ctx.R
        if not dy.is_contiguous():
            dy = dy.contiguous()
        if dy.dtype != prec.act:
            dy = K.cast(dy, prec.act)
        N, Kd = wa.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if r0:
                dx[:, :r0].zero_()
            if r0 + R < x.shape[1]:
                dx[:, r0 + R:].zero_()
            _gemm_window(dy, wa, dx[:, r0:r0 + R], mma=prec.mma, wt=_wt_of(w, prec, dy))
        if ctx.needs_input_grad[1]:
            dw, sunk = _gbuf(w, (N, Kd))
            _wgrad(dy, x[:, r0:r0 + R], N, Kd, prec, dw)
            dw = _gret(w, dw, sunk)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db, sunk = _gbuf(b, (N,))
            K.colsum(dy.reshape(-1, N), db)
            db = _gret(b, db, sunk)
        return dx, dw, db, None, None, None, None


_TO_OUT_WGRAD_IN_WINDOW = True      # (test hook, round 5) to_out's weight gradient beside the pinv chain's backward (-0.23 % +- 0.15)
_deferred_bwd: dict = {}    # data_ptr of a data gradient -> a weight-gradient launch that its consumer runs where it has idle CUs


def run_deferred_bwd(dx: torch.Tensor) -> None:
    t = _deferred_bwd.pop(dx.data_ptr(), None)
    if t is not None:
        t()


def flush_deferred_bwd() -> None:
    """Launch whatever nobody claimed (a to_out whose data gradient did not reach a NystromCoreFn.backward): never lose a gradient."""
    while _deferred_bwd:
        _deferred_bwd.popitem()[1]()


_HEAD_FOLD_FIRST = False     # (experiment hook) retention_head's bias-gradient fold in front of its two products: graph topology probe
_BIAS_IN_PRODUCER = True     # (test hook, round 5) bias gradients left by the pass that wrote dy instead of an mh_colsum launch over it


def _linear_rows_bwd(ctx_needs, x, wa, w, b, r0, R, prec, dy, dx_dtype=None, defer_wgrad=False, db_have=None, db_table=None):
    """Backward of y = x[:, r0:r0+R] @ W^T + b given dy [B, R, N] in the activation dtype: (dx over all of x's rows, dW, db).
    defer_wgrad: when the weight gradient goes straight into the gradient sink, its launch is left to whoever consumes dx
    (run_deferred_bwd(dx)): NystromCoreFn.backward issues to_out's weight gradient beside the half-chip pinv chain."""
    N, Kd = wa.shape
    dx = dw = db = None
    if ctx_needs[0]:
        dx = torch.empty(x.shape, device=x.device, dtype=dx_dtype or x.dtype)
        fork = _tail_fork()
        if dy.dim() == 3 and dy.is_contiguous() and dx.dtype == bf16 and dy.shape[1] == R and (dy.shape[0] * R) % 256 <= R and _rows_window_ok(dy, dx, r0, R, Kd, prec):
            _rows_scatter(dy, wa, dx, r0, R, mma=prec.mma, wt=_wt_of(w, prec, dy))
        else:
            _gemm_window(dy, wa, dx[:, r0:r0 + R], mma=prec.mma, wt=_wt_of(w, prec, dy))
        with _tail_branch(fork, x.device):       # the fills of the rows outside the window: beside the product, not in front of it
            if r0:
                dx[:, :r0].zero_()
            if r0 + R < x.shape[1]:
                dx[:, r0 + R:].zero_()
    if ctx_needs[1]:
        dw, sunk = _gbuf(w, (N, Kd))
        if defer_wgrad and sunk and dx is not None and _TO_OUT_WGRAD_IN_WINDOW:
            dwb = dw

            def later():
                _wgrad(dy, x[:, r0:r0 + R], N, Kd, prec, dwb)
                _gret(w, dwb, True)
            _deferred_bwd[dx.data_ptr()] = later
            dw = None
        else:
            _wgrad(dy, x[:, r0:r0 + R], N, Kd, prec, dw)
            dw = _gret(w, dw, sunk)
    if b is not None and ctx_needs[2]:
        if db_have is not None:      # (buffer, came_from_sink): whoever wrote dy already accumulated its column sums
            db = _gret(b, *db_have)
        else:
            db, sunk = _gbuf(b, (N,))
            # db_table [blocks, N] f32: per-block column sums of dy left by the pass that wrote it
            K.colsum(dy.reshape(-1, N) if db_table is None else db_table, db)
            db = _gret(b, db, sunk)
    return dx, dw, db


_DROP_COLSUM = True      # (test hook)
_DROP_IN_LN_BWD = True      # (test hook, round 5) to_out's Dropout backward + bias gradient inside the LayerNorm backward that produces its dy


class _DropSite:
    """Hand-over between ToOutDropAddFn and the LayerNorm that reads its output: the forward leaves the Dropout's stream coordinates, the
    LayerNorm's backward (whose dx is the Dropout's upstream gradient) leaves grad = (data_ptr of that dx, masked bf16 gradient, its column
    sums [N] f32) for ToOutDropAddFn.backward."""
    __slots__ = ("p", "seed", "offset", "base", "shape", "grad")

    def __init__(self):
        self.p = self.seed = self.offset = self.base = self.shape = self.grad = None

    def fill(self, p, seed, offset, base, shape):
        self.p, self.seed, self.offset, self.base, self.shape = p, seed, offset, base, tuple(shape)


# (the site travels ON the output tensor — `out._drop_site`, read by LayerNormFn.forward — not in a table keyed by its address: a stale
#  address entry could meet an unrelated tensor once the allocator recycles the block, ADVICE r4 on _pending_lm_merge)



class ToOutDropAddFn(Function):
    """resid + Dropout_p(core[:, r0:r0+R] @ W^T + b) -> f32 in ONE launch: [3P] to_out = Sequential(Linear, Dropout), the `[:, -n:]`
    slice and TransLayer's residual add (models/mirror.py:312-313) ride in the projection's epilogue (mh_gemm_epi DROPADD): the
    bf16 projection output never goes through HBM and the separate dropout + add pass is gone.  Bit-identical to
    dropout_add(resid, LinearRowsFn(core, ...)) (the Linear's result is rounded to bf16 in the epilogue, the Philox mask is the
    one mh_dropout draws for the same (seed, offset, element))."""

    @staticmethod
    def forward(ctx, resid, core, w, b, r0, R, p, prec, site=None):
        wa = shadow(w, prec)
        Bn, _, Kd = core.shape
        N = wa.shape[0]
        out = torch.empty((Bn, R, N), device=core.device, dtype=f32)
        ctx.p, ctx.seed, ctx.offset, ctx.base = p, _dropout_state["seed"], _lite_offset(), _dropout_state["base"]
        _dropout_state["offset"] = ctx.offset + out.numel()
        _tap(out.shape, p, ctx.seed, ctx.offset, True)
        M = Bn * R
        tail = K.linear_fused_tail(M)
        bd = None if b is None else b.detach()
        fork = _tail_fork() if tail else None
        K.linear_fused(core, wa, bd, out, K.epi_dropadd(resid, p, ctx.seed, ctx.offset, ctx.base), window=(r0, R), m_rows=M - tail)
        if tail:     # the last rows of the last slide: composed ops on [tail, K] (same masks: the offsets are per element)
            with _tail_branch(fork, core.device):
                yt = _tail_rows(core[Bn - 1, r0 + R - tail:r0 + R], wa.t(), torch.empty((tail, N), device=core.device, dtype=bf16), bias=bd, mma=prec.mma)
                K.dropout_lite(yt, p, ctx.seed, ctx.offset + (M - tail) * N, ctx.base, add_to=resid.view(M, N)[M - tail:],
                               out=out.view(M, N)[M - tail:])
        ctx.save_for_backward(core, wa, w, b)
        ctx.r0, ctx.R, ctx.prec, ctx.res_key = r0, R, prec, resid.data_ptr()
        ctx.site = site
        if site is not None:
            site.fill(p, ctx.seed, ctx.offset, ctx.base, out.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        core, wa, w, b = ctx.saved_tensors
        dy = dy.contiguous()
        db_done, fused_db = None, False
        site, handed = ctx.site, None
        if site is not None and site.grad is not None:
            handed, site.grad = site.grad, None
            if handed[0] != dy.data_ptr() or tuple(handed[1].shape) != tuple(dy.shape):
                handed = None       # dy is not the LayerNorm's dx alone (another consumer's gradient was summed in): the plain pass below
        if handed is not None:
            # the LayerNorm backward that produced dy already wrote the masked bf16 gradient and its column sums (mh_layernorm_bwd_drop)
            gb = handed[1]
            if b is not None and ctx.needs_input_grad[3]:
                dbuf, sunk = _gbuf(b, (dy.shape[-1],))
                dbuf.add_(handed[2])
                db_done, fused_db = _gret(b, dbuf, sunk), True
        else:
            gb = torch.empty(dy.shape, device=dy.device, dtype=ctx.prec.act)
        if handed is not None:
            pass
        elif (_DROP_COLSUM and b is not None and ctx.needs_input_grad[3] and dy.dtype == f32 and gb.dtype == bf16
                and K.dropout_lite_colsum_ok(dy.shape[-1])):
            # the same pass leaves the bias gradient (column sums of the masked bf16 gradient): no mh_colsum launch over it
            dbuf, sunk = _gbuf(b, (dy.shape[-1],))
            K.dropout_lite_colsum(dy, ctx.p, ctx.seed, ctx.offset, ctx.base, gb, dbuf)
            db_done, fused_db = _gret(b, dbuf, sunk), True
        else:
            K.dropout_lite(dy, ctx.p, ctx.seed, ctx.offset, ctx.base, out=gb)       # masked, scaled gradient of the projection output
        _res_grads[ctx.res_key] = dy         # the block's LayerNorm accumulates its dx into the residual gradient (see AddFn)
        needs = list(ctx.needs_input_grad[1:4])
        if fused_db:
            needs[2] = False
        dcore, dw, db = _linear_rows_bwd(needs, core, wa, w, b, ctx.r0, ctx.R, ctx.prec, gb, defer_wgrad=True)
        if fused_db:
            db = db_done
        return dy, dcore, dw, db, None, None, None, None, None


def to_out_dropout_add(resid, core, w, b, r0: int, R: int, p: float, training: bool, prec: Precision):
    """x + Dropout(to_out(core)[:, r0:r0+R]): fused when the shapes are on the 256 x 256-tile kernel (bf16 policy, training)."""
    if (training and p > 0.0 and prec.act == bf16 and not prec.fp8_fwd and resid.dtype == f32 and resid.is_contiguous()
            and core.dtype == bf16 and tuple(resid.shape) == (core.shape[0], R, w.shape[0]) and (core.shape[0] * R * w.shape[0]) % 8 == 0
            and K.linear_fused_ok(core, shadow(w, prec), (r0, R))):
        site = _DropSite() if (_DROP_IN_LN_BWD and b is not None) else None
        out = ToOutDropAddFn.apply(resid, core, w, b, r0, R, p, prec, site)
        if site is not None:
            out._drop_site = site       # a LayerNorm that reads this tensor may do the Dropout's backward in its own (LayerNormFn)
        return out
    y = LinearRowsFn.apply(core, w, b, r0, R, prec, prec.act)
    return dropout_add(resid, y, p, training, lite=True)


class _ColsumToken:
    """Hand-over slot from the pass that writes a Linear's output gradient to that Linear's backward: `ws` [blocks, N] f32 holds the
    per-block column sums of the gradient tensor whose data_ptr is `ptr` (mh_mse_masked_bwd(colsum_ws=...))."""
    __slots__ = ("ws", "ptr")

    def __init__(self):
        self.ws = None
        self.ptr = 0


class HeadSqErrFn(Function):
    """pred = x[:, r0:r0+R] @ W^T + b (bf16) with the masked squared error against `tgt` accumulated by the same launch
    (mh_gemm_epi SQERR): retention_head + the WSI retention MSE of MIRRORLoss.forward (models/mirror.py:698-699,
    losses/mirror_loss.py:98-103).  The accumulator (mh_mse_masked_fwd's [sum of row means, masked rows]) rides on the
    returned prediction as `pred._sq = (acc, tgt, mask)`; MaskedMSEFn / MirrorLossTermsFn use it instead of their forward
    pass when they are handed the same target and mask.  The backward is LinearRowsFn's."""

    @staticmethod
    def forward(ctx, x, w, b, r0, R, prec, tgt, mask, acc, cs_tok=None):
        wa = shadow(w, prec)
        Bn, T, Kd = x.shape
        N = wa.shape[0]
        y = torch.empty((Bn, R, N), device=x.device, dtype=bf16)
        K.linear_fused(x, wa, None if b is None else b.detach(), y,
                       K.epi_sqerr(mask, tgt, tgt.stride(0), acc, R), window=(r0, R))
        ctx.save_for_backward(x, wa, w, b)
        ctx.r0, ctx.R, ctx.prec, ctx.cs_tok = r0, R, prec, cs_tok
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wa, w, b = ctx.saved_tensors
        prec = ctx.prec
        if not dy.is_contiguous():
            dy = dy.contiguous()
        if dy.dtype != prec.act:
            dy = K.cast(dy, prec.act)
        db_table, tok = None, ctx.cs_tok
        if tok is not None and tok.ws is not None:
            # the masked-MSE backward that wrote exactly this dy left its column sums per block: that small table is folded into the bias
            # gradient instead of reading dy again (a dy that is not that tensor — a second consumer of pred — takes the plain colsum).
            # The fold stays where the colsum launch stood, BEHIND the two products: in front of them it is the one child of the MSE
            # kernel on this stream, the ragged-row fork moves behind it, and the replayed graph then starts the whole RNA / heads
            # backward ~1 ms later (device probes, profiles/r05_h_*: +1.2 % of the step for a 7 us launch)
            if tok.ptr == dy.data_ptr():
                db_table = tok.ws
            tok.ws = None
        db_have = None
        if _HEAD_FOLD_FIRST and db_table is not None and b is not None and ctx.needs_input_grad[2]:
            dbuf, sunk = _gbuf(b, (wa.shape[0],))        # the measured-slower order (see above), kept as a hook for the probe experiments
            K.colsum(db_table, dbuf)
            db_have, db_table = (dbuf, sunk), None
        dx, dw, db = _linear_rows_bwd(ctx.needs_input_grad[0:3], x, wa, w, b, ctx.r0, ctx.R, prec, dy, db_table=db_table, db_have=db_have)
        return dx, dw, db, None, None, None, None, None, None, None


def head_sqerr(x, w, b, r0: int, R: int, prec: Precision, tgt, mask):
    """LinearRowsFn(x, w, b, r0, R) whose launch also fills the masked-MSE accumulator against tgt [B, R, N] (f32 rows, possibly a
    row window of a larger buffer) when the shapes are on the 256 x 256-tile kernel."""
    N = w.shape[0]
    if (tgt is not None and mask is not None and prec.act == bf16 and not prec.fp8_fwd and x.dtype == bf16 and tgt.dtype == f32
            and tgt.dim() == 3 and tuple(tgt.shape) == (x.shape[0], R, N) and tgt.stride(2) == 1 and tgt.stride(1) == N
            and mask.dtype == f32 and mask.is_contiguous() and tuple(mask.shape) == (x.shape[0], R) and R % 256 == 0
            and torch.is_grad_enabled() and K.linear_fused_ok(x, shadow(w, prec), (r0, R))):
        acc = zeros((2,), x.device)
        tok = _ColsumToken() if (b is not None and _BIAS_IN_PRODUCER) else None
        y = HeadSqErrFn.apply(x, w, b, r0, R, prec, tgt, mask, acc, tok)
        y._sq = (acc, tgt, mask)
        y._sq_cs = tok
        return y
    return LinearRowsFn.apply(x, w, b, r0, R, prec, prec.act)


def _cs_of(pred, acc):
    """The column-sum hand-over slot of the HeadSqErrFn that produced `pred`, when `acc` is that launch's accumulator."""
    sq, tok = getattr(pred, "_sq", None), getattr(pred, "_sq_cs", None)
    return tok if (sq is not None and tok is not None and acc is not None and sq[0] is acc) else None


def _mse_bwd_cs(tok, pred, tgt, dp, D):
    """(colsum_ws or None): arm the hand-over slot for dp when the kernel can leave the sums."""
    if tok is None or not K.mse_masked_bwd_colsum_ok(pred, tgt, dp, D):
        return None
    tok.ws, tok.ptr = torch.empty((K.MSE_CS_BLOCKS, D), device=dp.device, dtype=f32), dp.data_ptr()
    return tok.ws


def _sq_of(pred, tgt, mask):
    """The accumulator a HeadSqErrFn launch left for exactly this (prediction, target, mask) triple, or None."""
    sq = getattr(pred, "_sq", None)
    if sq is None:
        return None
    acc, t0, m0 = sq
    if (t0.data_ptr() == tgt.data_ptr() and tuple(t0.shape) == tuple(tgt.shape) and t0.stride() == tgt.stride()
            and m0.data_ptr() == mask.data_ptr() and tuple(m0.shape) == tuple(mask.shape)):
        return acc
    return None


class EmbedMaskPosFn(Function):
    """(mask ? mask_token : h @ W^T + b) + pos -> f32 residual stream in ONE launch: retention_embed, random_masking's token select
    and `+ retention_gene_embed` (models/mirror.py:636-643, :691-693) as the projection's epilogue (mh_gemm_epi MASKPOS).
    h [B, T, K] bf16; mask [B, T - first] (1 = masked); token [D]; pos [T, D]."""

    @staticmethod
    def forward(ctx, h32, h, w, b, mask, token, pos, first, prec):
        """h32: the tensor autograd tracks (its gradient is returned in f32); h: its bf16 copy, the operand."""
        wa = shadow(w, prec)
        Bn, T, Kd = h.shape
        N = wa.shape[0]
        out = torch.empty((Bn, T, N), device=h.device, dtype=f32)
        M = Bn * T
        tail = K.linear_fused_tail(M)
        bd = None if b is None else b.detach()
        tok, ps = token.detach().reshape(-1).contiguous(), pos.detach().reshape(T, N).contiguous()
        fork = _tail_fork() if tail else None
        K.linear_fused(h, wa, bd, out, K.epi_maskpos(mask, tok, ps, T, first), m_rows=M - tail)
        if tail and T - tail >= first:       # the last rows of the last slide through the composed ops
            with _tail_branch(fork, h.device):
                rt = _tail_rows(h[Bn - 1, T - tail:], wa.t(), torch.empty((tail, N), device=h.device, dtype=bf16), bias=bd, mma=prec.mma)
                K.mask_apply_fwd(rt.view(1, tail, N), mask[Bn - 1, T - tail - first:].contiguous(), tok, ps[T - tail:].reshape(-1), 1, tail, N, 0,
                                 False, out=out[Bn - 1:, T - tail:])
        elif tail:
            raise K.MirrorHipError("EmbedMaskPosFn: the tail rows cross the unmasked prefix")
        ctx.save_for_backward(h, wa, w, b, mask)
        ctx.geom = (Bn, T, N, first, token.shape, pos.shape, prec)
        ctx.params = (token, pos)
        return out

    @staticmethod
    def backward(ctx, dy):
        h, wa, w, b, mask = ctx.saved_tensors
        Bn, T, N, first, tshape, pshape, prec = ctx.geom
        dy = dy.contiguous()
        dr = torch.empty(dy.shape, device=dy.device, dtype=prec.act)          # gradient of the projection output (zero at masked rows)
        token, pos = ctx.params
        dtok, s_tok = _gbuf_n(token, (N,))          # the kernel accumulates into both: straight into the gradient arena
        dpos, s_pos = _gbuf_n(pos, (T * N,))
        # the bias gradient of the projection = column sums of dr: the same pass leaves them (round 5; no mh_colsum launch over dr)
        db_in = None
        if b is not None and ctx.needs_input_grad[3] and _BIAS_IN_PRODUCER and K.mask_apply_bwd_dbias_ok(dy, dr, dpos, N):
            db_in = _gbuf(b, (N,))
        K.mask_apply_bwd(dy, mask, dtok, dpos, Bn, T, N, first, False, out=dr, dbias=None if db_in is None else db_in[0])
        dtok, dpos = _gret(token, dtok, s_tok), _gret(pos, dpos, s_pos)
        # f32 data gradient: EncFanoutFn sums it with the target / cls gradients in one pass (mh_fanout_bwd reads f32)
        needs = (ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        dh, dw, db = _linear_rows_bwd(needs, h, wa, w, b, 0, T, prec, dr, dx_dtype=f32, db_have=db_in)
        return (dh, None, dw, db, None, None if dtok is None else dtok.reshape(tshape), None if dpos is None else dpos.reshape(pshape),
                None, None)


def embed_mask_pos(h, w, b, mask, token, pos, first: int, prec: Precision):
    """MaskApplyFn(linear(h, w, b), mask, token, pos, first): one launch when h has a bf16 copy (layer_norm(bf16_copy=True)) or is
    bf16 itself and the shapes are on the 256 x 256-tile kernel; the composed ops otherwise."""
    hb = h if h.dtype == bf16 else getattr(h, "_bf16", None)
    if (hb is not None and prec.act == bf16 and not prec.fp8_fwd and hb.is_contiguous() and tuple(hb.shape) == tuple(h.shape)
            and mask.dtype == f32 and mask.is_contiguous() and hb.shape[1] >= 256 and K.linear_fused_ok(hb, shadow(w, prec))
            and token.numel() == w.shape[0] and pos.numel() == h.shape[1] * w.shape[0]):
        return EmbedMaskPosFn.apply(h, hb, w, b, mask, token, pos, first, prec)
    r = linear(h, w, b, prec=prec)
    return MaskApplyFn.apply(r, mask, token, pos, first, False)


# ------------------------------------------------------------------ LayerNorm
class LayerNormFn(Function):
    """LayerNorm over the last dim of x [B, T, D] (f32 residual stream) using only the first `rows` rows
    of every batch (models/mirror.py:677-679) and writing them behind `pad` zero rows (the front padding of
    [3P] NystromAttention).  Output [B, pad + rows, D] in `out_dtype`."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, rows, pad, out_dtype, q8_key=None, dual=None, fan_slot=None):
        x = x.contiguous()
        Bn, T, D = x.shape
        y = torch.empty((Bn, pad + rows, D), device=x.device, dtype=out_dtype)
        ctx.fan_slot = fan_slot
        # x = resid + Dropout(to_out(.)) of the TransLayer in front: this norm's dx is that Dropout's upstream gradient (see backward)
        ctx.drop_site = getattr(x, "_drop_site", None) if (_DROP_IN_LN_BWD and pad == 0 and x.dtype == f32) else None
        if ctx.drop_site is not None and ctx.drop_site.shape != (Bn, T, D):
            ctx.drop_site = None
        if dual is not None:
            # f32 output + its bf16 copy in one pass (mh_layernorm_fwd_dual); the copy is handed over through `dual` (a
            # one-element list): it carries no gradient of its own, its consumers' gradients arrive through the f32 output
            mean = torch.empty((Bn * rows,), device=x.device, dtype=f32)
            rstd = torch.empty_like(mean)
            y16 = torch.empty((Bn, rows, D), device=x.device, dtype=bf16)
            K.layernorm_fwd_dual(x, gamma.detach(), beta.detach(), y, y16, mean, rstd, Bn, rows, D, T * D, rows * D, eps)
            dual.append(y16)
            ctx.save_for_backward(x, gamma, mean, rstd, beta)
            ctx.rows, ctx.pad = rows, pad
            return y
        if pad:
            y[:, :pad].zero_()
        mean = torch.empty((Bn * rows,), device=x.device, dtype=f32)
        rstd = torch.empty_like(mean)
        # fp8 forward policy: once the consumer's call site has a scale history (two exact steps), this launch also writes the
        # e4m3 copy the projection reads, with that site's delayed scale — no quantisation pass over the LayerNorm output
        st = _fp8_state
        site = st["sites"].get(q8_key) if (q8_key is not None and st["tick"] is not None) else None
        if (site is not None and st["host_step"] - site[1] >= 2 and x.dtype == f32 and out_dtype == bf16 and D % 4 == 0 and D <= 2048
                and _LN_Q8):
            q = torch.empty((Bn, pad + rows, D), device=x.device, dtype=torch.uint8)
            if pad:
                q[:, :pad].zero_()
            sc = K.layernorm_fwd_q8(x, gamma.detach(), beta.detach(), y[:, pad:], mean, rstd, Bn, rows, D, T * D, (pad + rows) * D, eps,
                                    q[:, pad:], site[0], st["tick"])
            _prequant[y.data_ptr()] = (q, sc)
        else:
            K.layernorm_fwd(x, gamma.detach(), beta.detach(), y[:, pad:], mean, rstd, Bn, rows, D, T * D, (pad + rows) * D, eps)
        ctx.save_for_backward(x, gamma, mean, rstd, beta)
        ctx.rows, ctx.pad = rows, pad
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd, beta = ctx.saved_tensors
        Bn, T, D = x.shape
        rows, pad = ctx.rows, ctx.pad
        if not dy.is_contiguous():
            dy = dy.contiguous()
        dg, sunk_g = _gbuf(gamma, (D,))
        db, sunk_b = _gbuf(beta, (D,))
        # this norm's output fans out (EncFanoutFn): the retention target's and the cls row's gradients were left in the slot instead of
        # being summed into a [B, T, D] tensor — they are additive to whatever arrives here as dy, and the launch below adds them as it
        # reads dy (mh_layernorm_bwd_fan), or mh_fanout_bwd materialises the sum when the shapes are off that form
        fan = None
        if ctx.fan_slot is not None and ctx.fan_slot.extra is not None:
            fan, ctx.fan_slot.extra = ctx.fan_slot.extra, None
            if pad or not K.layernorm_bwd_fan_ok(dy, x, x, fan[0], fan[2], Bn, rows, D):     # (rows < T is fine: x_bs carries the stride)
                dy, fan = K.fanout_bwd(dy.float(), fan[0], fan[1], fan[2], Bn, rows, D), None
        # pre-norm residual block  x + f(LN(x)):  the residual add's backward ran first and left its gradient for x in
        # _res_grads; LN adds its own dx INTO that tensor (mh_layernorm_bwd accumulate_dx) instead of handing autograd a
        # second [B, T, D] f32 gradient to sum (a 400 MB elementwise pass per block)
        G = _res_grads.pop(x.data_ptr(), None)
        if G is not None and G.numel() == x.numel() and G.dtype == x.dtype and G.is_contiguous():
            G = G.view(x.shape)          # the RNA blocks run on [B, D]: layer_norm() added a leading 1
            K.layernorm_bwd(dy[:, pad:], x, gamma.detach(), mean, rstd, G, dg, db, Bn, rows, D, T * D, (pad + rows) * D,
                            accumulate_dx=True, fan=fan)
            return None, _gret(gamma, dg, sunk_g), _gret(beta, db, sunk_b), None, None, None, None, None, None, None
        dx = torch.empty_like(x)
        if rows < T:
            dx[:, rows:].zero_()         # only the rows the norm never read (a zeros_like of [B, T, D] is a 268 MB fill at config 4)
        site, drop = ctx.drop_site, None
        if site is not None:
            gbd = torch.empty((Bn, T, D), device=x.device, dtype=bf16)
            cs = zeros((D,), x.device)
            if K.layernorm_bwd_drop_ok(dy, x, dx, gbd, cs, Bn, rows, D, site.offset) and (fan is None or dy.dtype == f32):
                if rows < T:
                    gbd[:, rows:].zero_()        # rows the norm never read: their gradient (and its Dropout backward) is zero
                drop = (gbd, site.p, site.seed, site.offset, site.base, cs)
                site.grad = (dx.data_ptr(), gbd, cs)
        K.layernorm_bwd(dy[:, pad:], x, gamma.detach(), mean, rstd, dx, dg, db, Bn, rows, D, T * D, (pad + rows) * D, fan=fan, drop=drop)
        return dx, _gret(gamma, dg, sunk_g), _gret(beta, db, sunk_b), None, None, None, None, None, None, None


def _alias(base: torch.Tensor, offset: int, size, stride) -> torch.Tensor:
    """A tensor on `base`'s storage (offset in elements from base's first element) that autograd does NOT track as a view of it:
    the two row ranges of one buffer handed out as separate Function outputs / gradients are written by raw kernels and by
    in-place fills, which autograd forbids on the views of a multi-output node."""
    t = torch.empty(0, device=base.device, dtype=base.dtype)
    t.set_(base.untyped_storage(), base.storage_offset() + int(offset), tuple(size), tuple(stride))
    return t


class NormQkvLmFn(Function):
    """LayerNorm + to_qkv of a Nystrom layer with the landmarks as EXTRA ROWS of the same products (models/mirror.py:298, :312; [3P]
    NystromAttention: front padding, to_qkv, `q_landmarks = reduce(q, '... (n l) d -> ... n d', 'sum') / l`).

    The landmarks are means over l consecutive positions of q and k, and to_qkv is linear and bias-free: they are to_qkv(xpm)[:, :2D]
    for xpm = the group means of the LayerNorm output.  mh_layernorm_fwd_lm writes those means (bf16) right BEHIND the padded
    sequence, so the operand of to_qkv is one [B n_p + B m, D] buffer and the landmarks come out as the last B m rows of its
    [B n_p + B m, 3D] result — no projection launch of their own, no pass over the q | k columns.  The backward is symmetric:
    NystromCoreFn returns d qkv and d lm as the two row ranges of ONE gradient buffer ([dq_l | dk_l | 0] rows), so the data gradient
    (whose last rows are d xpm, added to the rows' gradient by mh_layernorm_bwd_lm) and the weight gradient (one contraction over
    all B n_p + B m rows) need no landmark launches either.  Returns (qkv [B, n_p, 3D], lm [B, m, 2D] with row stride 3D).
    The v columns of the sequence rows are a deferred launch (run_deferred(qkv), under the pinv chain); the pad rows are skipped
    by the flat row-window kernels (K.gemm_rows_ext) when the geometry allows, else every physical row is multiplied."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, rows, pad, l, w, prec, rmask=None, lscale=None):
        """rmask (f32 [B, pad + rows], BASELINE config 4): the front-padded key-padding mask — masked rows leave the norm as zero rows and
        stay out of the landmark sums (mh_layernorm_fwd_lm); lscale (f32 [B, m]) = l / valid count per group: the landmark rows leave
        as the masked MEANS ([3P] `q_landmarks /= divisor`), NystromCoreFn then gets kmask with a None scale."""
        ctx.relu_slot = getattr(x, "_relu_slot", None)      # x is Fc1SeqFn's sequence (layer 1): see backward
        x = x.contiguous()
        Bn, T, D = x.shape
        n_p = pad + rows
        m = n_p // l
        P, E = Bn * n_p, Bn * m
        wa = shadow(w, prec)
        N3, c0 = wa.shape[0], 2 * wa.shape[1]
        xe = torch.empty((P + E, D), device=x.device, dtype=bf16)
        mean = torch.empty((Bn * rows,), device=x.device, dtype=f32)
        rstd = torch.empty_like(mean)
        xs = xe[:P].view(Bn, n_p, D)
        K.layernorm_fwd_lm(x, gamma.detach(), beta.detach(), xs, mean, rstd, None, Bn, rows, D, T * D, pad, l, eps, xpm_bf16=xe[P:],
                           row_mask=rmask, lm_scale=lscale)
        ctx.rmask, ctx.lscale = rmask, lscale
        qe = torch.empty((P + E, N3), device=x.device, dtype=bf16)
        qkv = _alias(qe, 0, (Bn, n_p, N3), (n_p * N3, N3, 1))
        fast = (pad > 0 and prec.mma == MH_BF16 and c0 % 256 == 0 and (N3 - c0) % 256 == 0
                and K.gemm_rows_ext_ok(Bn, n_p, pad, rows, E, D, c0, xe, qe[:, :c0]) and _rows_window_ok(xs, qkv[..., c0:], pad, rows, N3 - c0, prec))
        if fast and _DEFER_QK and E % 256 == 0 and _rows_window_ok(xs, qkv, pad, rows, N3, prec):
            # (round 5 experiment, off) only the LANDMARK rows of q | k are needed in front of the fork (sim2 and the pinv chain read nothing
            # else): a [B m, D] x [D, 2D] product on 64 workgroups; the sequence rows of q, k AND v as one launch under the chain
            K.gemm(xe[P:], wa[:c0].t(), out=qe[P:, :c0], mma=prec.mma)

            def later():      # q = k = v = 0 on the pad rows (they take part in the softmaxes as zero keys)
                fork = _tail_fork()
                _rows_window(xs, wa.t(), qkv, pad, rows, mma=prec.mma)
                with _tail_branch(fork, x.device):
                    qkv[:, :pad].zero_()
        elif fast:
            fork = _tail_fork()
            tail = K.gemm_rows_ext(xe, wa[:c0].t(), qe[:, :c0], Bn, n_p, pad, rows, E)
            if tail:
                with _tail_branch(fork, x.device):
                    _tail_rows(xe[P + E - tail:], wa[:c0].t(), qe[P + E - tail:, :c0], mma=prec.mma)

            def later():      # q = k = v = 0 on the pad rows (they take part in the softmaxes as zero keys): nobody reads them before
                fork = _tail_fork()       # the attention kernels, so the fill waits with the v columns — and runs beside them
                _rows_window(xs, wa[c0:].t(), qkv[..., c0:], pad, rows, mma=prec.mma)
                with _tail_branch(fork, x.device):
                    qkv[:, :pad].zero_()
        elif _LM_FIRST and _TILE_SIDE and prec.mma == MH_BF16 and prec.pinv_mma == MH_BF16 and m != K.PINV_CHAIN_M and K.gemm_tile_ok(m, m, m):
            # the template's geometry (m = 384): the Moore-Penrose iteration is the longer side of the window it opens in NystromCoreFn and
            # reads the landmarks only — their [B m, D] x [D, 2D] product stands in front of the fork, the sequence rows' q | k | v follow as
            # ONE launch beside the iteration (run_deferred) instead of q | k (165 us) in front of it
            K.gemm(xe[P:], wa[:c0].t(), out=qe[P:, :c0], mma=prec.mma)

            def later():
                K.gemm(xe[:P], wa.t(), out=qe[:P], mma=prec.mma)           # zero pad rows in, zero rows out
        else:
            K.gemm(xe, wa[:c0].t(), out=qe[:, :c0], mma=prec.mma)      # zero pad rows in, zero rows out

            def later():
                K.gemm(xe[:P], wa[c0:].t(), out=qe[:P, c0:], mma=prec.mma)
        _deferred[qkv.data_ptr()] = later
        ctx.save_for_backward(x, gamma, mean, rstd, beta, xe, wa, w)
        ctx.geo = (rows, pad, l, fast, prec)
        return qkv, _alias(qe, P * N3, (Bn, m, c0), (m * N3, N3, 1))

    @staticmethod
    def backward(ctx, dqkv, dlm):
        x, gamma, mean, rstd, beta, xe, wa, w = ctx.saved_tensors
        rows, pad, l, fast, prec = ctx.geo
        Bn, T, D = x.shape
        n_p = pad + rows
        m = n_p // l
        P, E = Bn * n_p, Bn * m
        N3, c0 = wa.shape[0], 2 * wa.shape[1]
        pend = _pending_lm_merge.pop(dlm.data_ptr(), None)         # NystromCoreFn.backward left the landmark rows' merge to this node
        merge = None if pend is None else pend[1]
        de = ext_rows_of(dqkv, dlm, P, E, N3, c0, copy_lm=merge is None)
        if pend is not None and de.data_ptr() != pend[0]:
            raise K.MirrorHipError("NormQkvLmFn.backward: the pending landmark merge belongs to another gradient buffer (a stale entry "
                                   "whose address was recycled)")
        dxe = torch.empty((P + E, D), device=x.device, dtype=bf16)
        # data gradient: the pad rows of dxe stay unwritten on the flat path (the LayerNorm backward reads the real rows only)
        if fast:
            fork = _tail_fork()
            # the sequence rows as whole rounds of the persistent kernel (B n rows x D columns = 2.0 rounds at c2; with the landmark
            # rows appended its 24-K-tile units would start a third round for 6 % more rows: +50 us against a 19 us launch of their
            # own on the 128 x 128 kernel, measured in the step's trace), the landmark rows [dq_l | dk_l] x W[:2D] separately
            _rows_window(de[:P].view(Bn, n_p, N3), wa, dxe[:P].view(Bn, n_p, D), pad, rows, mma=prec.mma, wt=shadow_t(w, prec))
            with _tail_branch(fork, x.device):      # beside the sequence rows' product (inside a capture): ~40 us off the chain per layer
                if merge is not None:
                    merge()
                K.gemm(de[P:, :c0], wa[:c0], out=dxe[P:], mma=prec.mma)
        else:
            if merge is not None:
                merge()
            K.gemm(de, wa, out=dxe, mma=prec.mma)
        dw = None
        if ctx.needs_input_grad[7]:
            # one contraction over every physical row: the pad rows of xe are zero, the landmark rows carry [dq_l | dk_l | 0]
            dwb, sunk = _gbuf(w, (N3, D))
            _wgrad(de, xe, N3, D, prec, dwb)
            dw = _gret(w, dwb, sunk)
        dg, sunk_g = _gbuf(gamma, (D,))
        db, sunk_b = _gbuf(beta, (D,))
        dy = dxe[:P].view(Bn, n_p, D)
        gadd = dxe[P:].view(Bn, m, D)
        G = _res_grads.pop(x.data_ptr(), None)
        if G is not None and G.numel() == x.numel() and G.dtype == x.dtype and G.is_contiguous():
            # x = Fc1SeqFn's sequence [cls | relu(_fc1(wsi))] (layer 1): the rows behind the cls row leave as the ReLU-masked bf16 gradient
            # _fc1's weight gradient multiplies (Fc1SeqFn.backward picks it up: no pass over x and dx of its own)
            rslot = ctx.relu_slot if _RELU_IN_LN_BWD else None
            nrelu, fc1_b = (rslot.n, rslot.bias) if (rslot is not None and rslot.n is not None) else (None, None)
            dh = rdb = None
            if nrelu is not None and x.dtype == f32 and prec.act == bf16 and rows == T == nrelu + 1 and ctx.needs_input_grad[0]:
                dh = torch.empty((Bn, nrelu, D), device=x.device, dtype=bf16)
                if fc1_b is not None and fc1_b.requires_grad and _BIAS_IN_PRODUCER:
                    rdb = _gbuf(fc1_b, (D,))      # _fc1's bias gradient = the column sums of dh: a third partial row of this launch
                rslot.grad = (G.data_ptr(), dh, rdb)
            K.layernorm_bwd(dy[:, pad:], x, gamma.detach(), mean, rstd, G.view(x.shape), dg, db, Bn, rows, D, T * D, n_p * D,
                            accumulate_dx=True, gadd=gadd, pad=pad, l=l, relu_out=dh, relu_first=1, relu_db=None if rdb is None else rdb[0],
                            row_mask=ctx.rmask, lm_scale=ctx.lscale)
            dx = None
        else:
            dx = torch.empty_like(x)
            if rows < T:
                dx[:, rows:].zero_()         # only the rows the norm never read (a zeros_like of [B, T, D] is a 268 MB fill at config 4)
            K.layernorm_bwd(dy[:, pad:], x, gamma.detach(), mean, rstd, dx, dg, db, Bn, rows, D, T * D, n_p * D, gadd=gadd, pad=pad, l=l,
                            row_mask=ctx.rmask, lm_scale=ctx.lscale)
        return dx, _gret(gamma, dg, sunk_g), _gret(beta, db, sunk_b), None, None, None, None, dw, None, None, None


def ext_rows_alloc(Bn: int, n_p: int, m: int, N3: int, c0: int, device):
    """(de [B n_p + B m, N3] bf16, d qkv view [B, n_p, N3], d lm view [B, m, c0] with row stride N3): the gradient buffer NystromCoreFn
    fills for NormQkvLmFn.backward."""
    P, E = Bn * n_p, Bn * m
    de = torch.empty((P + E, N3), device=device, dtype=bf16)
    return de, _alias(de, 0, (Bn, n_p, N3), (n_p * N3, N3, 1)), _alias(de, P * N3, (Bn, m, c0), (m * N3, N3, 1))


def ext_rows_of(dqkv, dlm, P: int, E: int, N3: int, c0: int, copy_lm: bool = True):
    """The [P + E, N3] buffer of which dqkv / dlm are the two row ranges (ext_rows_alloc), or a freshly assembled copy."""
    if (dqkv.dtype == bf16 and dlm.dtype == bf16 and dqkv.is_contiguous() and dqkv.numel() == P * N3
            and dlm.data_ptr() == dqkv.data_ptr() + P * N3 * 2 and dlm.dim() == 3 and tuple(dlm.stride()) == (dlm.shape[1] * N3, N3, 1)
            and dlm.shape[0] * dlm.shape[1] == E and dlm.shape[2] == c0
            and dqkv.untyped_storage().nbytes() - dqkv.storage_offset() * 2 >= (P + E) * N3 * 2):
        return torch.as_strided(dqkv, (P + E, N3), (N3, 1))
    if not copy_lm:
        raise K.MirrorHipError("NormQkvLmFn.backward: a pending landmark merge needs the gradient views of ext_rows_alloc")
    de = torch.empty((P + E, N3), device=dqkv.device, dtype=bf16)
    de[:P].copy_(dqkv.reshape(P, N3))
    de[P:, :c0].copy_(dlm.reshape(E, c0))
    de[P:, c0:].zero_()
    return de


_KEYMASK_PLAN = True     # (test hook, round 5) the key-padding plan of a layer from ONE launch, shared by layers of one geometry
class KeyMask:
    """Key-padding mask of a token sequence [lead ones (cls) | src | src[:, :wrap] (square-pad rows)], src [B, n_src] bool
    (True = real patch): the `mask` argument of [3P] NystromAttention.forward as mirror_amd's TransLayer takes it (BASELINE
    config 4; the reference never passes one, models/mirror.py:312).  plan(pad, l) = (row mask of the front-padded sequence,
    landmark-group valid flag, l / (valid count + 1e-8)), all f32, from one launch (K.keymask_plan) and kept per (pad, l):
    layers of the same geometry (layer1 / layer2 of the encoder, both retention blocks) share it."""

    def __init__(self, src: torch.Tensor, lead: int = 0, wrap: int = 0):
        self.src = src if src.dtype == torch.bool else src.to(torch.bool)
        self.src = self.src.contiguous()
        self.lead, self.wrap = int(lead), int(wrap)
        self.shape = (src.shape[0], self.lead + src.shape[1] + self.wrap)
        self._plans = {}

    def dense(self) -> torch.Tensor:
        parts = ([torch.ones_like(self.src[:, :1]).expand(-1, self.lead)] if self.lead else []) + [self.src] + (
            [self.src[:, :self.wrap]] if self.wrap else [])
        return torch.cat(parts, dim=1) if len(parts) > 1 else self.src

    def plan(self, pad: int, l: int):  # noqa: E741
        key = (pad, l)
        if key not in self._plans:
            if _KEYMASK_PLAN:
                self._plans[key] = K.keymask_plan(self.src, self.lead, self.wrap, pad, l)
            else:
                n = self.shape[1]
                mrow = torch.nn.functional.pad(self.dense().to(torch.float32), (pad, 0), value=0.0).contiguous()
                cnt = mrow.reshape(mrow.shape[0], (n + pad) // l, l).sum(-1)
                self._plans[key] = (mrow, (cnt > 0).float().contiguous(), (float(l) / (cnt + 1e-8)).contiguous())
        return self._plans[key]


_LM_ROWS = True       # test hook (tests/test_fused_epilogue_gpu.py): False = the landmark kernels on q | k instead of NormQkvLmFn
_RELU_SQUARE_PAD = True      # (test hook, round 5) ... also when the sequence carries square-pad rows (N no square: config 4, template)
_RELU_IN_LN_BWD = True      # (test hook, round 5) _fc1's ReLU backward inside layer 1's LayerNorm backward
class _ReluSlot:
    """Hand-over between Fc1SeqFn and the LayerNorm + to_qkv node that reads its sequence (carried on the tensor: `seq._relu_slot`):
    rows 1 .. n of the sequence are a ReLU's output; grad = (data_ptr of the sequence's f32 gradient buffer, the bf16 ReLU-masked gradient
    layer 1's LayerNorm backward wrote, (bias-gradient buffer, came_from_sink) or None)."""
    __slots__ = ("n", "bias", "grad")

    def __init__(self):
        self.n = self.bias = self.grad = None


def layer_norm_landmarks_ok(x, rows: int, pad: int, l: int, prec: Precision) -> bool:
    """geometry / policy for NormQkvLmFn (bf16 activations: the landmark means are rounded like every other to_qkv operand row)"""
    return (_LM_ROWS and prec.act == bf16 and not prec.fp8_fwd and x.dim() == 3 and x.dtype == f32 and x.shape[-1] % 4 == 0
            and x.shape[-1] <= 2048 and (pad + rows) % l == 0 and x.shape[0] * rows >= 64)


_LN_Q8 = True      # LayerNorm writes the e4m3 copy of its output (fp8 policy)


def fp8_site_key(w: torch.Tensor, prec: "Precision"):
    """The key under which `linear(x, w)` keeps the delayed-scaling state of its activation operand (for a producer that
    quantises on its behalf: layer_norm(..., q8_key=))."""
    return (shadow(w, prec).data_ptr(), "x")


_LN_DUAL = True      # (test hook)
_FAN_IN_LN_BWD = True      # (test hook, round 5) the fan-out sum of the encoder output's gradients inside its LayerNorm's backward


class _FanSlot:
    """Hand-over slot from EncFanoutFn.backward to the backward of the LayerNorm whose output it fanned out:
    extra = (x bf16 [B, T - 1, D], alpha, cls f32 [B, D] or None) — the node's dy is (what autograd delivers) + alpha * x on rows 1.. + cls on row 0."""
    __slots__ = ("extra",)

    def __init__(self):
        self.extra = None



def layer_norm(x, gamma, beta, eps, *, rows=None, pad=0, out_dtype=f32, q8_key=None, bf16_copy=False):
    """bf16_copy=True (f32 in / f32 out, no padding): the same launch also writes a bf16 copy of the output, attached to the
    result as `y._bf16` for a consumer that takes bf16 operands (the retention_embed projection) — no cast pass."""
    squeeze = x.dim() == 2
    if squeeze:
        x = x.unsqueeze(0)
    r = x.shape[1] if rows is None else rows
    dual = None
    if (bf16_copy and _LN_DUAL and not squeeze and x.dtype == f32 and out_dtype == f32 and pad == 0 and x.shape[-1] % 4 == 0
            and x.shape[-1] <= 2048 and q8_key is None):
        dual = []
    slot = _FanSlot() if (_FAN_IN_LN_BWD and not squeeze and pad == 0 and out_dtype == f32 and x.dtype == f32) else None
    y = LayerNormFn.apply(x, gamma, beta, eps, r, pad, out_dtype, q8_key, dual, slot)
    if dual:
        y._bf16 = dual[0]
    if slot is not None:
        y._fan_slot = slot        # enc_fanout(y) may leave two of its three gradients here for this node's backward
    return y.squeeze(0) if squeeze else y


# ------------------------------------------------------------------ elementwise with autograd
_res_grads: dict = {}      # data_ptr of a residual-stream tensor -> the gradient tensor its LayerNorm should add into


class AddFn(Function):
    """a + b -> out_dtype (residual adds; a and b may differ in dtype).  residual=True marks `a` as the input of a
    pre-norm block x + f(LN(x)): the backward then publishes a's gradient for that LayerNorm to accumulate into."""

    @staticmethod
    def forward(ctx, a, b, out_dtype, residual):
        ctx.da, ctx.db = a.dtype, b.dtype
        ctx.res_key = a.data_ptr() if (residual and a.is_contiguous()) else None
        return K.add(a.contiguous(), b.contiguous(), out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        ga = K.cast(dy, ctx.da)
        if ctx.res_key is not None and ga.dtype == f32:
            _res_grads[ctx.res_key] = ga
        return ga, K.cast(dy, ctx.db), None, None


def add(a, b, out_dtype=f32, residual=False):
    return AddFn.apply(a, b, out_dtype, residual)


class GeluFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return K.gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        return K.gelu_bwd(x, dy if dy.dtype == x.dtype else K.cast(dy, x.dtype))


gelu = GeluFn.apply

_dropout_state = {"seed": 0x5EED, "offset": 0, "base": None}


def manual_seed(seed: int) -> None:
    _dropout_state["seed"], _dropout_state["offset"] = int(seed) & ((1 << 63) - 1), 0
    if _dropout_state["base"] is not None:
        _dropout_state["base"].zero_()


def dropout_step_begin(device) -> None:
    """Engine protocol for graph-safe dropout: every step starts at host offset 0 (so the per-call offsets are the same
    at every step and can be baked into a captured graph) on top of a per-step base that lives on the device."""
    if _dropout_state["base"] is None or _dropout_state["base"].device != torch.device(device):
        _dropout_state["base"] = torch.zeros(1, device=device, dtype=torch.int64)
    _dropout_state["offset"] = 0


def dropout_step_end() -> None:
    """Advance the device base by what this step consumed (a device-side add: captured with the step)."""
    base = _dropout_state["base"]
    if base is not None and _dropout_state["offset"]:
        base.add_(_dropout_state["offset"])
    _dropout_state["offset"] = 0


def dropout_step_take():
    """dropout_step_end for a caller that advances the device base inside a launch of its own (mh_adam's `counter`): returns
    (base tensor or None, offsets this step consumed) and resets the host offset."""
    base, used = _dropout_state["base"], _dropout_state["offset"]
    _dropout_state["offset"] = 0
    return (base, used) if (base is not None and used) else (None, 0)


def dropout_device_base_off() -> None:
    _dropout_state["base"] = None


_NOISE_OFFSET = 1 << 44      # the noise draws' own range of the Philox counter space: far above any dropout offset of a step


def noise_draws(B: int, N: int, D: int, L: int, device):
    """(rand [B, N], rand [B, D], randn [B, L], randn [B, L]) — the step's four draws (models/mirror.py:630, :516, :832-833) as ONE launch on
    the dropout stream's generator (mh_noise_draws; seed and per-step device base of manual_seed / dropout_step_begin): under a captured
    step torch's generator costs four launches plus two state fills in front of every replay.  The draws live at a FIXED offset of their
    own and take nothing from the running dropout offset — the RNA branch's HIP-graph replay has its dropout offsets baked in from 0
    (graphed.py) and the draws are issued in front of it; what makes them differ from step to step is the device base, so the caller
    advances the running offset by noise_draws_advance() at the END of its forward (a step without a single dropout site would otherwise
    repeat its masks)."""
    n0, n1 = (B * N + 3) // 4 * 4, (B * D + 3) // 4 * 4
    buf = K.noise_draws(n0 + n1, 2 * B * L, _dropout_state["seed"], _NOISE_OFFSET, _dropout_state["base"], device)
    e = n0 + n1
    return (buf[:B * N].view(B, N), buf[n0:n0 + B * D].view(B, D), buf[e:e + B * L].view(B, L), buf[e + B * L:e + 2 * B * L].view(B, L))


def noise_draws_advance() -> None:
    _dropout_state["offset"] = (_dropout_state["offset"] + 7) // 8 * 8 + 8


def _lite_offset() -> int:
    """The next offset of the lite dropout stream (8 elements per Philox block): the running offset rounded up to 8."""
    return (_dropout_state["offset"] + 7) // 8 * 8


# Test hook (train-mode parity): while this is a list, every dropout site of a forward pass appends
# (shape, p, seed, host offset, lite) in launch order; dropout_tap_masks() then regenerates the multipliers the kernels applied
# (0 or 1 / (1 - p)) so that an oracle run can be handed the very masks of a train-mode step.
_dropout_tap: Optional[list] = None


def _tap(shape, p, seed, offset, lite) -> None:
    if _dropout_tap is not None:
        _dropout_tap.append((tuple(shape), float(p), int(seed), int(offset), bool(lite)))


def dropout_tap_masks(records, device, base: Optional[torch.Tensor] = None):
    """The multiplier tensors (f32, 0 or 1 / (1 - p)) of the recorded dropout sites, regenerated by the dropout kernels
    themselves from (seed, offset [+ the device base the step ran under])."""
    out = []
    for shape, p, seed, offset, lite in records:
        ones = torch.ones(shape, device=device, dtype=f32)
        out.append(K.dropout_lite(ones, p, seed, offset, base) if lite else K.dropout(ones, p, seed, offset, dev_base=base))
    return out


class DropoutFn(Function):
    """nn.Dropout in training mode; the Philox mask is regenerated in backward from (seed, offset [+ device base])."""

    @staticmethod
    def forward(ctx, x, p):
        ctx.p, ctx.seed, ctx.offset, ctx.base = p, _dropout_state["seed"], _dropout_state["offset"], _dropout_state["base"]
        _dropout_state["offset"] += (x.numel() + 3) // 4 * 4
        _tap(x.shape, p, ctx.seed, ctx.offset, False)
        return K.dropout(x, p, ctx.seed, ctx.offset, dev_base=ctx.base)

    @staticmethod
    def backward(ctx, dy):
        return K.dropout(dy.contiguous(), ctx.p, ctx.seed, ctx.offset, dev_base=ctx.base), None


def dropout(x, p: float, training: bool):
    if not training or p == 0.0:
        return x
    return DropoutFn.apply(x, p)


class DropoutAddFn(Function):
    """a + dropout(b) -> f32 in one pass (pre-norm block x + Dropout(to_out(...)), models/mirror.py:312): saves the
    dropped copy of b in the forward and, in the backward, the cast of dy in front of the dropout (the kernel reads the
    f32 gradient and writes the masked, scaled activation-dtype gradient).  `a`'s gradient is dy itself and is published
    for the block's LayerNorm to accumulate into (see AddFn)."""

    @staticmethod
    def forward(ctx, a, b, p, lite=False):
        a, b = a.contiguous(), b.contiguous()
        ctx.lite = bool(lite) and b.numel() % 8 == 0        # the lite stream (mh_dropout_lite), the one ToOutDropAddFn's epilogue draws
        ctx.p, ctx.seed, ctx.base = p, _dropout_state["seed"], _dropout_state["base"]
        ctx.res_key, ctx.db = a.data_ptr(), b.dtype
        if ctx.lite:
            ctx.offset = _lite_offset()
            _dropout_state["offset"] = ctx.offset + b.numel()
            _tap(b.shape, p, ctx.seed, ctx.offset, True)
            return K.dropout_lite(b, p, ctx.seed, ctx.offset, ctx.base, add_to=a)
        ctx.offset = _dropout_state["offset"]
        _dropout_state["offset"] += (b.numel() + 3) // 4 * 4
        _tap(b.shape, p, ctx.seed, ctx.offset, False)
        return K.dropout_add(a, b, p, ctx.seed, ctx.offset, dev_base=ctx.base)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        gb = torch.empty(dy.shape, device=dy.device, dtype=ctx.db)
        if ctx.lite:
            K.dropout_lite(dy, ctx.p, ctx.seed, ctx.offset, ctx.base, out=gb)
        else:
            K.dropout(dy, ctx.p, ctx.seed, ctx.offset, out=gb, dev_base=ctx.base)
        _res_grads[ctx.res_key] = dy
        return dy, gb, None, None


def dropout_add(a, b, p: float, training: bool, lite: bool = False):
    """a + dropout(b) with `a` the f32 residual stream of a pre-norm block (falls back to the two-op form otherwise).
    lite=True: the masks of the lite stream (mh_dropout_lite) — the WSI layers, whose fused projection epilogue draws them too."""
    if training and p > 0.0 and a.dtype == f32 and a.shape == b.shape and b.numel() % 4 == 0:
        return DropoutAddFn.apply(a, b, p, lite)
    return add(a, dropout(b, p, training), f32, residual=True)


# ------------------------------------------------------------------ TransMIL sequence assembly
class Fc1SeqFn(Function):
    """seq = [cls | relu(wsi @ W^T + b) | first `add` tokens again]  (models/mirror.py:652-665).
    The GEMM epilogue (bias + ReLU) writes straight into rows 1..N of the f32 sequence buffer."""

    @staticmethod
    def forward(ctx, wsi, w, b, cls, add_len, prec, slot=None):
        Bn, N, Fd = wsi.shape
        D = w.shape[0]
        xa = wsi if wsi.dtype == prec.act else K.cast(wsi.contiguous(), prec.act)
        wa = shadow(w, prec)
        seq = torch.empty((Bn, 1 + N + add_len, D), device=wsi.device, dtype=f32)
        if not (prec.fp8_fwd and xa.dtype == bf16 and _fp8_linear(xa, wa, b.detach(), ACT_RELU, seq[:, 1:1 + N])):
            K.gemm(xa, wa.t(), out=seq[:, 1:1 + N], bias=b.detach(), act=ACT_RELU, mma=prec.mma)
        K.seq_finish(seq, cls.detach().reshape(-1).contiguous(), N, add_len)
        ctx.slot = None
        if slot is not None and prec.act == bf16 and (add_len == 0 or _RELU_SQUARE_PAD):
            # layer 1's LayerNorm backward may write the ReLU-masked gradient itself (NormQkvLmFn.backward): every row behind the cls row is
            # a ReLU's output — the add_len square-pad rows are copies of rows 1 .. add_len
            slot.n, slot.bias = N + add_len, b
            ctx.slot = slot
        ctx.save_for_backward(xa, wa, seq, w, b)
        ctx.add_len, ctx.prec, ctx.cls = add_len, prec, cls
        return seq

    @staticmethod
    def backward(ctx, dseq):
        xa, wa, seq, w, b = ctx.saved_tensors
        prec, add_len = ctx.prec, ctx.add_len
        Bn, N, Fd = xa.shape
        D = wa.shape[0]
        dseq = dseq.contiguous()
        # layer 1's LayerNorm backward already wrote relu'(x) * dx as bf16 (rows 1 .. N of dseq are then stale: only the cls row is read
        # below) and, with it, this Linear's bias gradient (the column sums of that tensor)
        dh = db_have = None
        if ctx.slot is not None and ctx.slot.grad is not None:
            (gptr, gdh, gdb), ctx.slot.grad = ctx.slot.grad, None
            if gptr == dseq.data_ptr():
                dh, db_have = gdh, gdb
        dcls, sunk_c = _gbuf_n(ctx.cls, (D,))
        if dh is not None:
            K.seq_finish_bwd(dseq, dcls, N + add_len, 0)      # the cls row only: the square-pad rows' fold happens on the handed-over rows
        else:
            if add_len:
                dseq = dseq.clone()  # the fold below is in place; autograd owns the incoming buffer
            K.seq_finish_bwd(dseq, dcls, N, add_len)
        dcls = _gret(ctx.cls, dcls, sunk_c)
        if dh is not None and tuple(dh.shape) != (Bn, N + add_len, D):
            raise K.MirrorHipError("Fc1SeqFn.backward: the ReLU-masked gradient handed over by the LayerNorm backward has another shape")
        dw, sunk_w = _gbuf(w, (D, Fd))
        if dh is not None and add_len:
            # the square-pad rows are copies of rows 1 .. add_len: their (already ReLU-masked: same x, same gate) gradients fold onto those
            # rows, then the weight gradient contracts the first N rows of every slide (a batch-strided operand: no 67 MB compaction)
            dh[:, :add_len] += dh[:, N:N + add_len]
            dh = dh[:, :N]
            K.gemm(dh.transpose(-1, -2), xa, out=dw.expand(Bn, D, Fd), accumulate=True, split_k=_split_k_for(N, D, Fd, batch=Bn), mma=prec.mma)
        else:
            if dh is None:
                dh = K.relu_bwd(seq[:, 1:1 + N], dseq[:, 1:1 + N], out_dtype=prec.act)   # [B,N,D] contiguous
            K.gemm(dh.reshape(Bn * N, D).t(), xa.reshape(Bn * N, Fd), out=dw, accumulate=True,
                   split_k=_split_k_for(Bn * N, D, Fd), mma=prec.mma)
        dw = _gret(w, dw, sunk_w)
        if db_have is not None:
            db = _gret(b, *db_have)
        else:
            db, sunk_b = _gbuf(b, (D,))
            K.colsum(dh.reshape(Bn * N, D), db)
            db = _gret(b, db, sunk_b)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.gemm(dh if dh.is_contiguous() else dh.contiguous(), wa, mma=prec.mma, out_dtype=f32)
        return dx, dw, db, None if dcls is None else dcls.reshape(1, 1, D), None, None, None


def fc1_seq(wsi, w, b, cls, add_len: int, prec: Precision):
    """[cls | relu(_fc1(wsi)) | first add_len tokens again] (Fc1SeqFn); the sequence carries the slot through which layer 1's LayerNorm
    backward hands the ReLU-masked gradient back."""
    slot = _ReluSlot() if _RELU_IN_LN_BWD else None
    seq = Fc1SeqFn.apply(wsi, w, b, cls, add_len, prec, slot)
    if slot is not None and slot.n is not None:
        seq._relu_slot = slot
    return seq


_PPEG_SCATTER = True      # (test hook)


class PPEGFn(Function):
    """PPEG.forward (models/mirror.py:324-331) as one merged depthwise 7x7 on the token-major sequence."""

    @staticmethod
    def forward(ctx, x, w7, b7, w5, b5, w3, b3, S):
        x = x.contiguous()
        merged, bsum = K.ppeg_merge(w7.detach(), w5.detach(), w3.detach(), b7.detach(), b5.detach(), b3.detach())
        ctx.save_for_backward(x, merged, bsum)
        ctx.S, ctx.params = S, (w7, b7, w5, b5, w3, b3)
        return K.ppeg(x, merged, bsum, S, flip=False)

    @staticmethod
    def backward(ctx, dy):
        x, merged, bsum = ctx.saved_tensors
        S = ctx.S
        D = x.shape[-1]
        dy = dy.contiguous()
        dx = K.ppeg(dy, merged, bsum, S, flip=True)
        dm = zeros(tuple(merged.shape), merged.device)
        dbs = zeros(tuple(bsum.shape), bsum.device)
        K.ppeg_wgrad(x, dy, dm, dbs, S)
        if not _PPEG_SCATTER:       # A/B: the torch-side split (13 tiny launches with autograd's accumulation)
            dm = dm.t().reshape(D, 1, 7, 7)
            return (dx, dm.contiguous(), dbs, dm[:, :, 1:6, 1:6].contiguous(), dbs.clone(), dm[:, :, 2:5, 2:5].contiguous(), dbs.clone(), None)
        # merged gradient -> the six parameter gradients in one accumulating launch (straight into the engine's arena)
        bufs = [_gbuf(p, tuple(p.shape)) for p in ctx.params]
        K.ppeg_grad_scatter(dm, dbs, *[bufs[i][0] for i in (0, 2, 4, 1, 3, 5)])
        g = [_gret(p, b, sunk) for p, (b, sunk) in zip(ctx.params, bufs)]
        return (dx, g[0], g[1], g[2], g[3], g[4], g[5], None)


# ------------------------------------------------------------------ Nystrom attention core
_side_streams: dict = {}


def _side_stream(device, which: int = 0) -> torch.cuda.Stream:
    """Per-device helper streams: 0 = half-chip pinv chain, 1 = RNA encoder + alignment / style heads."""
    key = (torch.device(device).index or 0, which)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


def join_side_streams(device, stream=None) -> None:
    """`stream` (default: the current one) waits for everything queued on the helper streams of `device`.  Autograd's end-of-backward
    synchronisation only covers streams on which an AccumulateGrad node ran; with the gradient sink the weight gradients are
    written by kernels launched inside Function.backward (or by a replayed HIP graph, graphed.py) on the branch's stream, so
    whoever reads the gradient arena next (all-reduce, clip, Adam) has to join those streams itself."""
    idx = torch.device(device).index or 0
    cur = stream if stream is not None else torch.cuda.current_stream()
    for (d, _), st in _side_streams.items():
        if d == idx and st != cur:
            cur.wait_stream(st)


def _heads(t3: torch.Tensor, which: int, parts: int, h: int) -> torch.Tensor:
    """[B, T, parts*D] buffer -> [B, h, T, dh] view of column block `which` (heads are dh-wide column slices)."""
    Bn, T, Dt = t3.shape
    dh = Dt // parts // h
    return t3.view(Bn, T, parts, h, dh)[:, :, which].permute(0, 2, 1, 3)


def pinv_forward(a2: torch.Tensor, iters: int, pm: int, sd: torch.dtype):
    """[3P] moore_penrose_iter_pinv: z0 = a2^T / (max rowsum * max colsum) over the WHOLE tensor, then
    z <- 1/4 z (13I - P (15I - P (7I - P))), P = a2 z.  The middle factor is expanded, T2 = 15I - 7P + P.P, so every
    step is one GEMM with an epilogue (diag / R addend) and nothing else is launched.  Iterates and intermediates
    are stored in `sd` (f32, or bf16 when the GEMMs round their operands to bf16 anyway) and KEPT for the backward.
    Returns (z_final, [(z_k, P_k, T2_k, T3_k)], stats)."""
    st = K.pinv_absmax(a2, zeros((4,), a2.device).view(torch.int64))
    z = K.cast(K.pinv_z0(a2, st), sd)
    saved = []
    for _ in range(iters):
        P = K.gemm(a2, z, mma=pm, out_dtype=sd)
        T2 = K.gemm(P, P, diag=15.0, R=P, rcoef=-7.0, mma=pm, out_dtype=sd)
        T3 = K.gemm(P, T2, alpha=-1.0, diag=13.0, mma=pm, out_dtype=sd)
        zn = K.gemm(z, T3, alpha=0.25, mma=pm, out_dtype=sd)
        saved.append((z, P, T2, T3))
        z = zn
    return z, saved, st


def pinv_backward(a2, saved, st, dZ, pm, sd):
    """Reverse mode through the iterations: 8 GEMMs per step, gradients accumulate in f32."""
    dX = torch.zeros_like(a2)
    dz = dZ
    tr = lambda t: t.transpose(-1, -2)  # noqa: E731
    for z, P, T2, T3 in reversed(saved):
        dT3 = K.gemm(tr(z), dz, alpha=0.25, mma=pm, out_dtype=sd)                  # z' = 1/4 z T3
        dz_new = K.gemm(dz, tr(T3), alpha=0.25, mma=pm, out_dtype=f32)
        dP = K.gemm(dT3, tr(T2), alpha=-1.0, mma=pm, out_dtype=f32)                # T3 = 13I - P T2
        dT2 = K.gemm(tr(P), dT3, alpha=-1.0, mma=pm, out_dtype=f32)
        K.gemm(dT2, tr(P), out=dP, accumulate=True, R=dT2, rcoef=-7.0, mma=pm)     # T2 = 15I - 7P + P P
        K.gemm(tr(P), dT2, out=dP, accumulate=True, mma=pm)
        K.gemm(dP, tr(z), out=dX, accumulate=True, mma=pm)                         # P = a2 z
        K.gemm(tr(a2), dP, out=dz_new, accumulate=True, mma=pm)
        dz = dz_new
    z0 = saved[0][0]
    K.pinv_z0_bwd(a2, K.cast(z0, f32), dz, st, dX)
    return dX


def pinv_forward_tile(a2: torch.Tensor, iters: int, want_a2b: bool = False):
    """pinv_forward for landmark counts the 192 x 384 tile kernel takes (the template's m = 384): every operand bf16, f32
    accumulation inside a product, one launch per product at one workgroup per CU (csrc/gemm_tile.hip).  The iterates are
    written into stacks (zs[k] = z_k; pt[k] = (T2_k, P_k)) so that the backward pass can sum products over an iteration's or
    the whole chain's operand pairs in one launch (K.gemm_ksum)."""
    st = K.pinv_absmax(a2, zeros((4,), a2.device).view(torch.int64))
    a2b = K.cast(a2, bf16)
    zs = torch.empty((iters + 1,) + tuple(a2.shape), device=a2.device, dtype=bf16)
    pt = torch.empty((iters, 2) + tuple(a2.shape), device=a2.device, dtype=bf16)
    K.cast(K.pinv_z0(a2, st), bf16, out=zs[0])
    saved = []
    for k in range(iters):
        z, P, T2 = zs[k], pt[k, 1], pt[k, 0]
        K.gemm(a2b, z, P, mma=MH_BF16)
        K.gemm(P, P, T2, diag=15.0, R=P, rcoef=-7.0, mma=MH_BF16)
        T3 = K.gemm(P, T2, alpha=-1.0, diag=13.0, mma=MH_BF16)
        K.gemm(z, T3, zs[k + 1], alpha=0.25, mma=MH_BF16)
        saved.append((z, P, T2, T3))
    if want_a2b:        # the bf16 copy of attn2 is the backward's operand too: handed over instead of cast again (38 MB kept, a 113 MB pass saved)
        return zs[iters], saved, st, a2b
    return zs[iters], saved, st


_TILE_SIDE = True      # (test hook, round 5) the tile-kernel pinv iteration (m = 384) on the side stream beside the attention sides' products
_PINV_R32 = True      # (test hook, round 5) f32 partial sums of the tile-kernel pinv backward as addends of bf16-output products


def pinv_backward_tile(a2, saved, st, dZ, a2b=None):
    """Reverse mode of pinv_forward_tile, operands bf16.  Sums of products that share an operand layout run as ONE launch over
    several operand pairs (K.gemm_ksum: the f32 sum stays in the accumulators): dP's two `x @ y^T` terms per iteration, and
    dX = sum_k dP_k z_k^T over the whole chain at the end; the launch that completes a sum writes its bf16 copy (the next
    operand) in the same epilogue."""
    tr = lambda t: t.transpose(-1, -2)  # noqa: E731
    iters = len(saved)
    if a2b is None:
        a2b = K.cast(a2, bf16)
    dz = K.cast(dZ, bf16) if dZ.dtype != bf16 else dZ
    dzn = dZ
    z0 = saved[0][0]
    stacked = (iters > 1 and saved[1][0].data_ptr() - z0.data_ptr() == z0.numel() * 2
               and saved[0][1].data_ptr() - saved[0][2].data_ptr() == z0.numel() * 2)     # pinv_forward_tile's stacks
    dPs = torch.empty((iters,) + tuple(a2.shape), device=a2.device, dtype=bf16)
    dd = torch.empty((2,) + tuple(a2.shape), device=a2.device, dtype=bf16)
    for k in range(iters - 1, -1, -1):
        z, P, T2, T3 = saved[k]
        nT3, dT2, dPb = dd[0], dd[1], dPs[k]
        K.gemm(tr(z), dz, nT3, alpha=-0.25, mma=MH_BF16)                               # z' = 1/4 z T3:  nT3 = -dT3
        dzn = K.gemm(dz, tr(T3), alpha=0.25, mma=MH_BF16, out_dtype=f32)
        K.gemm(tr(P), nT3, dT2, mma=MH_BF16)                                           # T3 = 13I - P T2: dT2 = -P^T dT3
        if stacked:     # dP = -dT3 T2^T + dT2 P^T - 7 dT2 (T2 = 15I - 7P + P P) in one launch over the pairs (nT3, T2), (dT2, P)
            dP = K.gemm_ksum(dd, tr(_pair(T2, P)), R=dT2, rcoef=-7.0, mma=MH_BF16, out_dtype=f32)
        else:
            dP = K.gemm(nT3, tr(T2), mma=MH_BF16, out_dtype=f32)
            K.gemm(dT2, tr(P), out=dP, accumulate=True, R=dT2, rcoef=-7.0, mma=MH_BF16)
        # the third term joins the f32 sum of the other two and leaves as bf16 — the only form the products below read: the f32 sum is
        # an addend (R), not a read-modify-write destination (round 5: 75 MB of f32 stores less per product at the template's m = 384)
        if _PINV_R32:
            K.gemm(tr(P), dT2, out=dPb, R=dP, rcoef=1.0, mma=MH_BF16)
        else:
            K.gemm(tr(P), dT2, out=dP, accumulate=True, mma=MH_BF16, c2=dPb)
        dz = torch.empty(dzn.shape, device=dzn.device, dtype=bf16)
        if _PINV_R32 and k > 0:
            K.gemm(tr(a2b), dPb, out=dz, R=dzn, rcoef=1.0, mma=MH_BF16)                    # P = a2 z; only the last d z (k = 0) is read in f32
        else:
            K.gemm(tr(a2b), dPb, out=dzn, accumulate=True, mma=MH_BF16, c2=dz)
    if stacked:
        zst = torch.as_strided(z0, (iters,) + tuple(z0.shape), (z0.numel(),) + tuple(z0.stride()))
        dX = K.gemm_ksum(dPs, tr(zst), mma=MH_BF16, out_dtype=f32)
    else:
        dX = torch.zeros_like(a2)
        for k in range(iters):
            K.gemm(dPs[k], tr(saved[k][0]), out=dX, accumulate=True, mma=MH_BF16)
    # z_0 is formed on the fly from attn2 and the maxima (as on the chain path: no f32 copy of the bf16 z_0 the forward stored)
    K.pinv_z0_bwd(a2, None, dzn, st, dX)
    return dX


def _pair(first: torch.Tensor, second: torch.Tensor) -> torch.Tensor:
    """[2, ...] view of two equally shaped tensors that sit one tensor apart in memory (first, then second)."""
    return torch.as_strided(first, (2,) + tuple(first.shape), (first.numel(),) + tuple(first.stride()))


_S2_TAIL = True      # (test hook)
_Z0_ROWS = True      # (test hook)
_SIM2_MASKED = True      # (test hook, round 5) the one-launch sim2 + softmax + maxima + panels also under a key-padding mask (config 4)
_SIM2_SIDE = True      # nys_sim2 opens the chain's branch instead of preceding the fork
_S2_SIDE = True      # sim2's landmark gradients on the chain's stream
_LM_MERGE_LATE = True      # (test hook) the landmark rows' merge + data gradient beside the sequence rows' data gradient (-0.24 % +- 0.29)
_pending_lm_merge: dict = {}      # data_ptr of the landmark-gradient view NystromCoreFn.backward returned -> (address of the buffer
                                  # `de` the merge writes into, its deferred merge launch); see pending_lm_merge_reset


def pending_lm_merge_reset(where: str, strict: bool = False) -> None:
    """Entries left behind (a standalone TransLayer whose NormQkvLmFn never ran its backward, an aborted backward pass) pin tens of MB
    through their closures and — once the allocator recycles the address — could match a later gradient view.  TrainEngine clears the
    table at the start of a step and, with strict=True, raises behind loss.backward() if a merge was never consumed."""
    if _pending_lm_merge:
        n = len(_pending_lm_merge)
        _pending_lm_merge.clear()
        if strict:
            raise K.MirrorHipError(f"{where}: {n} deferred landmark-gradient merge(s) were never run (NormQkvLmFn.backward did not follow "
                                   "NystromCoreFn.backward)")
_W2_ON_CHAIN = True      # (test hook) w2 = pinv (attn3 v) at the end of the chain's branch instead of behind the join (-0.22 % +- 0.06)
_DZ_DAV = True      # (test hook)
_RC_FUSED = True      # (test hook) res_conv inside attn3's forward launch, its two gradients as one pass over dout (round 5)
_A1_DQ_IN_WINDOW = False     # (test hook, round 5) attn1's dq kernel beside the pinv chain's backward: -0.35 % +- 0.02 when it was built, but
                             # +0.80 % +- 0.20 (8 ABBA rounds) on the round's final tree — the one-pass attn3 backward and the other kernels that joined the
                             # window since made its main side the longer one; dq stands in front of the fork again (profiles/r05_p_*)
# (measured and deleted in round 4, see DESIGN.md section 6 round 3: nys_dz_dav on the chain's branch +0.32 %, attn3's delta out of
#  nys_dz_dav +0.32 %, the chain branch joined in front of the landmark projection's backward +0.02 %, res_conv's weight gradient on
#  the chain's stream +0.26 % or in front of the fork: neutral)


class NystromCoreFn(Function):
    """[3P] NystromAttention between to_qkv and to_out (called at models/mirror.py:312):
    qkv [B, n_p, 3D] -> out [B, n_p, D] = softmax(q k_l^T) . pinv(softmax(q_l k_l^T)) . softmax(q_l k^T) v + res_conv(v).
    The product is associated as a1 @ (a2inv @ (a3 @ v)) (algebraically equal to the reference's
    (a1 @ a2inv) @ (a3 @ v), 2 m^2 D instead of 2 n_p m^2 h flops)."""

    @staticmethod
    def forward(ctx, qkv, res_w, heads, l, iters, prec, kmask=None, q8_key=None, lm_ext=None):
        """lm_ext: the q | k landmarks [B, m, 2D] computed by the caller (NormQkvLmFn: rows of to_qkv's output, row stride 3D); their
        gradient is returned instead of being scattered into dqkv.
        kmask: None, or the package's key-padding mask prepared by TransLayer as (rows [B, n_p], landmarks [B, m],
        landmark scale [B, m]) float tensors: rows of qkv that are masked out are already zero (the caller zeroes the
        LayerNorm output in front of the bias-free to_qkv); here the landmark means become masked means and the three
        similarity matrices are masked_fill'ed before their softmax."""
        qkv = qkv.contiguous()
        Bn, n_p, D3 = qkv.shape
        D, h = D3 // 3, heads
        dh = D // h
        A, mma, pm = prec.act, prec.mma, prec.pinv_mma
        scale = dh ** -0.5
        q, k, v = (_heads(qkv, i, 3, h) for i in range(3))
        if lm_ext is not None:      # rows of to_qkv's output behind the sequence (NormQkvLmFn): row stride 3D, read in place
            lm = lm_ext if K._lm_ld(lm_ext) > 0 else lm_ext.contiguous()
        else:
            lm = K.landmark_fwd(qkv, l)
        ctx.lm_ext = lm_ext is not None
        if kmask is not None:
            mrow, mlm, lscale = kmask
            if lscale is not None:       # (None: the caller's landmark rows are masked means already — NormQkvLmFn(lscale))
                lm = K.row_scale(lm if lm.is_contiguous() else lm.contiguous(), lscale)      # sum over the group / (valid count + 1e-8)
        ql, kl = _heads(lm, 0, 2, h), _heads(lm, 1, 2, h)
        m_l = lm.shape[1]
        chain = pm == MH_BF16 and m_l == K.PINV_CHAIN_M    # whole iteration in one launch (pinv_panel.hip)
        # sim2, its softmax, the tensor-wide abs-sum maxima and the chain's operand packing in ONE launch (nystrom_sim2.hip)
        one = chain and (kmask is None or _SIM2_MASKED) and dh == 64 and K.nys_sim2_ok(lm, h)      # (mask-aware since round 5: mlm of mh_nys_sim2)
        mlm_s2 = None if kmask is None else mlm
        z0f = None
        sim2_side = one and _SIM2_SIDE and _Z0_ROWS
        if sim2_side:
            # nothing on the main stream needs attn2 before the join: the launch (128 workgroups, half of the chip, ~45 us) opens the
            # chain's branch instead of standing in front of the fork.  Buffers from the main stream's allocator, as chain_saved.
            a2, xt = K.nys_sim2_alloc(lm, h)
            st = zeros((4,), qkv.device).view(torch.int64)
        elif one:
            a2, xt, z0f, st = K.nys_sim2(lm, h, scale, zeros((4,), qkv.device).view(torch.int64), want_z0f=not _Z0_ROWS, mlm=mlm_s2)
        else:
            a2 = K.gemm(ql, kl.transpose(-1, -2), alpha=scale, mma=mma, out_dtype=f32)      # [B,h,m,m]
            if kmask is None:
                K.softmax_fwd(a2, a2)
            else:
                K.softmax_masked_fwd(a2, mlm, mlm, a2)
        sd = f32 if pm == MH_F32 else bf16
        side = None
        if chain:
            # B*h workgroups with a 128 KiB LDS image each: the chain owns B*h CUs and nothing else.  At B*h = 128 that
            # is half of the chip, so it runs on a side stream beside the attn3 side on the main stream; they meet
            # again at w2 = pinv @ (a3 v).  Chain-private matrices are column-major (see mirror_hip.h).
            chain_saved = K.pinv_chain_saved_alloc(iters, Bn * h, m_l, qkv.device)
            z0 = None
            if not one:
                st = K.pinv_absmax(a2, zeros((4,), a2.device).view(torch.int64))
                z0, xt = K.pinv_chain_prep(a2, st, K.pinv_chain_z0_slot(chain_saved))
            zfT = torch.empty((Bn, h, m_l, m_l), device=qkv.device, dtype=bf16)
            zf = zfT.transpose(-1, -2)
            side = _side_stream(qkv.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if sim2_side:
                    K.nys_sim2(lm, h, scale, st, out=(a2, xt), mlm=mlm_s2)
                if one and _Z0_ROWS:      # z_0 from the rows of attn2 inside the chain launch: nys_sim2 has no second pass and no f32 transpose
                    K.pinv_chain_fwd(xt, chain_saved, zfT, iters, z0f=a2, stats=st, z0_rowmajor=True)
                else:
                    K.pinv_chain_fwd(xt, chain_saved, zfT, iters, z0f=z0f, stats=st if one else None)
            saved = [(xt, chain_saved, z0 if z0 is not None else st)]      # (no stored f32 z_0 on the one-launch path: a placeholder)
            ctx.z0_stored = z0 is not None
            K.shared_chip = True         # until the join below: no persistent GEMM kernel beside the half-chip chain
        tile = (not chain) and pm == MH_BF16 and K.gemm_tile_ok(m_l, m_l, m_l)
        if tile and _TILE_SIDE:
            # the template's m = 384: the iteration is 24 dependent one-round launches (~36 us each, half of that fixed cost) that need
            # attn2 only — they run on the side stream beside sim1 / sim3, their softmax and attn3 v, and meet them at w2 (round 5)
            side = _side_stream(qkv.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                zf, saved, st, ctx.a2b = pinv_forward_tile(a2, iters, want_a2b=True)
        run_deferred(qkv)        # the v columns of to_qkv: nothing above reads them (landmarks are means of q and k)
        fused = K.nys_fused_ok(qkv, h, m_l) and lm.dtype == bf16      # nystrom_fused.hip: sim1 / sim3 never reach HBM (mask-aware)
        lse1 = lse3 = a1 = a3 = None
        out = torch.empty((Bn, n_p, D), device=qkv.device, dtype=A)
        rc_in_a3 = fused and _RC_FUSED and A == bf16 and res_w.numel() == h * 33
        if fused:
            # res_conv(v) rides on attn3's forward (it stages the v tiles anyway): no launch of its own, no second read of v
            av, lse3 = K.nys_attn3_fwd(qkv, lm, h, scale, kmask,
                                       rc=(res_w.detach().reshape(-1).contiguous(), out) if rc_in_a3 else None)   # [B,h,m,dh] f32
        else:
            # sim1's rows (length m) fit one 192 x 384 tile at the template's m = 384: the softmax runs in that GEMM's epilogue
            sm1 = (kmask is None and A == bf16 and mma == MH_BF16 and q.dtype == bf16
                   and K.gemm_softmax_ok(n_p, lm.shape[1], dh, q.dtype, kl.dtype))
            if sm1:
                a1 = K.gemm(q, kl.transpose(-1, -2), alpha=scale, mma=mma, out_dtype=bf16, softmax=True)   # [B,h,n_p,m]
            else:
                a1 = K.gemm(q, kl.transpose(-1, -2), alpha=scale, mma=mma, out_dtype=f32)
            a3 = K.gemm(ql, k.transpose(-1, -2), alpha=scale, mma=mma, out_dtype=f32)   # [B,h,m,n_p]
            if kmask is None:
                if not sm1:
                    a1 = K.softmax_fwd(a1, a1 if A == f32 else None, out_dtype=A)
                a3 = K.softmax_fwd(a3, a3 if A == f32 else None, out_dtype=A)
            else:
                a1 = K.softmax_masked_fwd(a1, mrow, mlm, a1 if A == f32 else None, out_dtype=A)
                a3 = K.softmax_masked_fwd(a3, mlm, mrow, a3 if A == f32 else None, out_dtype=A)
        if tile and not _TILE_SIDE:
            zf, saved, st, ctx.a2b = pinv_forward_tile(a2, iters, want_a2b=True)
        elif not chain and not tile:
            zf, saved, st = pinv_forward(a2, iters, pm, sd)
        if not fused:
            av = K.gemm(a3, v, mma=mma, out_dtype=f32)                                   # [B,h,m,dh]
        # GEMMs that meet activation-dtype tensors cannot use the exact-f32 MFMA unless the activations are f32 too
        pio = pm if (pm == MH_BF16 or A == f32) else mma
        w2 = None
        if side is not None and fused and _W2_ON_CHAIN:
            # w2 = pinv(attn2) (attn3 v) is a [256 x 256] x [256 x 64] product per (b, h): ~11 us of launch that stood alone behind the
            # join.  av is complete here (the main side of the window is past attn3), the pseudo-inverse when the chain's stream gets to
            # it: the product goes to the END of the chain's branch, beside res_conv on the main side, and the join covers it.
            w2 = torch.empty((Bn, h, m_l, dh), device=qkv.device, dtype=A)
            av_ready = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(side):
                side.wait_event(av_ready)
                K.gemm(zf, av, out=w2, mma=pio)
        if fused and not rc_in_a3:   # res_conv(v) does not need the pseudo-inverse: it runs under the chain, attn1 then adds to it
            K.resconv(qkv[..., 2 * D:], res_w.detach().contiguous(), out, h, transpose=False, accumulate=False)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            if chain:
                K.shared_chip = False
        if w2 is None:
            w2 = K.gemm(zf, av, mma=pio, out_dtype=A)
        o1 = None
        if fused:
            # attn1's own rows (without res_conv's), kept for the backward: delta[n] = sum_d dO[n, d] o1[n, d] there
            o1 = torch.empty_like(out)
            # fp8 forward policy: once to_out's call site has a scale history, attn1 also writes the e4m3 copy that projection reads
            st8 = _fp8_state
            site = st8["sites"].get(q8_key) if (q8_key is not None and st8["tick"] is not None and kmask is None) else None
            if site is not None and st8["host_step"] - site[1] >= 2 and _LN_Q8:
                q8 = torch.empty(out.shape, device=out.device, dtype=torch.uint8)
                lse1, sc8 = K.nys_attn1_fwd_q8(qkv, lm, w2, out, h, scale, True, q8, site[0], st8["tick"], o1=o1)
                _prequant[out.data_ptr()] = (q8, sc8)
            else:
                lse1 = K.nys_attn1_fwd(qkv, lm, w2, out, h, scale, accumulate=True, kmask=kmask, o1=o1)
        else:
            K.gemm(a1, w2, out=_heads(out, 0, 1, h), mma=mma)
            K.resconv(qkv[..., 2 * D:], res_w.detach().contiguous(), out, h, transpose=False, accumulate=True)
        stats = (lse1, lse3) if fused else (a1, a3)
        ctx.has_o1 = o1 is not None
        ctx.save_for_backward(qkv, res_w, lm, stats[0], a2, stats[1], av, w2, st, zfT if chain else zf,
                              *([o1] if o1 is not None else []), *[t for it in saved for t in it])
        ctx.cfg = (heads, l, prec)
        ctx.chain = (chain, iters, fused)
        ctx.tile = tile
        ctx.kmask = kmask
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, res_w, lm, a1, a2, a3, av, w2, st, zf, *flat = ctx.saved_tensors
        o1 = None
        if ctx.has_o1:
            o1, flat = flat[0], flat[1:]
        chain, iters, fused = ctx.chain
        kmask = ctx.kmask
        if kmask is not None:
            mrow, mlm, lscale = kmask
        sm_bwd = (lambda y, dy, rm, cm: K.softmax_bwd(y, dy)) if kmask is None else K.softmax_masked_bwd
        zfT = zf if chain else None         # the chain's column-major output as saved
        if chain:
            zf = zf.transpose(-1, -2)       # saved as the column-major chain output
        saved = None if chain else [tuple(flat[i:i + 4]) for i in range(0, len(flat), 4)]
        heads, l, prec = ctx.cfg
        A, mma, pm = prec.act, prec.mma, prec.pinv_mma
        Bn, n_p, D3 = qkv.shape
        D, h = D3 // 3, heads
        dh = D // h
        m = n_p // l
        scale = dh ** -0.5
        tr = lambda t: t.transpose(-1, -2)  # noqa: E731
        dout = dout.contiguous()
        if dout.dtype != A:
            dout = K.cast(dout, A)
        q, k, v = (_heads(qkv, i, 3, h) for i in range(3))
        ql, kl = _heads(lm, 0, 2, h), _heads(lm, 1, 2, h)
        dO = _heads(dout, 0, 1, h)
        # with caller-made landmarks (NormQkvLmFn) d qkv and d lm are the two row ranges of ONE buffer: to_qkv's backward then
        # multiplies the landmark rows [dq_l | dk_l | 0] together with the sequence rows
        de = dlm_out = None
        if ctx.lm_ext and A == bf16:
            de, dqkv, dlm_out = ext_rows_alloc(Bn, n_p, m, D3, 2 * D, qkv.device)
        else:
            dqkv = torch.empty_like(qkv)
        dq, dk, dv = (_heads(dqkv, i, 3, h) for i in range(3))
        rw = res_w.detach().contiguous()
        dres, dres_sunk = _gbuf(res_w, (rw.numel(),))     # the 33-tap filters' gradient goes straight into the grad arena
        # out = a1 @ w2 ; w2 = Z @ av ; av = a3 @ v.  dZ first: it is all the pinv backward needs (the res_conv weight
        # gradient does not depend on it and runs beside the chain, below).
        if fused:
            lse1, lse3 = a1, a3
            dW2 = zeros((Bn, h, m, dh), qkv.device)
            dlm = zeros((Bn, m, 2 * D), qkv.device)
            delta1 = torch.empty_like(lse1)
            # dW2 and dk_l (+ delta1 from the saved rows of attn1): all the chain's backward waits for.  dq follows BESIDE the chain
            K.nys_attn1_bwd(qkv, lm, w2, dout, lse1, o1, delta1, dqkv, dW2, dlm, h, scale, kmask, which=1)
            if not (chain and _A1_DQ_IN_WINDOW):
                K.nys_attn1_bwd(qkv, lm, w2, dout, lse1, o1, delta1, dqkv, dW2, dlm, h, scale, kmask, which=2)
        else:
            dW2 = K.gemm(tr(a1), dO, mma=mma, out_dtype=f32)                             # [B,h,m,dh]
            dlm = torch.empty((Bn, m, 2 * D), device=qkv.device, dtype=f32)
        dql, dkl = _heads(dlm, 0, 2, h), _heads(dlm, 1, 2, h)
        sd = f32 if pm == MH_F32 else bf16
        pio = pm if (pm == MH_BF16 or A == f32) else mma
        # dZ = dW2 av^T (the chain's input, packed) and dAV = Z^T dW2: ONE launch on the fused bf16 path (nystrom_sim2.hip)
        one2 = (_DZ_DAV and chain and fused and A == bf16 and pio == MH_BF16 and dh == 64 and m == 256 and dW2.dtype == f32 and av.dtype == f32)
        dAV = dzb = None
        if one2:
            dzb, dAV = K.nys_dz_dav(dW2, av.contiguous(), zfT)
        else:
            # (the tile-kernel iteration reads d Z in bf16 only: the product rounds it itself instead of a cast pass behind an f32 copy)
            dZ = K.gemm(dW2, tr(av), mma=pm, out_dtype=bf16 if (ctx.tile and _PINV_R32 and iters >= 1) else f32)      # [B,h,m,m]
        side = dlm2 = None
        if chain:
            xb, chain_saved, z0 = flat
            work = K.pinv_chain_work_alloc(iters, Bn * h, m, qkv.device)
            dS2 = torch.empty_like(a2)
            dz0 = torch.empty_like(a2)
            if dzb is None:
                dzb = K.pinv_chain_pack(dZ)
            if _S2_SIDE and fused and (kmask is None or _SIM2_MASKED):
                # sim2's share of the landmark gradients leaves the serial tail behind the join: the chain's stream has slack
                # in this window.  The attention kernels on the main stream add into dlm with atomics meanwhile, so the two
                # products go to a buffer of their own and one add (which is also the cast) merges them after the join.
                dlm2 = torch.empty_like(dlm)
            side = _side_stream(qkv.device)      # half-chip chain again, beside the softmax backward / dq / dk work
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                K.pinv_chain_bwd(xb, chain_saved, dzb, work, dS2, dz0, iters)
                if _S2_TAIL and (kmask is None or _SIM2_MASKED) and not ctx.z0_stored and a2.shape[-1] == 256:
                    # z_0 backward, the maxima's sub-gradients and attn2's (masked) softmax backward: one pass
                    K.pinv_s2_bwd(a2, dz0, st, dS2, _zeroed1(a2.device), mlm=mlm if kmask else None, heads=h)
                else:
                    K.pinv_z0_bwd(a2, z0 if ctx.z0_stored else None, dz0, st, dS2, _zeroed1(a2.device))
                    sm_bwd(a2, dS2, mlm if kmask else None, mlm if kmask else None)
                if dlm2 is not None:
                    K.gemm(tr(dS2), ql, out=_heads(dlm2, 1, 2, h), alpha=scale, mma=pio)
                    K.gemm(dS2, kl, out=_heads(dlm2, 0, 2, h), alpha=scale, mma=pio)
            K.shared_chip = True         # until the join below
        tile_side = ctx.tile and _TILE_SIDE and not chain
        if tile_side:      # the 54 launches of the m = 384 iteration's backward beside the attention sides' gradient products
            dZ_ready = torch.cuda.current_stream().record_event()
            side = _side_stream(qkv.device)
            with torch.cuda.stream(side):
                side.wait_event(dZ_ready)
                dS2 = pinv_backward_tile(a2, saved, st, dZ, getattr(ctx, "a2b", None))
                sm_bwd(a2, dS2, mlm if kmask else None, mlm if kmask else None)
        run_deferred_bwd(dout)      # to_out's weight gradient (ToOutDropAddFn left it to us): beside the chain when there is one
        if fused and chain and _A1_DQ_IN_WINDOW:
            K.nys_attn1_bwd(qkv, lm, w2, dout, lse1, o1, delta1, dqkv, dW2, dlm, h, scale, kmask, which=2)     # dq: beside the chain
        rc_bwd_one = fused and _RC_FUSED and A == bf16      # both res_conv gradients in ONE pass over dout, behind attn3's backward
        if not rc_bwd_one:
            K.resconv_wgrad(qkv[..., 2 * D:], dout, dres, h)
        if dAV is None:
            dAV = K.gemm(tr(zf), dW2, mma=pio, out_dtype=A)                              # [B,h,m,dh]
        if fused:
            K.nys_attn3_bwd(qkv, lm, av, dAV, lse3, dqkv, dlm, h, scale, kmask)          # dk, dv, dq_l
            if rc_bwd_one:
                K.resconv_bwd(dout, qkv[..., 2 * D:], rw, dqkv[..., 2 * D:], dres, h)
            else:
                K.resconv(dout, rw, dqkv[..., 2 * D:], h, transpose=True, accumulate=True)
        else:
            if (kmask is None and A == bf16 and mma == MH_BF16 and a1.dtype == bf16 and a1.is_contiguous()
                    and K.gemm_softmax_ok(n_p, m, dh, dO.dtype, w2.dtype)):
                dS1 = K.gemm(dO, tr(w2), mma=mma, out_dtype=bf16, softmax_bwd_of=a1)    # softmax backward in the epilogue
            else:
                dS1 = K.gemm(dO, tr(w2), mma=mma, out_dtype=A)                           # [B,h,n_p,m]
                sm_bwd(a1, dS1, mrow if kmask else None, mlm if kmask else None)
            dS3 = K.gemm(dAV, tr(v), mma=mma, out_dtype=A)                               # [B,h,m,n_p]
            K.gemm(tr(a3), dAV, out=dv, mma=mma)
            K.resconv(dout, rw, dqkv[..., 2 * D:], h, transpose=True, accumulate=True)
            sm_bwd(a3, dS3, mlm if kmask else None, mrow if kmask else None)
            # similarities: s1 = scale q kl^T, s2 = scale ql kl^T, s3 = scale ql k^T
            K.gemm(dS1, kl, out=dq, alpha=scale, mma=mma)
            K.gemm(tr(dS3), ql, out=dk, alpha=scale, mma=mma)
            K.gemm(tr(dS1), q, out=dkl, alpha=scale, mma=mma)
            K.gemm(dS3, k, out=dql, alpha=scale, mma=mma)
        if tile_side:
            torch.cuda.current_stream().wait_stream(side)
        elif side is not None:
            torch.cuda.current_stream().wait_stream(side)
            K.shared_chip = False
            del work
        else:
            dS2 = pinv_backward_tile(a2, saved, st, dZ, getattr(ctx, "a2b", None)) if ctx.tile else pinv_backward(a2, saved, st, dZ, pm, sd)
            sm_bwd(a2, dS2, mlm if kmask else None, mlm if kmask else None)
        if dlm2 is None:
            K.gemm(tr(dS2), ql, out=dkl, alpha=scale, accumulate=True, mma=pio)
            K.gemm(dS2, kl, out=dql, alpha=scale, accumulate=True, mma=pio)
        dres = _gret(res_w, dres, dres_sunk)
        dres = None if dres is None else dres.view_as(res_w)
        if de is not None and (kmask is None or (lscale is None and dlm2 is not None)):
            # the merge of the two f32 partial sums is also the cast, written as rows [dq_l | dk_l | 0] behind the sequence rows.
            # Nothing but the landmark rows' own products needs it: NormQkvLmFn.backward runs it (and the landmark rows' data gradient)
            # on a parallel branch beside the data gradient of the sequence rows instead of in front of it
            merge = lambda: K.lm_merge(dlm, dlm2, de[Bn * n_p:], D3 - 2 * D)      # noqa: E731
            if _LM_MERGE_LATE:
                _pending_lm_merge[dlm_out.data_ptr()] = (de.data_ptr(), merge)
            else:
                merge()
            return dqkv, dres, None, None, None, None, None, None, dlm_out
        if dlm2 is not None:
            dlm = K.add(dlm, dlm2, out_dtype=A)
        if kmask is not None and lscale is not None:
            dlm = K.row_scale(dlm, lscale)
        if ctx.lm_ext:         # the landmarks came from the caller: their gradient goes back to it (no scatter into dqkv)
            if de is not None:
                dlm_out.copy_(dlm)
                de[Bn * n_p:, 2 * D:].zero_()
                return dqkv, dres, None, None, None, None, None, None, dlm_out
            return dqkv, dres, None, None, None, None, None, None, dlm
        K.landmark_bwd(K.cast(dlm, A), dqkv, l)
        return dqkv, dres, None, None, None, None, None, None, None


class RowScaleFn(Function):
    """y[..., r, :] = x[..., r, :] * s[..., r] with a constant per-row factor (key-padding mask: rows zeroed in front of the
    bias-free to_qkv, which is what the package's `q, k, v = map(lambda t: t * mask[..., None], (q, k, v))` amounts to)."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.save_for_backward(s)
        return K.row_scale(x.contiguous(), s)

    @staticmethod
    def backward(ctx, dy):
        (s,) = ctx.saved_tensors
        return K.row_scale(dy.contiguous(), s), None


# ------------------------------------------------------------------ timeline probes (profiling aid, MIRROR_PROBE=1)
# Where the branches of a step really start and end on the device, without a tracer attached (a tracer re-times the queues):
# one-thread launches that store the device wall clock, named by call site; tools/exp/probe_timeline.py prints them.
_PROBE = os.environ.get("MIRROR_PROBE", "0") != "0"
_probe_names: List[str] = []
_probe_buf: Optional[torch.Tensor] = None


def probe(name: str) -> None:
    global _probe_buf
    if not _PROBE:
        return
    if _probe_buf is None:
        _probe_buf = torch.zeros(256, dtype=torch.int64, device="cuda")
    if name not in _probe_names:
        _probe_names.append(name)
    K.timestamp(_probe_buf[_probe_names.index(name):_probe_names.index(name) + 1])


def probe_read() -> dict:
    torch.cuda.synchronize()
    v = _probe_buf[:len(_probe_names)].tolist()
    return dict(zip(_probe_names, v))


class ProbeFn(Function):
    """identity; stamps `name`.fwd in the forward and `name`.bwd when the gradient of this point is complete"""

    @staticmethod
    def forward(ctx, x, name):
        ctx.name = name
        probe(name + ".fwd")
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        probe(ctx.name + ".bwd")
        return g, None


def probe_point(x: torch.Tensor, name: str) -> torch.Tensor:
    if not (_PROBE and x.requires_grad):
        return x
    y = ProbeFn.apply(x, name)
    for attr in ("_bf16", "_relu_slot", "_drop_site", "_fan_slot"):      # hand-over slots that travel on the tensor
        if getattr(x, attr, None) is not None:
            setattr(y, attr, getattr(x, attr))
    return y


# ------------------------------------------------------------------ masking
def rank_mask(noise: torch.Tensor, len_keep: int) -> torch.Tensor:
    return K.rank_mask(noise, len_keep)


class MaskApplyFn(Function):
    """y[b,t] = mask_token where mask[b,t-first] else x[b,t];  y += pos  (models/mirror.py:636-643 + :693; :521-527 + :549).
    y is f32 (it starts a residual stream); dx comes back in x's dtype."""

    @staticmethod
    def forward(ctx, x, mask, token, pos, first, token_scalar):
        x = x.contiguous()
        if x.dim() == 2:           # RNA: channels are the masked axis -> [B, T=D, 1]
            Bn, T, D = x.shape[0], x.shape[1], 1
        else:
            Bn, T, D = x.shape
        y = torch.empty(x.shape, device=x.device, dtype=f32)
        K.mask_apply_fwd(x, mask, token.detach().reshape(-1).contiguous(), pos.detach().reshape(-1).contiguous(),
                         Bn, T, D, first, token_scalar, out=y)
        ctx.save_for_backward(mask)
        ctx.geom = (Bn, T, D, first, token_scalar, token.shape, pos.shape, x.dtype)
        ctx.params = (token, pos)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        Bn, T, D, first, token_scalar, tshape, pshape, xdt = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty(dy.shape, device=dy.device, dtype=xdt)
        token, pos = ctx.params
        dtok, s_tok = _gbuf_n(token, (1 if token_scalar else D,))
        dpos, s_pos = _gbuf_n(pos, (T * D,))
        K.mask_apply_bwd(dy, mask, dtok, dpos, Bn, T, D, first, token_scalar, out=dx)
        dtok, dpos = _gret(token, dtok, s_tok), _gret(pos, dpos, s_pos)
        return dx, None, None if dtok is None else dtok.reshape(tshape), None if dpos is None else dpos.reshape(pshape), None, None


# ------------------------------------------------------------------ encoder-output fan-out
class _FanToken:
    """Hand-over slot between the consumer of E[:, 1:] (the masked MSE) and EncFanoutFn.backward."""
    __slots__ = ("grad",)

    def __init__(self):
        self.grad = None       # (tensor [B, T-1, D], alpha): the target's gradient is alpha * tensor


class EncFanoutFn(Function):
    """The WSI encoder output E [B, T, D] feeds three consumers (models/mirror.py:684-700, :833): the retention decoder
    (all of E), the retention target (rows 1..) and the cls row.  autograd would sum three [B, T, D] f32 gradients with
    zero-filled slice backwards (~1.3 GB of traffic); this node builds dE in one pass (mh_fanout_bwd), and the masked
    MSE hands its target gradient over as -dpred instead of materialising it."""

    @staticmethod
    def forward(ctx, E, tok, slot=None):
        ctx.set_materialize_grads(False)
        ctx.tok, ctx.shape, ctx.slot = tok, tuple(E.shape), slot
        return E.view_as(E), E[:, 1:], E[:, 0]

    @staticmethod
    def backward(ctx, g_full, g_tgt, g_cls):
        Bn, T, D = ctx.shape
        x, alpha = (None, 0.0) if ctx.tok.grad is None else ctx.tok.grad
        ctx.tok.grad = None
        late = None
        if g_tgt is not None:                  # a consumer that returned a materialised gradient for the target
            if x is None:
                x, alpha = g_tgt.contiguous(), 1.0
            else:
                late = g_tgt
        gf = None if g_full is None else g_full.contiguous().float()
        c = None if g_cls is None else g_cls.contiguous().float()
        if (ctx.slot is not None and _FAN_IN_LN_BWD and late is None and gf is not None and x is not None and x.dtype == bf16
                and x.is_contiguous() and ctx.slot.extra is None):
            # E is a LayerNorm's output: its backward adds the other two gradients while it reads this one (no [B, T, D] sum tensor)
            ctx.slot.extra = (x, alpha, c)
            return gf, None, None
        dE = K.fanout_bwd(gf, x, alpha, c, Bn, T, D)
        if late is not None:
            dE[:, 1:] += late
        return dE, None, None


def enc_fanout(E: torch.Tensor):
    """(E, E[:, 1:], E[:, 0]) with the gradients of the three summed by one kernel."""
    if not (torch.is_grad_enabled() and E.requires_grad and E.dim() == 3 and E.is_contiguous() and E.dtype == f32):
        return E, E[:, 1:], E[:, 0]
    tok = _FanToken()
    full, tgt, cls = EncFanoutFn.apply(E, tok, getattr(E, "_fan_slot", None))
    tgt._fan_token = tok
    if getattr(E, "_bf16", None) is not None:
        full._bf16 = E._bf16          # layer_norm(..., bf16_copy=True): the decoder's projection reads this copy
    return full, tgt, cls


# ------------------------------------------------------------------ small heads
class L2NormRowFn(Function):
    """F.normalize(x, dim=-1)[:, row] for x [B, T, D] (or [B, D]); only that row is touched."""

    @staticmethod
    def forward(ctx, x, eps, out_dtype):
        if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= x.shape[1]:
            (Bn, D), rs = x.shape, x.stride(0)      # rows of a larger buffer (the cls row of every slide): read in place
        else:
            x = x.contiguous()
            if x.dim() == 3:
                Bn, T, D = x.shape
                rs = T * D
            else:
                (Bn, D), rs = x.shape, x.shape[1]
        y, nrm = K.l2norm_fwd(x, Bn, D, rs, eps, out_dtype)
        ctx.save_for_backward(y, nrm)
        ctx.geom = (tuple(x.shape), x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, nrm = ctx.saved_tensors
        shape, xdt = ctx.geom
        dy = dy.contiguous()
        if dy.dtype != y.dtype:
            dy = K.cast(dy, y.dtype)
        if len(shape) == 2:          # every row is written: no fill
            dx, rs = torch.empty(shape, device=dy.device, dtype=xdt), shape[1]
        else:
            dx = zeros(shape, dy.device) if xdt == f32 else torch.zeros(shape, device=dy.device, dtype=xdt)
            rs = shape[1] * shape[2]
        K.l2norm_bwd(y, nrm, dy, dx, shape[0], shape[-1], rs, accumulate=False)
        return dx, None, None


class HeadAttnFn(Function):
    """Attention.forward on [B, D] inputs (models/mirror.py:77-99): softmax over the heads axis + permutation."""

    @staticmethod
    def forward(ctx, qkv, H):
        qkv = qkv.contiguous()
        out, attn = K.headattn_fwd(qkv, H)
        ctx.save_for_backward(qkv, attn)
        ctx.H = H
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, attn = ctx.saved_tensors
        dout = dout.contiguous()
        if dout.dtype != qkv.dtype:
            dout = K.cast(dout, qkv.dtype)
        return K.headattn_bwd(qkv, attn, dout, ctx.H), None


_RNA_FUSED = True      # 0 = the composed Block


class RnaBlockFn(Function):
    """Block.forward (models/mirror.py:149-152) on [B, D] rows as ONE C-ABI call per direction (mh_rna_block_fwd / _bwd):
    LayerNorm, bias, GELU, the three dropouts and both residual adds ride in the GEMM kernels' prologues / epilogues."""

    NAMES = ("g1", "be1", "w_qkv", "b_qkv", "w_proj", "b_proj", "g2", "be2", "w_fc1", "b_fc1", "w_fc2", "b_fc2")

    @staticmethod
    def forward(ctx, x, g1, be1, wqkv, bqkv, wproj, bproj, g2, be2, w1, b1, w2, b2, H, eps, p, prec):
        masters = dict(zip(RnaBlockFn.NAMES, (g1, be1, wqkv, bqkv, wproj, bproj, g2, be2, w1, b1, w2, b2)))
        params = {}
        for k, v in masters.items():
            if v is None:
                params[k] = None
            elif k.startswith("w_"):
                params[k] = shadow(v, prec).contiguous()
            else:
                params[k] = v.detach().reshape(-1)
        B, D = x.shape
        Hh = w1.shape[0]
        st = _dropout_state
        ctx.drop = (float(p), st["seed"], st["offset"], st["base"]) if p > 0.0 else (0.0, 0, 0, None)
        if p > 0.0:
            q4 = lambda n: (n + 3) // 4 * 4  # noqa: E731
            o0 = st["offset"]
            st["offset"] += 2 * q4(B * D) + q4(B * Hh)
            # the kernels' three sites, in their offset order: proj output, fc1 activation, fc2 output
            _tap((B, D), p, st["seed"], o0, False)
            _tap((B, Hh), p, st["seed"], o0 + q4(B * D), False)
            _tap((B, D), p, st["seed"], o0 + q4(B * D) + q4(B * Hh), False)
        y, saved = K.rna_block_fwd(x, params, H, eps, *ctx.drop)
        ctx.masters, ctx.saved, ctx.cfg = masters, saved, (H, eps, prec)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        H, eps, prec = ctx.cfg
        masters = ctx.masters
        params, params_t, grads, rets = {}, {}, {}, {}
        for k, v in masters.items():
            if v is None:
                params[k] = None
                continue
            if k.startswith("w_"):
                params[k] = shadow(v, prec).contiguous()
                params_t["wt_" + k[2:]] = shadow_t(v, prec)
                gk = "dw_" + k[2:]
            else:
                params[k] = v.detach().reshape(-1)
                gk = {"g1": "dg1", "be1": "dbe1", "g2": "dg2", "be2": "dbe2"}.get(k, "d" + k)
            buf, sunk = _gbuf(v, tuple(v.shape))
            grads[gk] = buf
            rets[k] = (v, buf, sunk)
        dx = K.rna_block_bwd(x, dy.contiguous().float(), params, params_t, grads, ctx.saved, H, eps, *ctx.drop)
        out = [dx]
        for k in RnaBlockFn.NAMES:
            if k in rets:
                v, buf, sunk = rets[k]
                out.append(_gret(v, buf, sunk))
            else:
                out.append(None)
        return tuple(out) + (None, None, None, None)


def rna_block(x, blk, prec: Precision, training: bool):
    """x [B, D] f32 through `blk` (a models.mirror.Block): fused when the geometry fits, None otherwise (the caller then
    runs the composed ops)."""
    a, m = blk.attn, blk.mlp
    p = float(a.proj_drop) if training else 0.0
    if not (_RNA_FUSED and prec.act == bf16 and x.dtype == f32 and K.rna_block_ok(x, x.shape[-1], m.fc1.weight.shape[0], a.num_heads)
            and float(m.drop) == float(a.proj_drop) and blk.norm1.eps == blk.norm2.eps):
        return None
    return RnaBlockFn.apply(x.contiguous(), blk.norm1.weight, blk.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                            blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                            a.num_heads, float(blk.norm1.eps), p, prec)


class ExpFn(Function):
    """x.exp() for a small f32 parameter (`logit_scale.exp()`, models/mirror.py:911): one launch each way, the gradient summed
    straight into the parameter's arena slot (torch's exp costs a mul in the backward and an add in AccumulateGrad)."""

    @staticmethod
    def forward(ctx, x):
        y = K.exp_fwd(x.detach().contiguous())
        ctx.save_for_backward(y)
        ctx.x = x
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx, sunk = _gbuf_n(ctx.x, tuple(y.shape))
        K.exp_bwd(dy.contiguous().float(), y, dx, accumulate=True)      # _gbuf_n's own buffer starts at zero
        return _gret(ctx.x, dx, sunk)


def exp(x: torch.Tensor) -> torch.Tensor:
    return ExpFn.apply(x) if (x.is_cuda and x.dtype == f32) else x.exp()


class ReparamFn(Function):
    """z = mu + eps * exp(0.5 * logstd) (models/mirror.py:830-833).  Returns (z, mu, logstd): the caller hands THESE mu / logstd on
    (to the KL term), so what comes back for them arrives at this node and is summed inside its one backward launch — mu and logstd
    keep a single consumer and autograd has no `grad += grad` launch to add (two per call)."""

    @staticmethod
    def forward(ctx, mu, logstd, eps):
        ctx.set_materialize_grads(False)
        mu_c, ls_c, eps = mu.contiguous(), logstd.contiguous(), eps.contiguous()
        ctx.save_for_backward(ls_c, eps)
        return K.reparam_fwd(mu_c, ls_c, eps), mu, logstd

    @staticmethod
    def backward(ctx, dz, dmu_o, dls_o):
        logstd, eps = ctx.saved_tensors
        if dz is None:
            return dmu_o, dls_o, None
        f = lambda g: None if g is None else (g if g.dtype == f32 else g.float()).contiguous()
        dmu, dls = K.reparam_bwd(logstd, eps, dz.contiguous(), f(dmu_o), f(dls_o))
        return dmu, dls, None


# ------------------------------------------------------------------ losses
class CERowsFn(Function):
    """sum_r coef * CE(scale * scale_mul * G[r, :], label_off + r) as a [1] tensor (or per-row losses)."""

    @staticmethod
    def forward(ctx, G, scale, scale_mul, label_off, coef, per_row):
        G = G.contiguous()
        R = G.shape[0]
        out = zeros((1,), G.device)
        rows = torch.empty((R,), device=G.device, dtype=f32) if per_row else None
        sc = None if scale is None else scale.detach().reshape(1)
        lse = K.ce_rows_fwd(G, sc, scale_mul, label_off, coef, out, rows)
        ctx.save_for_backward(G, sc, lse)
        ctx.cfg = (scale_mul, label_off, coef, per_row, scale is not None and scale.requires_grad, None if scale is None else scale.shape)
        return rows if per_row else out

    @staticmethod
    def backward(ctx, g):
        G, sc, lse = ctx.saved_tensors
        scale_mul, label_off, coef, per_row, want_ds, sshape = ctx.cfg
        g = g.contiguous().float()
        ds = zeros((1,), G.device) if sc is not None else None
        dG = K.ce_rows_bwd(G, sc, scale_mul, lse, g, per_row, coef, ds, label_off)
        return dG, (ds.reshape(sshape) if want_ds else None), None, None, None, None


class MaskedMSEFn(Function):
    """sum(mask * mean_D (p - t)^2) / sum(mask)  (losses/mirror_loss.py:98-103); gradients flow to BOTH p and t.
    The target may be a row window of a larger buffer (encoder_output[:, 1:]) and is read in place."""

    @staticmethod
    def forward(ctx, pred, tgt, mask, D, tok=None):
        pred = pred.contiguous()
        if not (tgt.is_contiguous() or (tgt.dim() == 3 and tgt.stride(2) == 1 and tgt.stride(1) == D)):
            tgt = tgt.contiguous()
        acc = _sq_of(pred, tgt, mask)        # filled by the projection that produced pred (HeadSqErrFn)
        ctx.cs_tok = _cs_of(pred, acc)
        mask = mask.contiguous().float()
        rows = pred.numel() // D
        if acc is None:
            acc = zeros((2,), pred.device)
            K.mse_masked_fwd(pred, tgt, mask, acc, rows, D)
        ctx.save_for_backward(pred, tgt, mask, acc)
        ctx.D, ctx.tok = D, tok
        return _div(acc)

    @staticmethod
    def backward(ctx, g):
        pred, tgt, mask, acc = ctx.saved_tensors
        D, tok = ctx.D, ctx.tok
        dp = torch.empty_like(pred)
        hand_over = tok is not None and ctx.needs_input_grad[1]
        dtg = torch.empty(tgt.shape, device=tgt.device, dtype=tgt.dtype) if (ctx.needs_input_grad[1] and not hand_over) else None
        K.mse_masked_bwd(pred, tgt, mask, acc, g.contiguous().float().reshape(1), dp, dtg, pred.numel() // D, D,
                         colsum_ws=_mse_bwd_cs(ctx.cs_tok, pred, tgt, dp, D))
        if hand_over:
            tok.grad = (dp, -1.0)          # EncFanoutFn.backward folds -dpred into the encoder-output gradient
        return dp, dtg, None, None, None


def masked_mse(pred, tgt, mask, D):
    return MaskedMSEFn.apply(pred, tgt, mask, D, getattr(tgt, "_fan_token", None))


def _div(acc: torch.Tensor) -> torch.Tensor:
    # one scalar division: the only arithmetic left to torch in the loss (0-d glue, no host sync)
    return (acc[0] / acc[1]).reshape(())


class WeightedSumFn(Function):
    """total = sum_i w_i * term_i (losses/mirror_loss.py:121-127) in one launch; the backward hands w_i * g to every term."""

    @staticmethod
    def forward(ctx, weights, *terms):
        ctx.weights, ctx.shapes = tuple(float(w) for w in weights), [t.shape for t in terms]
        return K.weighted_sum([t.detach().float().contiguous() for t in terms], ctx.weights)

    @staticmethod
    def backward(ctx, g):
        d = K.weighted_sum_bwd(g.contiguous().float().reshape(1), ctx.weights)
        return (None,) + tuple(d[i].reshape(sh) for i, sh in enumerate(ctx.shapes))


class StyleKLFn(Function):
    """coef * sum(exp(ls) + mu^2 - 1 - ls)  (losses/mirror_loss.py:105-112)."""

    @staticmethod
    def forward(ctx, mu, ls, coef):
        mu, ls = mu.contiguous().float(), ls.contiguous().float()
        out = zeros((1,), mu.device)
        K.kl_fwd(mu, ls, out, coef)
        ctx.save_for_backward(mu, ls)
        ctx.coef = coef
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        mu, ls = ctx.saved_tensors
        dmu, dls = K.kl_bwd(mu, ls, g.contiguous().float().reshape(1), ctx.coef)
        return dmu, dls, None


class SymKLFn(Function):
    """coef * sum_b sum_k (p_r - p_w)(log p_r - log p_w)  (losses/mirror_loss.py:114-119)."""

    @staticmethod
    def forward(ctx, w, r, coef):
        w, r = w.contiguous().float(), r.contiguous().float()
        out = zeros((1,), w.device)
        K.symkl_fwd(w, r, out, coef)
        ctx.save_for_backward(w, r)
        ctx.coef = coef
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        w, r = ctx.saved_tensors
        dw, dr = K.symkl_bwd(w, r, g.contiguous().float().reshape(1), ctx.coef)
        return dw, dr, None


_LOSS_FUSED = True      # MIRRORLoss as two launches + the WSI MSE


def loss_terms_fusable(align, rna, style, scores) -> bool:
    """MIRRORLoss's small terms run as mh_loss_terms_fwd / _bwd when every operand is an f32 device tensor, the RNA
    mask is data and, for a rank-local alignment term (align = (wsi, rna, scale); None = computed by the caller
    over the gathered batch), the batch fits the kernel's alignment block."""
    rp, rt, rm = rna
    ts = (rp, rt) + tuple(style) + tuple(scores) + (tuple(align) if align is not None else ())
    if not _LOSS_FUSED or any((not t.is_cuda) or t.dtype != f32 for t in ts) or not rm.is_cuda:
        return False
    if rm.requires_grad or rp.numel() == 0 or rp.shape != rt.shape or rm.numel() != rp.numel():
        return False
    wmu, wls, rmu, rls = style
    if wmu.dim() != 2 or rmu.dim() != 2 or wmu.shape != wls.shape or rmu.shape != rls.shape or wmu.numel() == 0 or rmu.numel() == 0:
        return False
    if scores[0].dim() != 2 or scores[0].shape != scores[1].shape or scores[0].numel() == 0:
        return False
    if align is not None:
        wa, ra, sc = align
        if wa.dim() != 2 or wa.shape != ra.shape or sc.numel() != 1 or not K.loss_terms_ok(wa.shape[0], wa.shape[1]):
            return False
    return True


class MirrorLossTermsFn(Function):
    """MIRRORLoss (losses/mirror_loss.py:74-135) as the WSI retention MSE + ONE launch for everything else, each way.
    Returns (total, alignment, wsi retention, rna retention, style, cluster); the usual backward arrives through `total`
    alone (gradients of single terms are honoured too, through a small torch-side vector).  wa / ra / scale None: the
    alignment term is `align_ext`, computed by the caller over the gathered batch."""

    @staticmethod
    def forward(ctx, weights, wa, ra, scale, align_ext, wpred, wtgt, wmask, Dw, tok, rp, rt, rm, wmu, wls, rmu, rls, wsc, rsc):
        ctx.set_materialize_grads(False)
        dev = rp.device
        acc = _sq_of(wpred, wtgt, wmask)     # filled by the projection that produced wpred (HeadSqErrFn)
        ctx.cs_tok = _cs_of(wpred, acc)
        wpred = wpred.contiguous()
        if not (wtgt.is_contiguous() or (wtgt.dim() == 3 and wtgt.stride(2) == 1 and wtgt.stride(1) == Dw)):
            wtgt = wtgt.contiguous()
        wmask = wmask.contiguous().float()
        if acc is None:
            acc = zeros((2,), dev)
            K.mse_masked_fwd(wpred, wtgt, wmask, acc, wpred.numel() // Dw, Dw)
        local = wa is not None
        t = dict(rna_pred=rp.contiguous(), rna_tgt=rt.contiguous(), rna_mask=rm.contiguous().float(), w_mu=wmu.contiguous(),
                 w_logstd=wls.contiguous(), r_mu=rmu.contiguous(), r_logstd=rls.contiguous(), w_score=wsc.contiguous(),
                 r_score=rsc.contiguous(), wsi_acc=acc, scratch=zeros((8,), dev), out=torch.empty((8,), device=dev, dtype=f32))
        if local:
            Bq = wa.shape[0]
            t.update(wsi_emb=wa.contiguous(), rna_emb=ra.contiguous(), logit_scale=scale.detach().reshape(1).contiguous(),
                     save=torch.empty((Bq * Bq + 2 * Bq,), device=dev, dtype=f32))
        elif align_ext is not None:
            t["align_ext"] = align_ext.detach().float().reshape(1).contiguous()
        K.loss_terms_fwd(weights, t)
        ctx.t, ctx.weights, ctx.big = t, tuple(float(w) for w in weights), (wpred, wtgt, wmask, acc, Dw, tok)
        ctx.shapes = (None if scale is None else scale.shape, None if align_ext is None else align_ext.shape)
        out = t["out"]
        return tuple(out[i].reshape(()) for i in range(6))

    @staticmethod
    def backward(ctx, g_total, g_align, g_wsi, g_rna, g_style, g_clu):
        t, weights = dict(ctx.t), ctx.weights
        wpred, wtgt, wmask, acc, Dw, tok = ctx.big
        dev = wpred.device
        extra = (g_align, g_wsi, g_rna, g_style, g_style, g_clu)
        if any(g is not None for g in extra):        # a single term was backpropagated through as well: rare, torch-side glue
            t["g_terms"] = torch.stack([torch.zeros((), device=dev) if g is None else g.float().reshape(()) for g in extra]).contiguous()
        if g_total is not None:
            t["g_total"] = g_total.contiguous().float().reshape(1)
        local = "wsi_emb" in t
        names = ["rna_pred", "w_mu", "w_logstd", "r_mu", "r_logstd", "w_score", "r_score"] + (["wsi_emb", "rna_emb"] if local else [])
        for n in names:
            t["d_" + n] = torch.empty_like(t[n])
        if ctx.needs_input_grad[11]:
            t["d_rna_tgt"] = torch.empty_like(t["rna_tgt"])
        if local:
            t["d_logit_scale"] = torch.empty((1,), device=dev, dtype=f32)
        else:
            t["d_align_ext"] = torch.empty((1,), device=dev, dtype=f32)
        K.loss_terms_bwd(weights, t)
        # WSI retention: its own HBM-bound kernel; upstream = weight * g_total (+ the single-term gradient)
        dp = torch.empty_like(wpred)
        hand_over = tok is not None and ctx.needs_input_grad[6]
        dtg = torch.empty(wtgt.shape, device=dev, dtype=wtgt.dtype) if (ctx.needs_input_grad[6] and not hand_over) else None
        if "g_terms" in t or g_total is None:
            gb = t["g_terms"][1:2] + (weights[1] * t["g_total"] if g_total is not None else 0.0)
            K.mse_masked_bwd(wpred, wtgt, wmask, acc, gb.contiguous(), dp, dtg, wpred.numel() // Dw, Dw,
                             colsum_ws=_mse_bwd_cs(ctx.cs_tok, wpred, wtgt, dp, Dw))
        else:
            K.mse_masked_bwd(wpred, wtgt, wmask, acc, t["g_total"], dp, dtg, wpred.numel() // Dw, Dw, gmul=weights[1],
                             colsum_ws=_mse_bwd_cs(ctx.cs_tok, wpred, wtgt, dp, Dw))
        if hand_over:
            tok.grad = (dp, -1.0)          # EncFanoutFn.backward folds -dpred into the encoder-output gradient
        ssh, ash = ctx.shapes
        return (None, t.get("d_wsi_emb"), t.get("d_rna_emb"), t["d_logit_scale"].reshape(ssh) if local else None,
                t["d_align_ext"].reshape(ash) if (not local and ash is not None) else None, dp, dtg, None, None, None,
                t["d_rna_pred"], t.get("d_rna_tgt"), None, t["d_w_mu"], t["d_w_logstd"], t["d_r_mu"], t["d_r_logstd"], t["d_w_score"], t["d_r_score"])


class MatmulNTFn(Function):
    """G = a @ b^T in f32 (similarity logits of ClipLoss / InfoNCE, losses/mirror_loss.py:39-40)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous().float(), b.contiguous().float()
        ctx.save_for_backward(a, b)
        Kd = a.shape[1]
        if a.shape[0] <= 128 and b.shape[0] <= 128 and Kd >= 256:
            # a [B, D] x [D, B] product is ONE workgroup walking D in exact-f32 steps of 16 (30 us at D = 512): split K
            out = zeros((a.shape[0], b.shape[0]), a.device)
            return K.gemm(a, b.t(), out=out, accumulate=True, split_k=max(2, min(16, Kd // 64)), mma=MH_F32)
        return K.gemm(a, b.t(), mma=MH_F32)

    @staticmethod
    def backward(ctx, dG):
        a, b = ctx.saved_tensors
        dG = dG.contiguous()
        return K.gemm(dG, b, mma=MH_F32), K.gemm(dG.t(), a, mma=MH_F32)
