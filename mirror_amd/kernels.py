"""Tensor-level launchers for the C-ABI kernels (no autograd here; see functional.py).

Every function takes HIP tensors, checks shapes on the host (a faulting kernel can take the whole
node down) and launches on torch's current stream.  There is no CPU / PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional


import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, MH_BF16, MH_F32, GemmDesc, MirrorHipError

_DT = {torch.float32: MH_F32, torch.bfloat16: MH_BF16}


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise MirrorHipError(f"unsupported dtype {t.dtype} (f32 / bf16 only)") from None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """hipStream_t of torch's current stream.  `torch.cuda.current_stream()` builds a Stream object (~9 us: ~4 ms of host
    time per step over ~450 launches); the raw accessor is ~0.3 us."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MirrorHipError("mirror_amd kernels need HIP device tensors: there is no CPU fallback path")


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _contig(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_contiguous():
        raise MirrorHipError(f"{name} must be contiguous, got strides {t.stride()} for shape {tuple(t.shape)}")
    return t


# ----------------------------------------------------------------------------- GEMM
def _mat(t: torch.Tensor):
    """[.., R, C] view (<= 2 batch dims) -> (t4, rowmajor, ld, s1, s2)."""
    if t.dim() < 2 or t.dim() > 4:
        raise MirrorHipError(f"gemm operand must have 2..4 dims, got {tuple(t.shape)}")
    while t.dim() < 4:
        t = t.unsqueeze(0)
    b1, b2, r, c = t.shape
    st = t.stride()
    if st[3] == 1 or c == 1:
        rowmajor, ld = True, (st[2] if r > 1 else max(c, 1))
    elif st[2] == 1 or r == 1:
        rowmajor, ld = False, (st[3] if c > 1 else max(r, 1))
    else:
        raise MirrorHipError(f"gemm operand needs a unit stride in one of its last two dims: {st}")
    return t, rowmajor, ld, (st[0] if b1 > 1 else 0), (st[1] if b2 > 1 else 0)


def gemm(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, *, alpha: float = 1.0,
         diag: float = 0.0, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, accumulate: bool = False,
         split_k: int = 1, mma: int = MH_F32, out_dtype: Optional[torch.dtype] = None,
         R: Optional[torch.Tensor] = None, rcoef: float = 0.0, c2: Optional[torch.Tensor] = None,
         kseg: Optional[tuple] = None, softmax: bool = False,
         softmax_bwd_of: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[..] (+)= act(alpha * a @ b + diag*I + bias + rcoef*R) with a [..,M,K], b [..,K,N] given as (possibly
    transposed / strided / broadcast) views; <= 2 leading batch dims.  c2: optional bf16 tensor shaped and strided like `out`
    that receives a copy of the final result (192 x 384 tile kernel only, see gemm_tile_ok).  kseg = (S, a_stride, b_stride):
    the product becomes sum_s a_s @ b_s over S operand pairs that start a_stride / b_stride elements apart (gemm_ksum).
    softmax=True: out = softmax(alpha * a @ b, dim=-1) as bf16, the rows normalised in the epilogue (gemm_softmax_ok shapes);
    softmax_bwd_of=P (bf16 probabilities, shaped and strided like out): out = P * (dP - sum(P * dP, -1)) with dP = alpha * a @ b."""
    _chk(a, b, out, bias)
    nd = max(a.dim(), b.dim())
    a4, a_rm, lda, sa1, sa2 = _mat(a)
    b4, b_rm, ldb, sb1, sb2 = _mat(b)
    M, K = a4.shape[2], a4.shape[3]
    K2, N = b4.shape[2], b4.shape[3]
    if K != K2:
        raise MirrorHipError(f"gemm: inner dims differ: {tuple(a.shape)} @ {tuple(b.shape)}")
    B1 = max(a4.shape[0], b4.shape[0])
    B2 = max(a4.shape[1], b4.shape[1])
    for t4 in (a4, b4):
        if t4.shape[0] not in (1, B1) or t4.shape[1] not in (1, B2):
            raise MirrorHipError(f"gemm: batch dims do not broadcast: {tuple(a.shape)} @ {tuple(b.shape)}")
    if a4.dtype != b4.dtype and mma == MH_F32:
        raise MirrorHipError(f"gemm: f32 MMA needs f32 operands, got {a4.dtype} x {b4.dtype}")
    if out is None:
        if accumulate:
            raise MirrorHipError("gemm: accumulate needs an explicit output")
        out = torch.empty((B1, B2, M, N), device=a.device, dtype=out_dtype or a.dtype)
        o4 = out
        out = out.reshape(out.shape[4 - nd:])
    else:
        o4 = out
        while o4.dim() < 4:
            o4 = o4.unsqueeze(0)
        if tuple(o4.shape) != (B1, B2, M, N):
            raise MirrorHipError(f"gemm: output shape {tuple(out.shape)} != {(B1, B2, M, N)}")
    so = o4.stride()
    if not (so[3] == 1 or N == 1):
        raise MirrorHipError("gemm: output must have unit stride in its last dim")
    if mma == MH_F32 and (a4.dtype != torch.float32 or b4.dtype != torch.float32 or o4.dtype != torch.float32):
        raise MirrorHipError("gemm: f32 MMA needs f32 operands and output")
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != N or not bias.is_contiguous():
            raise MirrorHipError("gemm: bias must be contiguous f32 [N]")
    d = GemmDesc()
    d.A, d.B, d.C, d.bias = a4.data_ptr(), b4.data_ptr(), o4.data_ptr(), _p(bias)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, (so[2] if M > 1 else max(N, 1))
    d.a_kc, d.b_kc = int(a_rm), int(not b_rm)
    d.dtA, d.dtB, d.dtC, d.mma = dt(a4), dt(b4), dt(o4), mma
    d.batch1, d.batch2 = B1, B2
    d.sA1, d.sA2, d.sB1, d.sB2 = sa1, sa2, sb1, sb2
    d.sC1, d.sC2 = (so[0] if B1 > 1 else 0), (so[1] if B2 > 1 else 0)
    d.alpha, d.diag, d.act, d.accumulate, d.split_k = alpha, diag, act, int(accumulate), max(1, int(split_k))
    if R is not None:
        _chk(R)
        tile = gemm_tile_ok(M, N, K, a4.dtype, b4.dtype) and bias is None and act == ACT_NONE and d.split_k == 1
        if R.numel() != o4.numel() or tuple(R.stride()) != tuple(out.stride()) or not (
                R.dtype == o4.dtype or (tile and R.dtype in (torch.bfloat16, torch.float32))):
            raise MirrorHipError("gemm: R must have the output's dtype (or bf16 / f32 on the 192 x 384 tile kernel), shape and strides")
        d.R, d.rcoef = R.data_ptr(), rcoef
        d.r_bf16 = 0 if R.dtype == o4.dtype else (1 if R.dtype == torch.bfloat16 else 2)
    if c2 is not None:
        _chk(c2)
        if c2.dtype != torch.bfloat16 or c2.numel() != o4.numel() or tuple(c2.stride()) != tuple(out.stride()):
            raise MirrorHipError("gemm: c2 must be a bf16 tensor with the output's shape and strides")
        d.C2 = c2.data_ptr()
    if softmax:
        if not (gemm_softmax_ok(M, N, K, a4.dtype, b4.dtype) and o4.dtype == torch.bfloat16 and bias is None and act == ACT_NONE
                and d.split_k == 1 and not accumulate and R is None and c2 is None and diag == 0.0):
            raise MirrorHipError("gemm: softmax=True needs bf16 operands and output, M % 192 == 0, N == 384, K % 8 == 0 and a plain product")
        d.row_softmax = 1
    if softmax_bwd_of is not None:
        P_ = softmax_bwd_of
        _chk(P_)
        if not (gemm_softmax_ok(M, N, K, a4.dtype, b4.dtype) and o4.dtype == torch.bfloat16 and P_.dtype == torch.bfloat16 and bias is None
                and act == ACT_NONE and d.split_k == 1 and not accumulate and R is None and c2 is None and diag == 0.0 and not softmax
                and P_.numel() == o4.numel() and tuple(P_.stride()) == tuple(out.stride())):
            raise MirrorHipError("gemm: softmax_bwd_of needs bf16 operands / output / probabilities (same layout as out), M % 192 == 0, N == 384")
        d.row_softmax, d.R = 2, P_.data_ptr()
    if kseg is not None and int(kseg[0]) > 1:
        if not (gemm_tile_ok(M, N, K, a4.dtype, b4.dtype) and bias is None and act == ACT_NONE and d.split_k == 1):
            raise MirrorHipError("gemm: a sum over operand pairs (kseg) needs the 192 x 384 tile kernel (see gemm_tile_ok)")
        d.k_segments, d.sA_seg, d.sB_seg = int(kseg[0]), int(kseg[1]), int(kseg[2])
    # reductions into one f32 C (split-K / a batch that broadcasts into C) on the large-tile kernel: give it room for plain
    # partial tiles + a fold pass (f32 atomics of a 64-way split cost more than the K loop).  `ws` stays alive until the call
    # is enqueued; the caching allocator keeps the block valid for stream-ordered use.
    wsb = int(_lib.load().mh_gemm_workspace_bytes(C.byref(d))) if (accumulate and _GEMM_WS and SPLITK_PARTIALS == "bf16") else 0
    if 0 < wsb <= (1 << 29):
        ws = torch.empty((wsb // 4,), device=a.device, dtype=torch.float32)
        d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
    d.shared_chip = int(shared_chip)
    prof = gemm_profiler
    if prof is None:
        _lib.call("mh_gemm", C.byref(d), stream=_stream())
    else:
        prof.launch(d, lambda: _lib.call("mh_gemm", C.byref(d), stream=_stream()))
    return out


# set (by functional.NystromCoreFn) while the half-chip pinv chain runs on its side stream: launches issued meanwhile must not
# use the persistent GEMM kernel (mh_gemm_desc.shared_chip)
shared_chip = False


EPI_DROPADD, EPI_MASKPOS, EPI_SQERR = 1, 2, 3
_EPI_ON = True       # the fused projection epilogues (mh_gemm_epi); test hook: False = composed projection + elementwise passes


def linear_fused_ok(x: torch.Tensor, w: torch.Tensor, window: Optional[tuple] = None) -> bool:
    """Shapes mh_gemm's fused epilogues take: bf16 activations x [B, T, K] (contiguous; optionally the row window
    [r0, r0 + R) of every batch), bf16 weight [N, K] contiguous, N % 256 == 0, K % 64 == 0, more than 256 rows in all."""
    if not (_EPI_ON and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 3 and x.is_contiguous() and w.dim() == 2
            and w.is_contiguous() and x.is_cuda):
        return False
    Bn, T, Kd = x.shape
    R = T if window is None else window[1]
    return (w.shape[1] == Kd and Kd % 64 == 0 and w.shape[0] % 256 == 0 and Bn * R > 256 and (window is None or R >= 256 or Bn == 1)
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


def linear_fused_tail(rows: int) -> int:
    """Flat rows that a fused launch leaves to the caller: a few rows past whole 256-row tiles (B x 4097 = 256 tiles + 16 rows) would
    cost every launch a third round of workgroups on 256 CUs for two almost empty tiles; the caller runs them through the composed
    ops (a [16, K] product and a 16-row elementwise launch)."""
    rem = rows % 256
    return rem if (rows > 512 and 0 < rem <= 64) else 0


def linear_fused(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, epi: "_lib.GemmEpi",
                 window: Optional[tuple] = None, m_rows: Optional[int] = None) -> torch.Tensor:
    """out [B * R, N] (flat rows, contiguous) = epilogue(x[:, r0:r0 + R] @ w^T + bias) on the 256 x 256-tile kernel with one of the
    fused epilogues of include/mirror_hip.h (mh_gemm_epi).  window = (r0, R) or None (all T rows).  The row windows of all batches
    are ONE flat problem (mh_gemm_desc.a_rows_per_batch): no ragged tile per slide, no tail launch."""
    _chk(x, w, out, bias)
    Bn, T, Kd = x.shape
    r0, R = (0, T) if window is None else window
    N = w.shape[0]
    if not out.is_contiguous() or out.numel() != Bn * R * N:
        raise MirrorHipError("linear_fused: out must be contiguous [B * R, N]")
    d = GemmDesc()
    d.A, d.B, d.C, d.bias = x.data_ptr() + r0 * Kd * 2, w.data_ptr(), out.data_ptr(), _p(bias)
    d.M, d.N, d.K = (Bn * R if m_rows is None else int(m_rows)), N, Kd      # m_rows: only the first m_rows flat rows (see linear_fused_tail)
    d.lda, d.ldb, d.ldc = Kd, Kd, N
    d.a_kc, d.b_kc = 1, 1
    d.dtA, d.dtB, d.dtC, d.mma = MH_BF16, MH_BF16, dt(out), MH_BF16
    d.batch1 = d.batch2 = 1
    d.alpha, d.split_k = 1.0, 1
    if R != T and Bn > 1:
        d.a_rows_per_batch, d.a_row_skip = R, T - R
    d.epi = C.pointer(epi)
    prof = gemm_profiler
    if prof is None:
        _lib.call("mh_gemm", C.byref(d), stream=_stream())
    else:
        prof.launch(d, lambda: _lib.call("mh_gemm", C.byref(d), stream=_stream()))
    return out


def gemm_rows_window(a3: torch.Tensor, b2: torch.Tensor, out3: torch.Tensor, r0: int, R: int, m_rows: Optional[int] = None,
                     a_r0: Optional[int] = None) -> None:
    """out3[:, r0:r0 + R] = a3[:, a_r0:a_r0 + R] @ b2 as ONE flat problem over the B * R real rows of buffers that hold T >= R rows per batch
    (mh_gemm_desc.a_rows_per_batch / c_rows_per_batch): [3P] NystromAttention's front-padded sequence — the pad rows are never multiplied.
    a3 [B, Ta, K] bf16 (rows K-contiguous; a_r0 defaults to r0, Ta == R: a plain contiguous operand), b2 a 2-D weight view [K, N] (W^T of
    an [N, K] row-major weight, or a row-major [K, N]), out3 [B, T, N] bf16 (may be a column slice of a wider buffer).  m_rows: only the
    first m_rows flat rows (whole 256-row tiles)."""
    _chk(a3, b2, out3)
    Bn, Ta, Kd = a3.shape
    T = out3.shape[1]
    N = b2.shape[1]
    bf = torch.bfloat16
    a_r0 = r0 if a_r0 is None else a_r0
    if not (a3.dtype == b2.dtype == out3.dtype == bf and a3.stride(2) == 1 and out3.stride(2) == 1 and a3.stride(0) == Ta * a3.stride(1)
            and out3.stride(0) == T * out3.stride(1) and tuple(out3.shape) == (Bn, T, N) and b2.shape[0] == Kd and 0 <= r0 and r0 + R <= T
            and 0 <= a_r0 and a_r0 + R <= Ta):
        raise MirrorHipError("gemm_rows_window: bad operands")
    d = GemmDesc()
    d.A, d.B, d.C, d.bias = a3.data_ptr() + a_r0 * a3.stride(1) * 2, b2.data_ptr(), out3.data_ptr() + r0 * out3.stride(1) * 2, None
    d.M, d.N, d.K = (Bn * R if m_rows is None else int(m_rows)), N, Kd
    d.lda, d.ldc = a3.stride(1), out3.stride(1)
    if b2.stride(0) == 1:            # W^T of a row-major [N, K] weight: contraction index contiguous
        d.ldb, d.b_kc = b2.stride(1), 1
    elif b2.stride(1) == 1:          # row-major [K, N]
        d.ldb, d.b_kc = b2.stride(0), 0
    else:
        raise MirrorHipError("gemm_rows_window: the weight view must have a unit stride")
    d.a_kc = 1
    d.dtA, d.dtB, d.dtC, d.mma = MH_BF16, MH_BF16, MH_BF16, MH_BF16
    d.batch1 = d.batch2 = 1
    d.alpha, d.split_k = 1.0, 1
    if Bn > 1:
        if R != Ta:
            d.a_rows_per_batch, d.a_row_skip = R, Ta - R
        if R != T:
            d.c_rows_per_batch, d.c_row_skip = R, T - R
    d.shared_chip = 1 if shared_chip else 0
    prof = gemm_profiler
    if prof is None:
        _lib.call("mh_gemm", C.byref(d), stream=_stream())
    else:
        prof.launch(d, lambda: _lib.call("mh_gemm", C.byref(d), stream=_stream()))


def gemm_rows_ext_ok(Bn: int, T: int, r0: int, R: int, extra: int, Kd: int, N: int, a2: torch.Tensor, out2: torch.Tensor) -> bool:
    """shapes for which gemm_rows_ext runs on the direct-to-LDS 256 x 256 kernels."""
    bf = torch.bfloat16
    M = Bn * R + extra
    return (a2.dim() == 2 and out2.dim() == 2 and a2.dtype == bf and out2.dtype == bf and a2.is_cuda and R >= 256 and M > 512 and 0 <= extra < R
            and r0 + R == T and M % 256 <= extra
            and N % 256 == 0 and Kd % 64 == 0 and a2.stride(1) == 1 and out2.stride(1) == 1 and a2.stride(0) % 8 == 0 and out2.stride(0) % 8 == 0
            and a2.data_ptr() % 16 == 0 and out2.data_ptr() % 16 == 0 and a2.shape[0] == Bn * T + extra and out2.shape[0] == Bn * T + extra
            and int(_lib.load().mh_gemm_select_pp(-1)) != 0)


def gemm_rows_ext(a2: torch.Tensor, b2: torch.Tensor, out2: torch.Tensor, Bn: int, T: int, r0: int, R: int, extra: int) -> int:
    """The row-window product of gemm_rows_window over buffers that hold `extra` more rows behind the Bn batches of T rows (the landmark
    rows of a Nystrom layer: a2 [Bn * T + extra, K], out2 [Bn * T + extra, N], 2-D views with unit column stride): ONE flat problem over
    the Bn * R real rows (r0 + R == T: the window ends with the batch) followed by the extra rows, M = Bn * R + extra.  The kernels'
    row remap r -> r + min(r / R, Bn - 1) * (T - R) (mh_gemm_desc.window_batches) lands the extra rows right behind the last batch.
    Only whole 256-row tiles are computed; returns the number of flat rows left over at the end (< 256, all of them extra rows) for the
    caller to finish (a2[-tail:] @ b2 -> out2[-tail:])."""
    _chk(a2, b2, out2)
    Kd, N = a2.shape[1], b2.shape[1]
    if not gemm_rows_ext_ok(Bn, T, r0, R, extra, Kd, N, a2, out2) or b2.shape[0] != Kd or b2.dtype != torch.bfloat16:
        raise MirrorHipError("gemm_rows_ext: bad operands")
    M = Bn * R + extra
    tail = M % 256
    d = GemmDesc()
    d.A, d.B, d.C, d.bias = a2.data_ptr() + r0 * a2.stride(0) * 2, b2.data_ptr(), out2.data_ptr() + r0 * out2.stride(0) * 2, None
    d.M, d.N, d.K = M - tail, N, Kd
    d.lda, d.ldc = a2.stride(0), out2.stride(0)
    if b2.stride(0) == 1:            # W^T of a row-major [N, K] weight: contraction index contiguous
        d.ldb, d.b_kc = b2.stride(1), 1
    elif b2.stride(1) == 1:          # row-major [K, N]
        d.ldb, d.b_kc = b2.stride(0), 0
    else:
        raise MirrorHipError("gemm_rows_ext: the weight view must have a unit stride")
    d.a_kc = 1
    d.dtA, d.dtB, d.dtC, d.mma = MH_BF16, MH_BF16, MH_BF16, MH_BF16
    d.batch1 = d.batch2 = 1
    d.alpha, d.split_k = 1.0, 1
    if R != T:
        d.a_rows_per_batch, d.a_row_skip = R, T - R
        d.c_rows_per_batch, d.c_row_skip = R, T - R
        d.window_batches = Bn          # the extra rows follow the last batch's window without a gap
    d.shared_chip = 1 if shared_chip else 0
    prof = gemm_profiler
    if prof is None:
        _lib.call("mh_gemm", C.byref(d), stream=_stream())
    else:
        prof.launch(d, lambda: _lib.call("mh_gemm", C.byref(d), stream=_stream()))
    return tail


def gemm_rows_window_ok(a3: torch.Tensor, out3: torch.Tensor, r0: int, R: int, N: int) -> bool:
    """shapes / build for which gemm_rows_window runs (the direct-to-LDS 256 x 256 kernels)."""
    bf = torch.bfloat16
    return (a3.dim() == 3 and out3.dim() == 3 and a3.dtype == bf and out3.dtype == bf and a3.is_cuda and R >= 256 and a3.shape[0] * R > 512
            and N % 256 == 0 and a3.shape[2] % 64 == 0 and a3.stride(2) == 1 and out3.stride(2) == 1 and a3.stride(1) % 8 == 0
            and out3.stride(1) % 8 == 0 and a3.data_ptr() % 16 == 0 and out3.data_ptr() % 16 == 0
            and a3.stride(0) == a3.shape[1] * a3.stride(1) and out3.stride(0) == out3.shape[1] * out3.stride(1)
            and int(_lib.load().mh_gemm_select_pp(-1)) != 0)


def epi_dropadd(resid: torch.Tensor, p: float, seed: int, offset: int, dev_base) -> "_lib.GemmEpi":
    _chk(resid)
    e = _lib.GemmEpi()
    e.kind, e.resid, e.p, e.seed, e.offset, e.dev_base = EPI_DROPADD, resid.data_ptr(), float(p), int(seed), int(offset), _p(dev_base)
    return e


def epi_maskpos(mask: torch.Tensor, token: torch.Tensor, pos: torch.Tensor, rows_per_batch: int, first: int) -> "_lib.GemmEpi":
    _chk(mask, token, pos)
    e = _lib.GemmEpi()
    e.kind, e.mask, e.token, e.pos, e.rows_per_batch, e.first = EPI_MASKPOS, mask.data_ptr(), token.data_ptr(), pos.data_ptr(), int(rows_per_batch), int(first)
    return e


def epi_sqerr(mask: torch.Tensor, tgt: torch.Tensor, tgt_bs: int, sq: torch.Tensor, rows_per_batch: int) -> "_lib.GemmEpi":
    _chk(mask, tgt, sq)
    e = _lib.GemmEpi()
    e.kind, e.mask, e.tgt, e.tgt_bs, e.sq, e.rows_per_batch = EPI_SQERR, mask.data_ptr(), tgt.data_ptr(), int(tgt_bs), sq.data_ptr(), int(rows_per_batch)
    return e


def gemm_softmax_ok(M: int, N: int, Kd: int, dta=torch.bfloat16, dtb=torch.bfloat16) -> bool:
    """Shapes whose row softmax the 192 x 384 tile kernel computes in its epilogue (a whole row of length N = 384 in one tile)."""
    return M % 192 == 0 and N == 384 and Kd % 8 == 0 and dta == torch.bfloat16 and dtb == torch.bfloat16


def gemm_ksum(a_stack: torch.Tensor, b_stack: torch.Tensor, out: Optional[torch.Tensor] = None, **kw) -> torch.Tensor:
    """sum_s a_stack[s] @ b_stack[s] in ONE launch and one accumulator (192 x 384 tile kernel): the operand pairs sit a constant
    stride apart (the leading dim of the two stacks; a transpose of the trailing dims or a strided slice of a longer stack
    is fine), so the K loop simply walks from one pair to the next and the f32 sum never takes a read-modify-write through HBM."""
    if a_stack.shape[0] != b_stack.shape[0] or a_stack.shape[0] < 1:
        raise MirrorHipError("gemm_ksum: the stacks need the same, non-zero leading length")
    return gemm(a_stack[0], b_stack[0], out, kseg=(a_stack.shape[0], a_stack.stride(0), b_stack.stride(0)), **kw)


def gemm_tile_ok(M: int, N: int, Kd: int, dta=torch.bfloat16, dtb=torch.bfloat16) -> bool:
    """Shapes the 192 x 384 tile kernel (csrc/gemm_tile.hip) takes: the batched 384-cubed products of the template's pinv."""
    return M % 192 == 0 and N % 384 == 0 and Kd % 64 == 0 and dta == torch.bfloat16 and dtb == torch.bfloat16


_GEMM_WS = True      # split-K partials in a workspace + fold pass (False: f32 atomics; test hook)
# Precision policy of the split-K weight-gradient reductions.  "bf16": every K-slice's f32 accumulator is rounded to bf16 on
# its way to the workspace and the fold sums in f32 (half the partial traffic: -0.77 % step time); the
# extra error of an element is bounded by parts * 2^-9 * max |partial| — relative to the largest partial, not to the final sum
# (tests/test_kernels_gpu.py::test_split_k_bf16_partials_error_bound_under_cancellation).  "f32": no workspace, f32 atomics into C
# (exact f32 partial sums, order not reproducible).  The bf16 form is the bf16 training policy's default; MIRROR_SPLITK_PARTIALS=f32
# selects the other for a run, and the fp32 parity policy never splits into bf16 (mh_gemm_workspace_bytes is 0 for f32 operands).
import os as _os
SPLITK_PARTIALS = _os.environ.get("MIRROR_SPLITK_PARTIALS", "bf16")
if SPLITK_PARTIALS not in ("bf16", "f32"):
    raise MirrorHipError(f"MIRROR_SPLITK_PARTIALS={SPLITK_PARTIALS!r}: 'bf16' or 'f32'")


class GemmProfiler:
    """Times GEMM (and named fused-kernel) launches with HIP events on the launch stream (bench.py's roofline leg).  A GEMM launch is
    named by the LIBRARY after the fact (mh_gemm_variant_name: the template instance mh_gemm dispatched to, as rocprofv3 reports
    it), so nothing here restates the dispatch rules.  only=<variant>: summary() keeps just that instance."""

    def __init__(self, only: Optional[str] = None):
        self.only = only
        self.records = []   # (variant, flops, start_event, end_event)

    def launch(self, d: GemmDesc, fn) -> None:
        if torch.cuda.is_current_stream_capturing():
            fn()                 # events recorded inside a capture (the RNA branch graph) are not timing events
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()          # recorded on the stream the kernel is launched on (torch's current stream)
        fn()
        e1.record()
        name = (_lib.load().mh_gemm_variant_name() or b"").decode()
        if d.epi and not name.endswith(f",epi{d.epi.contents.kind}>"):
            name = name[:-1] + f",epi{d.epi.contents.kind}>"
        self.records.append((name, 2.0 * d.M * d.N * d.K * d.batch1 * d.batch2, e0, e1))

    def launch_named(self, v: str, flops: float, fn) -> None:
        if (self.only is not None and v != self.only) or torch.cuda.is_current_stream_capturing():
            fn()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.records.append((v, flops, e0, e1))

    def summary(self):
        """variant -> dict(launches, total_ms, flops); call after a device synchronise."""
        out = {}
        for v, fl, e0, e1 in self.records:
            if self.only is not None and v != self.only:
                continue
            s = out.setdefault(v, {"launches": 0, "total_ms": 0.0, "flops": 0.0})
            s["launches"] += 1
            s["total_ms"] += e0.elapsed_time(e1)
            s["flops"] += fl
        return out


gemm_profiler: Optional[GemmProfiler] = None


# ----------------------------------------------------------------------------- fp8 forward projections (config 5)
def quant_fp8(x: torch.Tensor):
    """Per-tensor e4m3 quantisation: (q uint8 like x, scale f32[1]) with x ~= q * scale."""
    _chk(x)
    _contig(x, "quant_fp8 input")
    if x.numel() % 4:
        raise MirrorHipError("quant_fp8: numel must be a multiple of 4")
    q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    scale = torch.empty((1,), device=x.device, dtype=torch.float32)
    scratch = torch.empty((1,), device=x.device, dtype=torch.int32)
    _lib.call("mh_quant_fp8", _p(x), x.numel(), _p(q), _p(scale), _p(scratch), dt(x), stream=_stream())
    return q, scale


def quant_fp8_delayed(x: torch.Tensor, ring: torch.Tensor, tick: torch.Tensor, margin: float = 1.25):
    """One-pass e4m3 quantisation with the previous step's scale (ring: int32[3] device state of this call site, tick: device f32
    step counter).  Returns (q, scale) like quant_fp8."""
    _chk(x, ring, tick)
    _contig(x, "quant_fp8_delayed input")
    if x.numel() % 4 or ring.numel() != 3 or ring.dtype != torch.int32 or tick.dtype != torch.float32:
        raise MirrorHipError("quant_fp8_delayed: bad operands")
    q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    scale = torch.empty((1,), device=x.device, dtype=torch.float32)
    _lib.call("mh_quant_fp8_delayed", _p(x), x.numel(), _p(q), _p(scale), _p(ring), _p(tick), float(margin), dt(x), stream=_stream())
    return q, scale


def gemm_fp8_ok(a: torch.Tensor, n_out: int) -> bool:
    """a [..., R, K] as e4m3 bytes: K % 64 == 0, N % 128 == 0, rows K-contiguous, <= one batch dim with 16-byte strides."""
    Kd = a.shape[-1]
    if Kd % 64 or n_out % 128 or a.stride(-1) != 1 or a.dim() not in (2, 3) or a.stride(-2) % 16:
        return False
    return a.dim() == 2 or a.stride(0) % 16 == 0


def gemm_fp8(aq: torch.Tensor, sa: torch.Tensor, wq: torch.Tensor, sw: torch.Tensor, out: torch.Tensor,
             bias: Optional[torch.Tensor] = None, act: int = ACT_NONE) -> torch.Tensor:
    """out[..., R, N] = act(sa * sw * aq @ wq^T + bias); aq uint8 [R, K] or [B, R, K] (a row window is fine), wq uint8 [N, K]
    contiguous, out f32 / bf16 with unit column stride (a row window of a larger buffer is fine)."""
    _chk(aq, wq, out, sa, sw, bias)
    N, Kd = wq.shape
    assert aq.dtype == torch.uint8 and wq.dtype == torch.uint8 and wq.is_contiguous() and aq.shape[-1] == Kd
    assert out.stride(-1) == 1 and out.shape[-1] == N and out.shape[:-1] == aq.shape[:-1]
    if aq.dim() == 3:
        batch, M, a_bs, c_bs = aq.shape[0], aq.shape[1], aq.stride(0), out.stride(0)
    else:
        batch, M, a_bs, c_bs = 1, aq.shape[0], 0, 0
    _lib.call("mh_gemm_fp8", _p(aq), aq.stride(-2), a_bs, _p(wq), Kd, _p(out), out.stride(-2), c_bs, batch, _p(sa), _p(sw), _p(bias),
              act, M, N, Kd, dt(out), stream=_stream())
    return out


# ----------------------------------------------------------------------------- skinny-M linears
def skinny_ok(x2d: torch.Tensor, w: torch.Tensor) -> bool:
    """bf16 or f32 x [M<=32, K] (row-strided ok; f32 is rounded to bf16 on load: no cast launch) against bf16 W [N, K] contiguous.
    K % 32 == 0 with 16-byte aligned rows runs on 16-byte fragments, anything else (10234 genes, a 1975-wide MLP) on the element-wise
    instance of the same kernel (skinny_vec_ok says which)."""
    return (x2d.dim() == 2 and w.dim() == 2 and x2d.dtype in (torch.bfloat16, torch.float32) and w.dtype == torch.bfloat16
            and 1 <= x2d.shape[0] <= 32 and x2d.shape[1] == w.shape[1] and w.shape[1] >= 1 and w.is_contiguous() and x2d.stride(1) == 1)


def skinny_vec_ok(x2d: torch.Tensor, w: torch.Tensor) -> bool:
    """The 16-byte-fragment instance of mh_skinny_fwd applies (what mh_skinny_fwd itself checks)."""
    return (skinny_ok(x2d, w) and w.shape[1] % 32 == 0 and x2d.stride(0) % 8 == 0 and x2d.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


def skinny_fwd(x2d: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: int, out_dtype, out: Optional[torch.Tensor] = None,
               addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out: optional [M, N] destination rows (any row stride that keeps rows 16-byte aligned, e.g. rows of a larger buffer);
    addend: optional f32 [M, N] summed in front of the activation (two data gradients of one x in two launches, no add)."""
    _chk(x2d, w, bias, out, addend)
    if addend is not None and (addend.dtype != torch.float32 or tuple(addend.shape) != (x2d.shape[0], w.shape[0]) or addend.stride(1) != 1):
        raise MirrorHipError("skinny_fwd: the addend is f32 [M, N] with unit inner stride")
    M, Kd = x2d.shape
    N = w.shape[0]
    y = torch.empty((M, N), device=x2d.device, dtype=out_dtype) if out is None else out
    if tuple(y.shape) != (M, N) or y.stride(1) != 1 or x2d.stride(1) != 1 or w.stride(1) != 1:
        raise MirrorHipError("skinny_fwd: bad operands")
    _lib.call("mh_skinny_fwd", _p(x2d), x2d.stride(0), _p(w), w.stride(0), _p(bias), _p(addend), 0 if addend is None else addend.stride(0),
              _p(y), y.stride(0), M, N, Kd, act, dt(x2d), dt(y), stream=_stream())
    return y


def skinny_rows_ok(x2d: torch.Tensor, w: torch.Tensor, y2d: torch.Tensor) -> bool:
    """mh_skinny_fwd's limits for rows of larger buffers (x2d [M <= 32, K], w [N, K], y2d [M, N]; unit inner strides)."""
    ok_dt = (torch.bfloat16, torch.float32)
    return (x2d.dim() == 2 and w.dim() == 2 and y2d.dim() == 2 and 1 <= x2d.shape[0] <= 32 and x2d.shape[1] % 32 == 0
            and x2d.shape[1] == w.shape[1] and tuple(y2d.shape) == (x2d.shape[0], w.shape[0])
            and x2d.stride(1) == 1 and w.stride(1) == 1 and y2d.stride(1) == 1
            and x2d.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and y2d.stride(0) % 8 == 0
            and x2d.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and y2d.data_ptr() % 16 == 0
            and x2d.dtype in ok_dt and y2d.dtype in ok_dt and w.dtype == torch.bfloat16)


def skinny_wgrad(dy2d: torch.Tensor, x2d: torch.Tensor, dw: torch.Tensor, accumulate: bool, db: Optional[torch.Tensor] = None) -> None:
    """dW (+)= dy^T x; with `db` ([N] f32) also db += column sums of dy (the bias gradient) in the same launch."""
    _chk(dy2d, x2d, dw, db)
    assert db is None or (db.dtype == torch.float32 and db.numel() == dy2d.shape[1] and db.is_contiguous())
    M, N = dy2d.shape
    Kd = x2d.shape[1]
    ok_dt = (torch.bfloat16, torch.float32)
    if (dy2d.dtype not in ok_dt or x2d.dtype not in ok_dt or dw.dtype != torch.float32 or x2d.shape[0] != M
            or tuple(dw.shape) != (N, Kd) or dy2d.stride(1) != 1 or x2d.stride(1) != 1 or dw.stride(1) != 1):
        raise MirrorHipError("skinny_wgrad: bad operands")
    _lib.call("mh_skinny_wgrad", _p(dy2d), dy2d.stride(0), _p(x2d), x2d.stride(0), _p(dw), dw.stride(0), _p(db), M, N, Kd,
              int(accumulate), dt(dy2d), dt(x2d), stream=_stream())


SKINNY_MANY_MAX = 32


def skinny_wgrad_many(items) -> None:
    """items: list of (dy2d, x2d, dw, db or None) as skinny_wgrad takes them (always accumulating); ONE launch per 32 items."""
    for i0 in range(0, len(items), SKINNY_MANY_MAX):
        chunk = items[i0:i0 + SKINNY_MANY_MAX]
        arr = (_lib.SkinnyWgradItem * len(chunk))()
        for e, (dy2d, x2d, dw, db) in zip(arr, chunk):
            _chk(dy2d, x2d, dw, db)
            M, N = dy2d.shape
            Kd = x2d.shape[1]
            if (dy2d.dtype not in (torch.bfloat16, torch.float32) or x2d.dtype not in (torch.bfloat16, torch.float32) or dw.dtype != torch.float32 or x2d.shape[0] != M
                    or tuple(dw.shape) != (N, Kd) or dy2d.stride(1) != 1 or x2d.stride(1) != 1 or dw.stride(1) != 1
                    or (db is not None and (db.dtype != torch.float32 or db.numel() != N or not db.is_contiguous()))):
                raise MirrorHipError("skinny_wgrad_many: bad operands")
            e.dy, e.lddy, e.x, e.ldx, e.dw, e.lddw, e.db = dy2d.data_ptr(), dy2d.stride(0), x2d.data_ptr(), x2d.stride(0), dw.data_ptr(), dw.stride(0), _p(db)
            e.M, e.N, e.K, e.dt_dy, e.dt_x = M, N, Kd, dt(dy2d), dt(x2d)
        _lib.call("mh_skinny_wgrad_many", arr, len(chunk), stream=_stream())


def transpose_bf16(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(x, out)
    _contig(x, "transpose input")
    R, Cc = x.shape
    y = out if out is not None else torch.empty((Cc, R), device=x.device, dtype=torch.bfloat16)
    _lib.call("mh_transpose_bf16", _p(x), _p(y), R, Cc, stream=_stream())
    return y


def transpose_bf16_many(src: torch.Tensor, dst: torch.Tensor, table: torch.Tensor, n: int, max_r: int, max_c: int,
                        vec_ok: bool = False) -> None:
    """vec_ok: the caller checked that every table entry has R % 8 == C % 8 == 0 and offsets that are multiples of 8."""
    _chk(src, dst, table)
    assert table.dtype == torch.int64 and table.is_contiguous() and table.numel() == 4 * n
    _lib.call("mh_transpose_bf16_many", _p(src), _p(dst), _p(table), n, max_r, max_c, int(vec_ok), stream=_stream())


# ----------------------------------------------------------------------------- row kernels
def layernorm_fwd(x, gamma, beta, y, mean, rstd, batches, rpb, D, x_bs, y_bs, eps):
    _chk(x, gamma, beta, y, mean, rstd)
    _lib.call("mh_layernorm_fwd", _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), batches, rpb, D, x_bs, y_bs,
              eps, dt(x), dt(y), stream=_stream())


def layernorm_fwd_dual(x, gamma, beta, y, y16, mean, rstd, batches, rpb, D, x_bs, y_bs, eps):
    """layernorm_fwd (f32 in, f32 out) that also writes the bf16 copy y16 (same row addressing as y)."""
    _chk(x, gamma, beta, y, y16, mean, rstd)
    if not (x.dtype == torch.float32 and y.dtype == torch.float32 and y16.dtype == torch.bfloat16):
        raise MirrorHipError("layernorm_fwd_dual: f32 input, f32 + bf16 outputs")
    _lib.call("mh_layernorm_fwd_dual", _p(x), _p(gamma), _p(beta), _p(y), _p(y16), _p(mean), _p(rstd), batches, rpb, D, x_bs, y_bs,
              eps, stream=_stream())


def layernorm_fwd_q8(x, gamma, beta, y, mean, rstd, batches, rpb, D, x_bs, y_bs, eps, q8, ring, tick, margin: float = 1.25):
    """layernorm_fwd (f32 in, bf16 out) that also writes the e4m3 copy q8 (uint8, y's row addressing) with the delayed scale of
    `ring` (int32[3] device state of the call site) / `tick`; returns the dequantisation factor (f32[1])."""
    _chk(x, gamma, beta, y, mean, rstd, q8, ring, tick)
    if not (x.dtype == torch.float32 and y.dtype == torch.bfloat16 and q8.dtype == torch.uint8 and ring.numel() == 3
            and ring.dtype == torch.int32 and tick.dtype == torch.float32):
        raise MirrorHipError("layernorm_fwd_q8: f32 input, bf16 + uint8 outputs, int32[3] ring, f32 tick")
    scale = torch.empty((1,), device=x.device, dtype=torch.float32)
    _lib.call("mh_layernorm_fwd_q8", _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), batches, rpb, D, x_bs, y_bs, eps,
              _p(q8), _p(ring), _p(tick), float(margin), _p(scale), stream=_stream())
    return scale


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, batches, rpb, D, x_bs, y_bs, accumulate_dx=False, gadd=None,
                  pad: int = 0, l: int = 1, relu_out=None, relu_first: int = 0, relu_db=None, fan=None, drop=None, row_mask=None, lm_scale=None):
    """gadd ([batches, (pad + rpb) / l, D], dy's dtype): the gradient of the landmark means layernorm_fwd_lm produced; every dy row also
    receives gadd[b, (i + pad) / l] / l (mh_layernorm_bwd_lm).  relu_out (bf16 [batches, R, D], with gadd only): rows [relu_first,
    relu_first + R) of x are a ReLU's output — their gradient leaves as bf16 (x > 0 ? dx : 0) in relu_out instead of f32 dx;
    relu_db (f32 [D], with relu_out): += the column sums of relu_out (the bias gradient of the Linear in front of the ReLU).
    fan = (src bf16 [batches, rpb - 1, D], alpha, cls f32 [batches, D] or None), without gadd: dy rows 1.. also receive alpha * src, row 0
    cls (mh_layernorm_bwd_fan; layernorm_bwd_fan_ok says whether the shapes are on that form).
    drop = (out bf16 [batches, >= rpb, D] (the Dropout's tensor; rows past rpb are the caller's), p, seed, offset, dev_base, db f32 [D]), without gadd: x is the output of resid + Dropout_p(Linear),
    so out receives the lite-stream dropout backward of this launch's dx and db its column sums (mh_layernorm_bwd_drop)."""
    _chk(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, relu_out, relu_db)
    if relu_db is not None and (relu_out is None or relu_db.dtype != torch.float32 or relu_db.numel() != D or not relu_db.is_contiguous()):
        raise MirrorHipError("layernorm_bwd: relu_db is a contiguous f32 [D] and rides on relu_out")
    if relu_out is not None and (gadd is None or relu_out.dtype != torch.bfloat16 or not relu_out.is_contiguous() or relu_out.dim() != 3
                                 or relu_out.shape[0] != batches or relu_out.shape[2] != D or relu_first + relu_out.shape[1] > rpb):
        raise MirrorHipError("layernorm_bwd: relu_out must be contiguous bf16 [batches, R, D] with relu_first + R <= rows (landmark form only)")
    rows = batches * rpb
    ws = None
    nbytes = int(_lib.load().mh_layernorm_bwd_workspace_bytes(rows, D))
    if nbytes:          # per-block dgamma / dbeta partials (folded by a second launch) instead of same-address atomics
        ws = torch.empty((nbytes // 4,), device=x.device, dtype=torch.float32)
    if lm_scale is not None:
        _chk(lm_scale)
        if gadd is None or lm_scale.dtype != torch.float32 or not lm_scale.is_contiguous() or lm_scale.numel() != batches * ((pad + rpb) // l):
            raise MirrorHipError("layernorm_bwd: lm_scale is a contiguous f32 [batches, (pad + rows) / l] and rides on the landmark form (gadd)")
    if row_mask is not None:
        _chk(row_mask)
        if gadd is None or row_mask.dtype != torch.float32 or not row_mask.is_contiguous() or row_mask.numel() != batches * (pad + rpb):
            raise MirrorHipError("layernorm_bwd: row_mask is a contiguous f32 [batches, pad + rows] and rides on the landmark form (gadd)")
    if gadd is not None:
        _chk(gadd)
        if gadd.dtype != dy.dtype or not gadd.is_contiguous() or gadd.numel() != batches * ((pad + rpb) // l) * D:
            raise MirrorHipError("layernorm_bwd: gadd must be contiguous [batches, (pad + rows) / l, D] in dy's dtype")
        _lib.call("mh_layernorm_bwd_lm", _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgamma), _p(dbeta),
                  batches, rpb, D, x_bs, y_bs, dt(x), dt(dy), dt(dx), int(accumulate_dx), _p(ws) if ws is not None else 0,
                  ws.numel() if ws is not None else 0, _p(gadd), int(pad), int(l), _p(relu_out), int(relu_first),
                  0 if relu_out is None else int(relu_out.shape[1]), _p(relu_db), _p(row_mask), _p(lm_scale), stream=_stream())
        return
    if drop is not None:
        dout, p_, seed_, off_, base_, ddb = drop
        src, alpha, cls = fan if fan is not None else (None, 0.0, None)
        _chk(dout, base_, ddb, src, cls)
        if (gadd is not None or ws is None or not layernorm_bwd_drop_ok(dy, x, dx, dout, ddb, batches, rpb, D, off_)
                or (fan is not None and not layernorm_bwd_fan_ok(dy, x, dx, src, cls, batches, rpb, D))):
            raise MirrorHipError("layernorm_bwd: drop needs f32 x / dx, a contiguous bf16 [batches, rows, D] output, D % 8 == 0, D <= 1024, offset % 8 == 0, >= 64 rows")
        _lib.call("mh_layernorm_bwd_drop", _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgamma), _p(dbeta),
                  batches, rpb, D, x_bs, y_bs, dt(dy), int(accumulate_dx), _p(ws), ws.numel(), _p(src), float(alpha), _p(cls),
                  _p(dout), float(p_), int(seed_), int(off_), _p(base_), _p(ddb), int(dout.shape[1]), stream=_stream())
        return
    if fan is not None:
        src, alpha, cls = fan
        _chk(src, cls)
        if not layernorm_bwd_fan_ok(dy, x, dx, src, cls, batches, rpb, D) or ws is None:
            raise MirrorHipError("layernorm_bwd: fan needs f32 x / dx, a contiguous bf16 [batches, rows - 1, D] source and >= 64 rows")
        _lib.call("mh_layernorm_bwd_fan", _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgamma), _p(dbeta),
                  batches, rpb, D, x_bs, y_bs, dt(dy), int(accumulate_dx), _p(ws), ws.numel(), _p(src), float(alpha), _p(cls), stream=_stream())
        return
    _lib.call("mh_layernorm_bwd", _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgamma), _p(dbeta),
              batches, rpb, D, x_bs, y_bs, dt(x), dt(dy), dt(dx), int(accumulate_dx), _p(ws) if ws is not None else 0,
              ws.numel() if ws is not None else 0, stream=_stream())


def layernorm_bwd_drop_ok(dy, x, dx, dout, ddb, batches: int, rpb: int, D: int, offset: int) -> bool:
    f = torch.float32
    return (x.dtype == f and dx.dtype == f and dy.dtype in (f, torch.bfloat16) and dout.dtype == torch.bfloat16 and dout.is_contiguous()
            and dout.dim() == 3 and dout.shape[0] == batches and dout.shape[1] >= rpb and dout.shape[2] == D and ddb.dtype == f and ddb.is_contiguous() and ddb.numel() == D and D % 8 == 0 and D <= 1024
            and offset % 8 == 0 and batches * rpb >= 64 and all(t.data_ptr() % 16 == 0 for t in (dy, x, dx, dout)))


def layernorm_bwd_fan_ok(dy, x, dx, src, cls, batches: int, rpb: int, D: int) -> bool:
    f = torch.float32
    return (dy.dtype in (f, torch.bfloat16) and x.dtype == f and dx.dtype == f and src.dtype == torch.bfloat16 and src.is_contiguous()
            and src.numel() == batches * (rpb - 1) * D and rpb >= 2 and batches * rpb >= 64 and D % 4 == 0
            and (cls is None or (cls.dtype == f and cls.is_contiguous() and cls.numel() == batches * D and cls.data_ptr() % 16 == 0))
            and all(t.data_ptr() % 16 == 0 for t in (dy, x, dx)) and src.data_ptr() % 8 == 0)


def layernorm_fwd_lm(x, gamma, beta, y, mean, rstd, xpm, batches, rows, D, x_bs, pad, l, eps, xpm_bf16=None, row_mask=None, lm_scale=None):
    """LayerNorm (f32 in, bf16 out behind `pad` zero rows, which this launch writes) + the landmark means xpm [batches, (pad + rows) / l, D]
    (f32) of its output rows (mh_layernorm_fwd_lm)."""
    _chk(x, gamma, beta, y, mean, rstd, xpm, xpm_bf16, row_mask, lm_scale)
    if lm_scale is not None and not (lm_scale.dtype == torch.float32 and lm_scale.is_contiguous() and lm_scale.numel() == batches * ((pad + rows) // l)):
        raise MirrorHipError("layernorm_fwd_lm: lm_scale is a contiguous f32 [batches, (pad + rows) / l]")
    if row_mask is not None and not (row_mask.dtype == torch.float32 and row_mask.is_contiguous() and row_mask.numel() == batches * (pad + rows)):
        raise MirrorHipError("layernorm_fwd_lm: row_mask is a contiguous f32 [batches, pad + rows]")
    if not (x.dtype == torch.float32 and y.dtype == torch.bfloat16 and y.is_contiguous() and (xpm is not None or xpm_bf16 is not None)
            and (xpm is None or (xpm.dtype == torch.float32 and xpm.is_contiguous()))):
        raise MirrorHipError("layernorm_fwd_lm: f32 input, contiguous bf16 rows and f32 (and / or bf16) landmark means")
    n_lm = batches * ((pad + rows) // l) * D
    if xpm_bf16 is not None and not (xpm_bf16.dtype == torch.bfloat16 and xpm_bf16.is_contiguous() and xpm_bf16.numel() == n_lm):
        raise MirrorHipError("layernorm_fwd_lm: xpm_bf16 must be a contiguous bf16 tensor of batches * (pad + rows) / l * D elements")
    _lib.call("mh_layernorm_fwd_lm", _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(xpm), _p(xpm_bf16), batches, rows, D, x_bs, int(pad),
              int(l), eps, _p(row_mask), _p(lm_scale), stream=_stream())


def softmax_fwd(x: torch.Tensor, y: Optional[torch.Tensor] = None, out_dtype=None) -> torch.Tensor:
    """softmax over the last dim of a contiguous tensor (in place when y is x)."""
    _chk(x, y)
    _contig(x, "softmax input")
    cols = x.shape[-1]
    if y is None:
        y = torch.empty_like(x, dtype=out_dtype or x.dtype)
    _contig(y, "softmax output")
    _lib.call("mh_softmax_fwd", _p(x), _p(y), x.numel() // max(cols, 1), cols, cols, cols, dt(x), dt(y), stream=_stream())
    return y


def softmax_bwd(y: torch.Tensor, dy: torch.Tensor, dx: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(y, dy, dx)
    _contig(y, "softmax y"), _contig(dy, "softmax dy")
    cols = y.shape[-1]
    if dx is None:
        dx = dy
    _lib.call("mh_softmax_bwd", _p(y), _p(dy), _p(dx), y.numel() // max(cols, 1), cols, cols, cols, cols, dt(y), dt(dy),
              dt(dx), stream=_stream())
    return dx


def softmax_masked_fwd(x: torch.Tensor, rowmask: torch.Tensor, colmask: torch.Tensor, y: Optional[torch.Tensor] = None,
                       out_dtype=None) -> torch.Tensor:
    """softmax(masked_fill(x, ~(rowmask[b, i] & colmask[b, j]), -max)) over the last dim of x [B, h, R, C]."""
    _chk(x, y, rowmask, colmask)
    _contig(x, "softmax input")
    Bn, h, R, Cc = x.shape
    assert rowmask.shape == (Bn, R) and colmask.shape == (Bn, Cc) and rowmask.dtype == colmask.dtype == torch.float32
    if y is None:
        y = torch.empty_like(x, dtype=out_dtype or x.dtype)
    _contig(y, "softmax output")
    _lib.call("mh_softmax_masked_fwd", _p(x), _p(y), _p(rowmask.contiguous()), _p(colmask.contiguous()), Bn, h, R, Cc, dt(x), dt(y),
              stream=_stream())
    return y


def softmax_masked_bwd(y: torch.Tensor, dy: torch.Tensor, rowmask: torch.Tensor, colmask: torch.Tensor) -> torch.Tensor:
    """In place on dy."""
    _chk(y, dy, rowmask, colmask)
    _contig(y, "softmax y"), _contig(dy, "softmax dy")
    Bn, h, R, Cc = y.shape
    _lib.call("mh_softmax_masked_bwd", _p(y), _p(dy), _p(dy), _p(rowmask.contiguous()), _p(colmask.contiguous()), Bn, h, R, Cc, dt(y),
              dt(dy), stream=_stream())
    return dy


def row_scale(x: torch.Tensor, scale: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[..., r, :] = x[..., r, :] * scale[..., r] (x contiguous, scale f32 with one entry per row)."""
    _chk(x, scale, out)
    _contig(x, "row_scale input")
    D = x.shape[-1]
    rows = x.numel() // max(D, 1)
    assert scale.numel() == rows and scale.dtype == torch.float32
    if out is None:
        out = torch.empty_like(x)
    _lib.call("mh_row_scale", _p(x), _p(scale.contiguous()), _p(out), rows, D, dt(x), stream=_stream())
    return out


def keymask_plan(mask: torch.Tensor, lead: int, wrap: int, pad: int, l: int):  # noqa: E741
    """(mrow [B, m*l], mlm [B, m], lscale [B, m]) f32 for the sequence [pad zeros | lead ones | mask | mask[:, :wrap]] (mh_keymask_plan)."""
    _chk(mask)
    _contig(mask, "key-padding mask")
    assert mask.dtype == torch.bool and mask.dim() == 2
    B, n_src = mask.shape
    n_tot = pad + lead + n_src + wrap
    assert l > 0 and n_tot % l == 0 and 0 <= wrap <= n_src, "the padded sequence must be m groups of l rows"
    m = n_tot // l
    mrow = torch.empty((B, n_tot), device=mask.device, dtype=torch.float32)
    mlm = torch.empty((B, m), device=mask.device, dtype=torch.float32)
    lsc = torch.empty((B, m), device=mask.device, dtype=torch.float32)
    _lib.call("mh_keymask_plan", _p(mask), _p(mrow), _p(mlm), _p(lsc), B, n_src, lead, wrap, pad, l, stream=_stream())
    return mrow, mlm, lsc


def l2norm_fwd(x2d_rows: torch.Tensor, rows: int, D: int, x_rs: int, eps: float, out_dtype):
    _chk(x2d_rows)
    y = torch.empty((rows, D), device=x2d_rows.device, dtype=out_dtype)
    nrm = torch.empty((rows,), device=x2d_rows.device, dtype=torch.float32)
    _lib.call("mh_l2norm_fwd", _p(x2d_rows), _p(y), _p(nrm), rows, D, x_rs, eps, dt(x2d_rows), dt(y), stream=_stream())
    return y, nrm


def l2norm_bwd(y, nrm, dy, dx, rows, D, dx_rs, accumulate):
    _chk(y, nrm, dy, dx)
    _lib.call("mh_l2norm_bwd", _p(y), _p(nrm), _p(dy), _p(dx), rows, D, dx_rs, 0.0, dt(y), dt(dy), dt(dx),
              int(accumulate), stream=_stream())


# ----------------------------------------------------------------------------- Nystrom pieces
def landmark_fwd(qkv: torch.Tensor, l: int) -> torch.Tensor:
    _chk(qkv)
    _contig(qkv, "qkv")
    B, n_p, D3 = qkv.shape
    D = D3 // 3
    lm = torch.empty((B, n_p // l, 2 * D), device=qkv.device, dtype=qkv.dtype)
    _lib.call("mh_landmark_fwd", _p(qkv), _p(lm), B, n_p, D, l, dt(qkv), stream=_stream())
    return lm


def landmark_bwd(dlm: torch.Tensor, dqkv: torch.Tensor, l: int) -> None:
    _chk(dlm, dqkv)
    _contig(dlm, "dlm"), _contig(dqkv, "dqkv")
    B, n_p, D3 = dqkv.shape
    if dlm.dtype != dqkv.dtype or tuple(dlm.shape) != (B, n_p // l, 2 * (D3 // 3)):
        raise MirrorHipError("landmark_bwd: shape/dtype mismatch")
    _lib.call("mh_landmark_bwd", _p(dlm), _p(dqkv), B, n_p, D3 // 3, l, dt(dqkv), stream=_stream())


def resconv(v_src: torch.Tensor, w: torch.Tensor, out: torch.Tensor, heads: int, transpose: bool, accumulate: bool) -> None:
    """v_src / out: [B, n_p, C] views with unit last stride (e.g. the v column block of qkv)."""
    _chk(v_src, w, out)
    B, n_p, Cc = v_src.shape
    if tuple(out.shape) != (B, n_p, Cc) or v_src.stride(2) != 1 or out.stride(2) != 1:
        raise MirrorHipError("resconv: bad views")
    taps = w.numel() // heads
    _lib.call("mh_resconv_fwd", _p(v_src), v_src.stride(1), v_src.stride(0), _p(_contig(w, "res_conv weight")), _p(out),
              out.stride(1), out.stride(0), B, n_p, heads, Cc // heads, taps, int(transpose), int(accumulate), dt(v_src),
              dt(out), stream=_stream())


def resconv_wgrad(v_src: torch.Tensor, dout: torch.Tensor, dw: torch.Tensor, heads: int) -> None:
    _chk(v_src, dout, dw)
    B, n_p, Cc = v_src.shape
    if tuple(dout.shape) != (B, n_p, Cc) or v_src.stride(2) != 1 or dout.stride(2) != 1 or dw.dtype != torch.float32:
        raise MirrorHipError("resconv_wgrad: bad views")
    taps = dw.numel() // heads
    _lib.call("mh_resconv_wgrad", _p(v_src), v_src.stride(1), v_src.stride(0), _p(dout), dout.stride(1), dout.stride(0),
              _p(dw), B, n_p, heads, Cc // heads, taps, dt(v_src), dt(dout), stream=_stream())


def resconv_bwd(dout: torch.Tensor, v_src: torch.Tensor, w: torch.Tensor, dv: torch.Tensor, dw: torch.Tensor, heads: int) -> None:
    """Both gradients of res_conv in one pass over dout (mh_resconv_bwd): dv += conv^T(dout), dw += the tap gradient.  dout / v_src /
    dv: [B, n_p, C] views with unit last stride (v_src, dv: the v column blocks of qkv / d qkv); dw f32 [heads * taps]."""
    _chk(dout, v_src, w, dv, dw)
    B, n_p, Cc = v_src.shape
    if (tuple(dout.shape) != (B, n_p, Cc) or tuple(dv.shape) != (B, n_p, Cc) or v_src.stride(2) != 1 or dout.stride(2) != 1 or dv.stride(2) != 1
            or dw.dtype != torch.float32 or dv.dtype != v_src.dtype):
        raise MirrorHipError("resconv_bwd: bad views")
    taps = dw.numel() // heads
    nbytes = int(_lib.load().mh_resconv_bwd_workspace_bytes(B, n_p, heads, Cc // heads, taps))
    ws = torch.empty((max(nbytes // 4, 1),), device=dout.device, dtype=torch.float32)
    _lib.call("mh_resconv_bwd", _p(dout), dout.stride(1), dout.stride(0), _p(v_src), v_src.stride(1), v_src.stride(0),
              _p(_contig(w, "res_conv weight")), _p(dv), dv.stride(1), dv.stride(0), _p(dw), _p(ws), nbytes // 4, B, n_p, heads, Cc // heads, taps,
              dt(v_src), dt(dout), stream=_stream())


def pinv_absmax(x: torch.Tensor, stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """stats: optional zeroed int64[2] (the caller's pre-zeroed arena saves the fill launch)."""
    _chk(x)
    _contig(x, "pinv input")
    m = x.shape[-1]
    if stats is None:
        stats = torch.zeros(2, device=x.device, dtype=torch.int64)
    _lib.call("mh_pinv_absmax", _p(x), _p(stats), x.numel() // (m * m), m, stream=_stream())
    return stats


def pinv_z0(x: torch.Tensor, stats: torch.Tensor) -> torch.Tensor:
    m = x.shape[-1]
    z0 = torch.empty_like(x)
    _lib.call("mh_pinv_z0", _p(x), _p(stats), _p(z0), x.numel() // (m * m), m, stream=_stream())
    return z0


def _scratch1(scratch, like):
    """(one-float scratch, zeroed flag): a caller-provided f32[1] that already holds 0 (a slice of the step's zero arena), else a
    fresh one the entry point clears itself."""
    if scratch is None:
        return torch.empty(1, device=like.device, dtype=torch.float32), 0
    if scratch.dtype != torch.float32 or scratch.numel() != 1 or scratch.device != like.device:
        raise MirrorHipError("scratch: one f32 on the operands' device")
    return scratch, 1


def pinv_z0_bwd(x, z0, dz0, stats, dx, zeroed_scratch=None) -> None:
    """z0 None: formed on the fly from x and the maxima (nys_sim2 path: no stored f32 z_0)."""
    _chk(x, z0, dz0, stats, dx, zeroed_scratch)
    m = x.shape[-1]
    scratch, zf = _scratch1(zeroed_scratch, x)
    _lib.call("mh_pinv_z0_bwd", _p(x), _p(z0), _p(_contig(dz0, "dz0")), _p(stats), _p(dx), _p(scratch), zf,
              x.numel() // (m * m), m, stream=_stream())


def pinv_s2_bwd(p, dz0, stats, dx, zeroed_scratch=None, mlm=None, heads: int = 1) -> None:
    """dx (the chain's gradient wrt attn2 on entry) -> gradient wrt sim2's logits, one pass (mh_pinv_s2_bwd: m = 256).  mlm (f32 [B, m]) +
    heads: the valid-landmark flags of a key-padding mask (p [B, heads, m, m]): filled entries get no gradient."""
    _chk(p, dz0, stats, dx, zeroed_scratch, mlm)
    if mlm is not None and not (mlm.dtype == torch.float32 and mlm.is_contiguous() and mlm.numel() * heads * p.shape[-1] == p.numel()):
        raise MirrorHipError("pinv_s2_bwd: mlm is a contiguous f32 [B, m] with p [B, heads, m, m]")
    m = p.shape[-1]
    if not (p.dtype == dz0.dtype == dx.dtype == torch.float32 and p.is_contiguous() and dz0.is_contiguous() and dx.is_contiguous()
            and p.shape == dz0.shape == dx.shape and p.shape[-2] == m):
        raise MirrorHipError("pinv_s2_bwd: contiguous f32 [.., m, m] tensors of one shape")
    scratch, zf = _scratch1(zeroed_scratch, p)
    _lib.call("mh_pinv_s2_bwd", _p(p), _p(dz0), _p(stats), _p(dx), _p(scratch), zf, p.numel() // (m * m), m, _p(mlm), int(heads), stream=_stream())


PINV_CHAIN_M = 256


def pinv_chain_saved_alloc(iters: int, BH: int, m: int, device) -> torch.Tensor:
    """The chain's saved-iterate buffer (what mh_pinv_chain_fwd writes for mh_pinv_chain_bwd): [iters, 4, BH, m, m] bf16,
    slot k = (z_k, P_k, T2_k, T3_k), panel native."""
    nbytes = int(_lib.load().mh_pinv_chain_workspace_bytes(BH, m, iters, 0))
    if nbytes != iters * 4 * BH * m * m * 2:
        raise MirrorHipError(f"pinv chain: m={m} unsupported (mh_pinv_chain_workspace_bytes says {nbytes})")
    return torch.empty((iters, 4, BH, m, m), device=device, dtype=torch.bfloat16)


def pinv_chain_work_alloc(iters: int, BH: int, m: int, device) -> torch.Tensor:
    """The backward's scratch (mh_pinv_chain_workspace_bytes(which=1)): [iters + 4, BH, m, m] bf16."""
    n = int(_lib.load().mh_pinv_chain_workspace_bytes(BH, m, iters, 1)) // 2
    return torch.empty((n,), device=device, dtype=torch.bfloat16)


def pinv_chain_z0_slot(saved: torch.Tensor) -> torch.Tensor:
    """Where mh_pinv_chain_prep has to put the panel-native z_0."""
    return saved[0, 0]


def pinv_chain_prep(x: torch.Tensor, stats: torch.Tensor, z0cm_out: torch.Tensor):
    """attn2 (f32 [.., m, m]) -> (z0 f32 = x^T/(c r) row-major, XP = panel-native bf16 x); writes the panel-native z_0 into
    `z0cm_out` (normally saved[0, 0] of the chain).  Panel-native layout: include/mirror_hip.h."""
    _chk(x, stats, z0cm_out)
    _contig(x, "pinv_chain_prep input")
    m = x.shape[-1]
    BH = x.numel() // (m * m)
    if x.dtype != torch.float32 or z0cm_out.dtype != torch.bfloat16 or not z0cm_out.is_contiguous() or z0cm_out.numel() != x.numel():
        raise MirrorHipError("pinv_chain_prep: bad operands")
    z0 = torch.empty_like(x)
    xt = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    _lib.call("mh_pinv_chain_prep", _p(x), _p(stats), _p(z0), _p(xt), _p(z0cm_out), BH, m, stream=_stream())
    return z0, xt


def pinv_chain_pack(dz: torch.Tensor) -> torch.Tensor:
    """d z_iters (f32 row-major [.., m, m]) -> the panel-native bf16 input of pinv_chain_bwd."""
    _chk(dz)
    _contig(dz, "pinv_chain_pack input")
    m = dz.shape[-1]
    if dz.dtype != torch.float32:
        raise MirrorHipError("pinv_chain_pack: f32 input expected")
    up = torch.empty(dz.shape, device=dz.device, dtype=torch.bfloat16)
    _lib.call("mh_pinv_chain_pack", _p(dz), _p(up), dz.numel() // (m * m), m, stream=_stream())
    return up


def nys_dz_dav(dw2: torch.Tensor, av: torch.Tensor, zfT: torch.Tensor, want_delta3: bool = False, out=None):
    """(PN((dw2 av^T)^T) bf16 [.., m, m] for mh_pinv_chain_bwd, dAV = Z^T dw2 bf16 [.., m, dh]) in one launch (mh_nys_dz_dav);
    want_delta3: a third result, sum_d dAV av f32 [.., m], for nys_attn3_bwd(delta3=...).  out: the two bf16 result buffers, for a
    caller that launches on another stream than the one that owns the memory."""
    _chk(dw2, av, zfT)
    m, dh = dw2.shape[-2], dw2.shape[-1]
    BH = dw2.numel() // (m * dh)
    if not (dw2.dtype == av.dtype == torch.float32 and zfT.dtype == torch.bfloat16 and dw2.is_contiguous() and av.is_contiguous()
            and zfT.is_contiguous() and av.shape == dw2.shape and zfT.numel() == BH * m * m):
        raise MirrorHipError("nys_dz_dav: contiguous f32 [.., m, dh] gradients / products and the bf16 column-major chain output")
    if out is None:
        up = torch.empty(zfT.shape, device=dw2.device, dtype=torch.bfloat16)
        dav = torch.empty(dw2.shape, device=dw2.device, dtype=torch.bfloat16)
    else:
        up, dav = out
        _chk(up, dav)
        if not (up.dtype == dav.dtype == torch.bfloat16 and up.is_contiguous() and dav.is_contiguous() and up.numel() == zfT.numel()
                and dav.shape == dw2.shape):
            raise MirrorHipError("nys_dz_dav: out must be contiguous bf16 buffers shaped like zfT and dw2")
    delta3 = torch.empty(dw2.shape[:-1], device=dw2.device, dtype=torch.float32) if want_delta3 else None
    _lib.call("mh_nys_dz_dav", _p(dw2), _p(av), _p(zfT), _p(up), _p(dav), _p(delta3), BH, m, dh, stream=_stream())
    return (up, dav, delta3) if want_delta3 else (up, dav)


_NYS_SIM2 = True       # test hook: False = sim2 as GEMM + softmax + absmax + operand packing (tests/test_fused_epilogue_gpu.py)


def nys_sim2_ok(lm: torch.Tensor, heads: int) -> bool:
    """mh_nys_sim2's geometry: bf16 landmarks [B, 256, 2 D] with D = heads * 64."""
    return (_NYS_SIM2 and lm.dim() == 3 and lm.dtype == torch.bfloat16 and _lm_ld(lm) > 0 and lm.shape[1] == PINV_CHAIN_M
            and lm.shape[2] == 2 * heads * 64)


def nys_sim2_alloc(lm: torch.Tensor, heads: int):
    """(attn2, xt) buffers of nys_sim2, for a caller that launches it on another stream than the one that owns the memory."""
    Bn, m, _ = lm.shape
    return (torch.empty((Bn, heads, m, m), device=lm.device, dtype=torch.float32),
            torch.empty((Bn, heads, m, m), device=lm.device, dtype=torch.bfloat16))


def nys_sim2(lm: torch.Tensor, heads: int, scale: float, stats: Optional[torch.Tensor] = None, want_z0f: bool = False, out=None, mlm=None):
    """(attn2 f32 [B, h, m, m], xt = panel-native bf16 attn2, z0f, stats) in one launch.  z0f (want_z0f) = panel-native f32 attn2^T, the
    unscaled z_0; by default None: the chain forward forms z_0 from the ROWS of attn2 (pinv_chain_fwd(z0f=a2, z0_rowmajor=True)) and the
    launch skips its second pass.  out: (attn2, xt) from nys_sim2_alloc.  mlm (f32 [B, m]): valid-landmark flags of a key-padding mask
    (masked_fill in front of the softmax; no z0f then)."""
    _chk(lm, stats, mlm)
    if mlm is not None and (want_z0f or mlm.dtype != torch.float32 or not mlm.is_contiguous() or mlm.numel() != lm.shape[0] * lm.shape[1]):
        raise MirrorHipError("nys_sim2: mlm is a contiguous f32 [B, m] and excludes want_z0f")
    Bn, m, D2 = lm.shape
    D = D2 // 2
    if stats is None:
        stats = torch.zeros(2, device=lm.device, dtype=torch.int64)
    a2, xt = out if out is not None else nys_sim2_alloc(lm, heads)
    _chk(a2, xt)
    if not (a2.shape == xt.shape == (Bn, heads, m, m) and a2.dtype == torch.float32 and xt.dtype == torch.bfloat16
            and a2.is_contiguous() and xt.is_contiguous()):
        raise MirrorHipError("nys_sim2: out must be the contiguous (f32, bf16) [B, h, m, m] pair of nys_sim2_alloc")
    z0f = torch.empty((Bn, heads, m, m), device=lm.device, dtype=torch.float32) if want_z0f else None
    _lib.call("mh_nys_sim2", _p(lm), _p(a2), _p(xt), _p(z0f), _p(stats), Bn, m, D, heads, float(scale), _lm_ld(lm), _p(mlm), stream=_stream())
    return a2, xt, z0f, stats


def pinv_chain_fwd(XT: torch.Tensor, saved: torch.Tensor, zfT: torch.Tensor, iters: int, z0f: Optional[torch.Tensor] = None,
                   stats: Optional[torch.Tensor] = None, z0_rowmajor: bool = False) -> None:
    """XT = panel-native x, saved[0, 0] = panel-native z_0 (or z0f / stats from nys_sim2: the kernel then forms z_0 itself and
    writes saved[0, 0]); zfT receives the column-major z_iters (= pinv^T row-major)."""
    _chk(XT, saved, zfT, z0f, stats)
    m = XT.shape[-1]
    BH = XT.numel() // (m * m)
    bf = torch.bfloat16
    if not (XT.dtype == bf and saved.dtype == bf and zfT.dtype == bf and XT.is_contiguous() and saved.is_contiguous()
            and zfT.is_contiguous() and saved.numel() == iters * 4 * BH * m * m and zfT.numel() == BH * m * m):
        raise MirrorHipError("pinv_chain_fwd: bad operands")
    if z0f is not None and not (z0f.dtype == torch.float32 and z0f.is_contiguous() and z0f.numel() == BH * m * m and stats is not None):
        raise MirrorHipError("pinv_chain_fwd: bad z0f / stats")
    fn = lambda: _lib.call("mh_pinv_chain_fwd", _p(XT), _p(saved), _p(zfT), BH, m, iters, _p(z0f), _p(stats), int(z0_rowmajor), stream=_stream())  # noqa: E731
    if gemm_profiler is None:
        fn()
    else:
        gemm_profiler.launch_named("pinv_panel_fwd_kernel", iters * 4 * 2.0 * m ** 3 * BH, fn)


def pinv_chain_bwd(XT, saved, dzf, work, dX, dz0, iters: int) -> None:
    """dzf = pinv_chain_pack(d z_iters).  dX / dz0: row-major f32 outputs."""
    _chk(XT, saved, dzf, work, dX, dz0)
    m = XT.shape[-1]
    BH = XT.numel() // (m * m)
    bf = torch.bfloat16
    if not (all(t.dtype == bf and t.is_contiguous() for t in (XT, saved, dzf, work))
            and all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == BH * m * m for t in (dX, dz0))
            and saved.numel() == iters * 4 * BH * m * m and work.numel() >= (iters + 4) * BH * m * m and dzf.numel() == BH * m * m):
        raise MirrorHipError("pinv_chain_bwd: bad operands")
    fn = lambda: _lib.call("mh_pinv_chain_bwd", _p(XT), _p(saved), _p(dzf), _p(work), _p(dX), _p(dz0), BH, m, iters,  # noqa: E731
                           stream=_stream())
    if gemm_profiler is None:
        fn()
    else:
        gemm_profiler.launch_named("pinv_panel_bwd2_kernel", iters * 8 * 2.0 * m ** 3 * BH, fn)


# ------------------------------------------------------------------ fused Nystrom attention sides (nystrom_fused.hip)
NYS_FUSED_M, NYS_FUSED_DH = 256, 64


def nys_fused_ok(qkv: torch.Tensor, heads: int, m: int) -> bool:
    """The fused kernels cover the TransMIL geometry: bf16 qkv, 64-wide heads, 256 landmarks."""
    D = qkv.shape[-1] // 3
    return qkv.dtype == torch.bfloat16 and D == heads * NYS_FUSED_DH and m == NYS_FUSED_M and qkv.shape[1] % m == 0


def _nys_check(name: str, B: int, h: int, n_p: int, **tensors) -> None:
    D = h * NYS_FUSED_DH
    want = {"qkv": ((B, n_p, 3 * D), torch.bfloat16), "dqkv": ((B, n_p, 3 * D), torch.bfloat16),
            "lm": ((B, NYS_FUSED_M, 2 * D), torch.bfloat16), "dlm": ((B, NYS_FUSED_M, 2 * D), torch.float32),
            "w2": ((B, h, NYS_FUSED_M, NYS_FUSED_DH), torch.bfloat16), "dw2": ((B, h, NYS_FUSED_M, NYS_FUSED_DH), torch.float32),
            "av": ((B, h, NYS_FUSED_M, NYS_FUSED_DH), torch.float32), "dav": ((B, h, NYS_FUSED_M, NYS_FUSED_DH), torch.bfloat16),
            "out": ((B, n_p, D), torch.bfloat16), "dout": ((B, n_p, D), torch.bfloat16), "o1": ((B, n_p, D), torch.bfloat16),
            "lse1": ((B, h, n_p), torch.float32), "delta1": ((B, h, n_p), torch.float32),
            "lse3": ((B, h, NYS_FUSED_M), torch.float32)}
    for k, t in tensors.items():
        shape, dtype = want[k]
        if k == "lm" and tuple(t.shape) == shape and t.dtype == dtype and _lm_ld(t) > 0:
            continue                       # landmark rows may sit in a wider buffer (row stride lm_ld, see _lm_ld)
        if tuple(t.shape) != shape or t.dtype != dtype or not t.is_contiguous():
            raise MirrorHipError(f"{name}: {k} must be contiguous {dtype} {shape}, got {t.dtype} {tuple(t.shape)} "
                                 f"contiguous={t.is_contiguous()}")


def _lm_ld(lm: torch.Tensor) -> int:
    """Row stride (elements) of a landmark tensor [B, m, 2D] whose rows are unit-stride, evenly spaced and batch-contiguous
    (a contiguous tensor, or the landmark rows behind the sequence in to_qkv's [rows, 3D] output); 0 if it is neither."""
    if lm.dim() != 3 or lm.stride(2) != 1 or lm.stride(0) != lm.shape[1] * lm.stride(1) or lm.stride(1) < lm.shape[2] or lm.stride(1) % 8:
        return 0
    return int(lm.stride(1))


def _nys_launch(name: str, flops: float, fn) -> None:
    if gemm_profiler is None:
        fn()
    else:
        gemm_profiler.launch_named(name, flops, fn)


def _nys_masks(kmask, B: int, n_p: int):
    """(mrow ptr, mlm ptr) of the key-padding mask pair prepared by TransLayer, or (None, None)."""
    if kmask is None:
        return None, None
    mrow, mlm = kmask[0], kmask[1]
    _chk(mrow, mlm)
    assert mrow.shape == (B, n_p) and mlm.shape == (B, NYS_FUSED_M) and mrow.dtype == mlm.dtype == torch.float32
    assert mrow.is_contiguous() and mlm.is_contiguous()
    return _p(mrow), _p(mlm)


def nys_attn1_fwd(qkv, lm, w2, out, heads: int, scale: float, accumulate: bool = False, kmask=None, o1=None) -> torch.Tensor:
    """out[:, :, head] (+)= softmax_m(scale q k_l^T) w2; returns the row logsumexp [B, h, n_p].  kmask: key-padding mask.
    o1: optional bf16 buffer shaped like out that receives the product alone (what nys_attn1_bwd takes delta1 from)."""
    _chk(qkv, lm, w2, out, o1)
    B, n_p, _ = qkv.shape
    lse1 = torch.empty((B, heads, n_p), device=qkv.device, dtype=torch.float32)
    _nys_check("nys_attn1_fwd", B, heads, n_p, qkv=qkv, lm=lm, w2=w2, out=out, **({} if o1 is None else {"o1": o1}))
    _nys_launch("nys_a1_fwd_kernel", 2 * 2.0 * n_p * NYS_FUSED_M * NYS_FUSED_DH * B * heads,
                lambda: _lib.call("mh_nys_attn1_fwd", _p(qkv), _p(lm), _p(w2), _p(out), _p(lse1), *_nys_masks(kmask, B, n_p), B, heads,
                                  n_p, NYS_FUSED_M, NYS_FUSED_DH, scale, int(accumulate), _lm_ld(lm), _p(o1), stream=_stream()))
    return lse1


def nys_attn1_fwd_q8(qkv, lm, w2, out, heads: int, scale: float, accumulate: bool, q8, ring, tick, margin: float = 1.25, o1=None):
    """nys_attn1_fwd (no mask) that also writes the e4m3 copy of `out` into q8 (uint8, out's shape) with the delayed scale of
    `ring` / `tick`; returns (lse1, dequantisation factor)."""
    _chk(qkv, lm, w2, out, q8, ring, tick)
    B, n_p, _ = qkv.shape
    if not (q8.dtype == torch.uint8 and q8.shape == out.shape and q8.is_contiguous() and ring.dtype == torch.int32 and ring.numel() == 3
            and tick.dtype == torch.float32):
        raise MirrorHipError("nys_attn1_fwd_q8: q8 uint8 shaped like out, int32[3] ring, f32 tick")
    lse1 = torch.empty((B, heads, n_p), device=qkv.device, dtype=torch.float32)
    sc = torch.empty((1,), device=qkv.device, dtype=torch.float32)
    _chk(o1)
    _nys_check("nys_attn1_fwd", B, heads, n_p, qkv=qkv, lm=lm.contiguous(), w2=w2, out=out, **({} if o1 is None else {"o1": o1}))
    lm = lm.contiguous()
    _nys_launch("nys_a1_fwd_kernel", 2 * 2.0 * n_p * NYS_FUSED_M * NYS_FUSED_DH * B * heads,
                lambda: _lib.call("mh_nys_attn1_fwd_q8", _p(qkv), _p(lm), _p(w2), _p(out), _p(lse1), B, heads, n_p, NYS_FUSED_M,
                                  NYS_FUSED_DH, scale, int(accumulate), _p(q8), _p(ring), _p(tick), float(margin), _p(sc), _p(o1),
                                  stream=_stream()))
    return lse1, sc


def nys_attn3_fwd(qkv, lm, heads: int, scale: float, kmask=None, rc=None):
    """av = softmax_n(scale q_l k^T) v as [B, h, m, dh] f32, and the row logsumexp [B, h, m].
    rc = (res_w f32 [h * 33] contiguous, out bf16 [B, n_p, D] contiguous): the same launch also writes out = res_conv(v)."""
    _chk(qkv, lm)
    B, n_p, _ = qkv.shape
    rc_w = rc_out = None
    if rc is not None:
        rc_w, rc_out = rc
        _chk(rc_w, rc_out)
        if (rc_w.dtype != torch.float32 or rc_w.numel() != heads * 33 or not rc_w.is_contiguous() or rc_out.dtype != torch.bfloat16
                or tuple(rc_out.shape) != (B, n_p, heads * NYS_FUSED_DH) or not rc_out.is_contiguous()):
            raise MirrorHipError("nys_attn3_fwd: rc = (f32 [h * 33] filters, contiguous bf16 [B, n_p, D] output)")
    av = torch.empty((B, heads, NYS_FUSED_M, NYS_FUSED_DH), device=qkv.device, dtype=torch.float32)
    lse3 = torch.empty((B, heads, NYS_FUSED_M), device=qkv.device, dtype=torch.float32)
    _nys_check("nys_attn3_fwd", B, heads, n_p, qkv=qkv, lm=lm)
    nws = int(_lib.load().mh_nys_attn3_ws_floats(B, heads, n_p))     # partial results of the sequence ranges
    ws = torch.empty((nws,), device=qkv.device, dtype=torch.float32) if nws else None
    _nys_launch("nys_a3_fwd_kernel", 2 * 2.0 * n_p * NYS_FUSED_M * NYS_FUSED_DH * B * heads,
                lambda: _lib.call("mh_nys_attn3_fwd", _p(qkv), _p(lm), _p(av), _p(lse3), _p(ws), nws, *_nys_masks(kmask, B, n_p), B,
                                  heads, n_p, NYS_FUSED_M, NYS_FUSED_DH, scale, _lm_ld(lm), _p(rc_w), _p(rc_out), stream=_stream()))
    return av, lse3


def nys_attn1_bwd(qkv, lm, w2, dout, lse1, o1, delta1, dqkv, dw2, dlm, heads: int, scale: float, kmask=None, which: int = 3) -> None:
    """attn1's backward in two parts (mh_nys_attn1_bwd): which & 1 — ADDS into dw2 and into the k_l half of dlm (both f32, zeroed by the
    caller) and WRITES delta1 [B, h, n_p] from the forward's saved rows o1; which & 2 — writes the q block of dqkv from delta1."""
    _chk(qkv, lm, w2, dout, lse1, o1, delta1, dqkv, dw2, dlm)
    B, n_p, _ = qkv.shape
    ts = dict(qkv=qkv, lm=lm, w2=w2, dout=dout, lse1=lse1, delta1=delta1)
    if which & 1:
        ts.update(o1=o1, dw2=dw2, dlm=dlm)
    if which & 2:
        ts.update(dqkv=dqkv)
    _nys_check("nys_attn1_bwd", B, heads, n_p, **ts)
    prods = (4 if which & 1 else 0) + (3 if which & 2 else 0)
    _nys_launch("nys_a1_bwd_kernels", prods * 2.0 * n_p * NYS_FUSED_M * NYS_FUSED_DH * B * heads,
                lambda: _lib.call("mh_nys_attn1_bwd", _p(qkv), _p(lm), _p(w2), _p(dout), _p(lse1), _p(o1), _p(delta1), _p(dqkv),
                                  _p(dw2), _p(dlm), *_nys_masks(kmask, B, n_p), B, heads, n_p, NYS_FUSED_M, NYS_FUSED_DH, scale,
                                  _lm_ld(lm), int(which), stream=_stream()))


NYS_A3_BWD_ONE_PASS = True      # (test hook, round 5) attn3's backward as one kernel (-1.21 % +- 0.17 step time against the dk / dv + dq_l pair)


def nys_attn3_bwd(qkv, lm, av, dav, lse3, dqkv, dlm, heads: int, scale: float, kmask=None, delta3=None, one_pass=None) -> None:
    """Writes the k and v blocks of dqkv; ADDS into the q_l half of dlm.  delta3: sum_d dav av from nys_dz_dav (skips a launch).
    one_pass: dk, dv, dq_l from one kernel (default) or the dk / dv kernel + the dq_l kernel."""
    one_pass = NYS_A3_BWD_ONE_PASS if one_pass is None else bool(one_pass)
    _chk(qkv, lm, av, dav, lse3, dqkv, dlm, delta3)
    B, n_p, _ = qkv.shape
    _nys_check("nys_attn3_bwd", B, heads, n_p, qkv=qkv, lm=lm, av=av, dav=dav, lse3=lse3, dqkv=dqkv, dlm=dlm)
    given = delta3 is not None
    if given and not (delta3.dtype == torch.float32 and delta3.is_contiguous() and delta3.numel() == lse3.numel()):
        raise MirrorHipError("nys_attn3_bwd: delta3 must be contiguous f32 [B, h, m]")
    if not given:
        delta3 = torch.empty_like(lse3)
    _nys_launch("nys_a3_bwd_kernels", 7 * 2.0 * n_p * NYS_FUSED_M * NYS_FUSED_DH * B * heads,
                lambda: _lib.call("mh_nys_attn3_bwd", _p(qkv), _p(lm), None if given else _p(av), _p(dav), _p(lse3), _p(delta3), _p(dqkv), _p(dlm),
                                  *_nys_masks(kmask, B, n_p), B, heads, n_p, NYS_FUSED_M, NYS_FUSED_DH, scale, _lm_ld(lm), int(one_pass),
                                  stream=_stream()))


def eye_minus(P: torch.Tensor, d: float) -> torch.Tensor:
    _chk(P)
    _contig(P, "eye_minus input")
    m = P.shape[-1]
    T = torch.empty_like(P)
    _lib.call("mh_eye_minus", _p(P), _p(T), d, P.numel() // (m * m), m, stream=_stream())
    return T


def seq_finish(seq: torch.Tensor, cls: torch.Tensor, N: int, add: int) -> None:
    _chk(seq, cls)
    B, n, D = seq.shape
    assert n == 1 + N + add and seq.is_contiguous() and cls.numel() == D and cls.dtype == torch.float32
    _lib.call("mh_seq_finish", _p(seq), _p(cls), B, N, add, D, dt(seq), stream=_stream())


def seq_finish_bwd(dseq: torch.Tensor, dcls: torch.Tensor, N: int, add: int) -> None:
    _chk(dseq, dcls)
    B, n, D = dseq.shape
    assert n == 1 + N + add and dseq.is_contiguous() and dcls.numel() == D and dcls.dtype == torch.float32
    _lib.call("mh_seq_finish_bwd", _p(dseq), _p(dcls), B, N, add, D, dt(dseq), stream=_stream())


def ppeg_merge(w7, w5, w3, b7, b5, b3):
    _chk(w7, w5, w3, b7, b5, b3)
    D = b7.numel()
    merged = torch.empty((49, D), device=w7.device, dtype=torch.float32)
    bsum = torch.empty((D,), device=w7.device, dtype=torch.float32)
    _lib.call("mh_ppeg_merge", _p(_contig(w7, "w7")), _p(_contig(w5, "w5")), _p(_contig(w3, "w3")), _p(b7), _p(b5), _p(b3),
              _p(merged), _p(bsum), D, stream=_stream())
    return merged, bsum


def ppeg(x: torch.Tensor, merged, bsum, S: int, flip: bool, out_dtype=None) -> torch.Tensor:
    _chk(x, merged, bsum)
    B, n, D = x.shape
    assert n == 1 + S * S and x.is_contiguous()
    y = torch.empty_like(x, dtype=out_dtype or x.dtype)
    _lib.call("mh_ppeg_fwd", _p(x), _p(y), _p(merged), _p(bsum), B, S, D, int(flip), dt(x), dt(y), stream=_stream())
    return y


def ppeg_wgrad(x, dout, dmerged, dbsum, S: int) -> None:
    _chk(x, dout, dmerged, dbsum)
    B, n, D = x.shape
    assert n == 1 + S * S and x.is_contiguous() and dout.is_contiguous() and dout.shape == x.shape
    _lib.call("mh_ppeg_wgrad", _p(x), _p(dout), _p(dmerged), _p(dbsum), B, S, D, dt(x), dt(dout), stream=_stream())


def ppeg_grad_scatter(dmerged, dbsum, dw7, dw5, dw3, db7, db5, db3) -> None:
    """The six PPEG parameter gradients += their share of the merged kernel's gradient (dmerged [49, D], dbsum [D])."""
    D = dbsum.numel()
    outs = (dw7, dw5, dw3, db7, db5, db3)
    _chk(dmerged, dbsum, *outs)
    for t, n in zip((dmerged,) + outs, (49 * D, 49 * D, 25 * D, 9 * D, D, D, D)):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise MirrorHipError(f"ppeg_grad_scatter: expected a contiguous f32 tensor of {n} elements, got {tuple(t.shape)} {t.dtype}")
    _lib.call("mh_ppeg_grad_scatter", _p(dmerged), _p(dbsum), *[_p(t) for t in outs], D, stream=_stream())


# ----------------------------------------------------------------------------- data feed
def gather_rows(src: torch.Tensor, rows: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r] = src[rows[r]] for a [S, F] bank and int64 row indices of any shape (result: rows.shape + (F,))."""
    _chk(src, rows)
    if src.dim() != 2 or not src.is_contiguous():
        raise MirrorHipError("gather_rows: the bank must be a contiguous [rows, F] tensor")
    if rows.dtype != torch.int64 or not rows.is_contiguous():
        raise MirrorHipError("gather_rows: row indices must be contiguous int64")
    R, Fd = rows.numel(), src.shape[1]
    if out is None:
        out = torch.empty(tuple(rows.shape) + (Fd,), device=src.device, dtype=src.dtype)
    assert out.is_contiguous() and out.numel() == R * Fd and out.dtype == src.dtype
    _lib.call("mh_gather_rows", _p(src), _p(rows), _p(out), R, Fd, src.shape[0], dt(src), stream=_stream())
    return out


# ----------------------------------------------------------------------------- masking
def rank_mask(noise: torch.Tensor, len_keep: int) -> torch.Tensor:
    _chk(noise)
    noise = _contig(noise.float(), "noise")
    B, N = noise.shape
    mask = torch.empty_like(noise)
    _lib.call("mh_rank_mask", _p(noise), _p(mask), B, N, len_keep, stream=_stream())
    return mask


def mask_apply_fwd(x, mask, token, pos, B, T, D, first, token_scalar, out=None) -> None:
    """out (same shape as x, any dtype) receives the result; in place when omitted."""
    out = x if out is None else out
    _chk(x, mask, token, pos, out)
    assert x.is_contiguous() and x.numel() == B * T * D and mask.numel() == B * (T - first) and pos.numel() == T * D
    assert out.is_contiguous() and out.numel() == x.numel()
    _lib.call("mh_mask_apply_fwd", _p(x), _p(out), _p(mask), _p(token), _p(pos), B, T, D, first, int(token_scalar), dt(x), dt(out),
              stream=_stream())


def mask_apply_bwd_dbias_ok(dy, out, dpos, D) -> bool:
    """Can mask_apply_bwd leave the column sums of dx too (the bias gradient of the Linear in front)?"""
    return bool(_lib.load().mh_mask_apply_bwd_dbias_ok(_p(dy), _p(out), _p(dpos), D, dt(dy), dt(out)))


def mask_apply_bwd(dy, mask, dtoken, dpos, B, T, D, first, token_scalar, out=None, dbias=None) -> None:
    """out (same shape as dy, any dtype) receives dx; in place when omitted.  dbias [D] f32: += column sums of dx."""
    out = dy if out is None else out
    _chk(dy, mask, dtoken, dpos, out)
    assert dy.is_contiguous() and dy.numel() == B * T * D and dpos.numel() == T * D
    assert out.is_contiguous() and out.numel() == dy.numel()
    if dbias is not None:
        _chk(dbias)
        assert dbias.dtype == torch.float32 and dbias.numel() == D and dbias.is_contiguous()
    _lib.call("mh_mask_apply_bwd", _p(dy), _p(out), _p(mask), _p(dtoken), _p(dpos), B, T, D, first, int(token_scalar), dt(dy),
              dt(out), _p(dbias), stream=_stream())


# ----------------------------------------------------------------------------- RNA attention
def headattn_fwd(qkv: torch.Tensor, H: int):
    _chk(qkv)
    _contig(qkv, "qkv")
    B, D3 = qkv.shape
    D = D3 // 3
    out = torch.empty((B, D), device=qkv.device, dtype=qkv.dtype)
    attn = torch.empty((B, H, H), device=qkv.device, dtype=torch.float32)
    _lib.call("mh_headattn_fwd", _p(qkv), _p(out), _p(attn), B, H, D // H, dt(qkv), stream=_stream())
    return out, attn


def headattn_bwd(qkv, attn, dout, H: int) -> torch.Tensor:
    _chk(qkv, attn, dout)
    B, D3 = qkv.shape
    dqkv = torch.empty_like(qkv)
    dout = _contig(dout, "dout")
    if dout.dtype != qkv.dtype:
        raise MirrorHipError("headattn_bwd: dtype mismatch")
    _lib.call("mh_headattn_bwd", _p(qkv), _p(attn), _p(dout), _p(dqkv), B, H, D3 // 3 // H, dt(qkv), stream=_stream())
    return dqkv


# ----------------------------------------------------------------------------- fused RNA Block (csrc/rna_block.hip)
def rna_block_ok(x: torch.Tensor, D: int, Hh: int, H: int) -> bool:
    """The fused Block covers [B <= 32, D] f32 rows with D, Hh multiples of 32 and H | D (c2: D = 512, Hh = 2048, H = 8)."""
    if not (x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous() and 1 <= x.shape[0] <= 32 and x.shape[1] == D
            and D % 32 == 0 and Hh % 32 == 0 and D % H == 0 and D <= 2048 and Hh <= 4096 and H <= 64):
        return False
    # the kernels stage the [B x K] operand of every Linear in LDS (csrc/rna_block.hip fwd_lds / launch_bwd): MT * 16 rows of
    # K + 8 bf16 plus the cross-wave reduction scratch must fit 158 KiB, K = the longest contraction = max(D, Hh)
    mt = 1 if x.shape[0] <= 16 else 2
    return mt * 16 * (max(D, Hh) + 8) * 2 + 4 * mt * 16 * 17 * 4 <= 158 * 1024


def rna_block_workspace_bytes(B: int, D: int, Hh: int) -> int:
    return int(_lib.load().mh_rna_block_workspace_bytes(B, D, Hh))


def _rna_desc(B, D, Hh, H, eps, p, seed, offset, dev_base, **ptrs) -> "_lib.RnaBlockDesc":
    d = _lib.RnaBlockDesc()
    d.B, d.D, d.Hh, d.H, d.eps, d.p_drop, d.seed, d.offset = B, D, Hh, H, eps, p, seed, offset
    d.dev_base = _p(dev_base)
    for k, v in ptrs.items():
        _chk(v)
        setattr(d, k, _p(v))
    return d


def rna_block_fwd(x, params: dict, H: int, eps: float, p: float, seed: int, offset: int, dev_base):
    """params: bf16 weights w_qkv [3D, D], w_proj [D, D], w_fc1 [Hh, D], w_fc2 [D, Hh]; f32 b_qkv (or None), b_proj, b_fc1,
    b_fc2, g1, be1, g2, be2.  Returns (y, saved) with saved = the tensors mh_rna_block_bwd reads."""
    import ctypes
    B, D = x.shape
    Hh = params["w_fc1"].shape[0]
    dev, bf, f = x.device, torch.bfloat16, torch.float32
    for k in ("w_qkv", "w_proj", "w_fc1", "w_fc2"):
        if params[k].dtype != bf or not params[k].is_contiguous():
            raise MirrorHipError(f"rna_block_fwd: {k} must be a contiguous bf16 matrix")
    saved = {"stats": torch.empty((4, B), device=dev, dtype=f), "qkv": torch.empty((B, 3 * D), device=dev, dtype=bf),
             "attn": torch.empty((B, H, H), device=dev, dtype=f), "o": torch.empty((B, D), device=dev, dtype=bf),
             "x1": torch.empty((B, D), device=dev, dtype=f), "u": torch.empty((B, Hh), device=dev, dtype=bf),
             "f": torch.empty((B, Hh), device=dev, dtype=bf)}
    y = torch.empty((B, D), device=dev, dtype=f)
    d = _rna_desc(B, D, Hh, H, eps, p, seed, offset, dev_base, x=x, y=y, **params, **saved)
    _lib.call("mh_rna_block_fwd", ctypes.byref(d), stream=_stream())
    return y, saved


def rna_block_bwd(x, dy, params: dict, params_t: dict, grads: dict, saved: dict, H: int, eps: float, p: float, seed: int, offset: int,
                  dev_base) -> torch.Tensor:
    """params_t: wt_qkv ... (bf16 transposes); grads: f32 accumulation buffers dw_* / db_* / dg1 / dbe1 / dg2 / dbe2."""
    import ctypes
    B, D = x.shape
    Hh = params["w_fc1"].shape[0]
    if dy.dtype != torch.float32 or not dy.is_contiguous() or dy.shape != x.shape:
        raise MirrorHipError("rna_block_bwd: dy must be a contiguous f32 [B, D] tensor")
    dx = torch.empty_like(x)
    scratch = torch.empty((rna_block_workspace_bytes(B, D, Hh),), device=x.device, dtype=torch.uint8)
    d = _rna_desc(B, D, Hh, H, eps, p, seed, offset, dev_base, x=x, dy=dy, dx=dx, scratch=scratch, **params, **params_t, **grads, **saved)
    _lib.call("mh_rna_block_bwd", ctypes.byref(d), stream=_stream())
    return dx


# ----------------------------------------------------------------------------- elementwise
def add(a: torch.Tensor, b: torch.Tensor, out_dtype=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(a, b, out)
    _contig(a, "add a"), _contig(b, "add b")
    if a.shape != b.shape:
        raise MirrorHipError(f"add: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
    y = out if out is not None else torch.empty_like(a, dtype=out_dtype or a.dtype)
    _lib.call("mh_add", _p(a), _p(b), _p(y), a.numel(), dt(a), dt(b), dt(y), stream=_stream())
    return y


def lm_merge(a: torch.Tensor, b: Optional[torch.Tensor], out2: torch.Tensor, zero_cols: int) -> None:
    """out2[:, :cols] = bf16(a + b), out2[:, cols:cols + zero_cols] = 0 for f32 a / b [rows, cols] (b may be None) and a bf16 2-D
    view out2 [rows, cols + zero_cols] with unit column stride (mh_lm_merge): the landmark gradient as rows of to_qkv's gradient buffer."""
    _chk(a, b, out2)
    rows, cols = a.numel() // a.shape[-1], a.shape[-1]
    if not (a.dtype == torch.float32 and a.is_contiguous() and (b is None or (b.dtype == torch.float32 and b.is_contiguous() and b.numel() == a.numel()))
            and out2.dim() == 2 and out2.dtype == torch.bfloat16 and out2.stride(1) == 1 and tuple(out2.shape) == (rows, cols + zero_cols)):
        raise MirrorHipError("lm_merge: f32 contiguous a / b, bf16 out2 [rows, cols + zero_cols] with unit column stride")
    _lib.call("mh_lm_merge", _p(a), _p(b), _p(out2), rows, cols, out2.stride(0), int(zero_cols), stream=_stream())


def cast(x: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if x.dtype == dtype and out is None:
        return x
    _chk(x, out)
    _contig(x, "cast input")
    y = torch.empty_like(x, dtype=dtype) if out is None else _contig(out, "cast output")
    if y.numel() != x.numel():
        raise MirrorHipError("cast: size mismatch")
    _lib.call("mh_cast", _p(x), _p(y), x.numel(), dt(x), dt(y), stream=_stream())
    return y


def gelu_fwd(x: torch.Tensor, out_dtype=None) -> torch.Tensor:
    _chk(x)
    _contig(x, "gelu input")
    y = torch.empty_like(x, dtype=out_dtype or x.dtype)
    _lib.call("mh_gelu_fwd", _p(x), _p(y), x.numel(), dt(x), dt(y), stream=_stream())
    return y


def gelu_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _chk(x, dy)
    dy = _contig(dy, "gelu dy")
    dx = torch.empty_like(dy)
    _lib.call("mh_gelu_bwd", _p(x), _p(dy), _p(dx), x.numel(), dt(x), dt(dy), dt(dx), stream=_stream())
    return dx


def relu_bwd(y: torch.Tensor, dy: torch.Tensor, out_dtype=None) -> torch.Tensor:
    """dx = dy * (y > 0).  y / dy: same shape; either contiguous, or 3-D with a contiguous [R, D] block per batch."""
    _chk(y, dy)
    if y.shape != dy.shape:
        raise MirrorHipError("relu_bwd: shape mismatch")
    dx = torch.empty(y.shape, device=y.device, dtype=out_dtype or dy.dtype)

    def blk(t):
        if t.is_contiguous():
            return 1, t.numel(), 0
        if t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == t.shape[2]:
            return t.shape[0], t.shape[1] * t.shape[2], t.stride(0)
        raise MirrorHipError(f"relu_bwd: unsupported strides {t.stride()}")

    by, ny, sy = blk(y)
    bd, nd, sd = blk(dy)
    batches = max(by, bd)
    npb = y.numel() // batches
    if by == 1:
        sy = npb
    if bd == 1:
        sd = npb
    _lib.call("mh_relu_bwd", _p(y), _p(dy), _p(dx), npb, batches, sy, sd, npb, dt(y), dt(dy), dt(dx), stream=_stream())
    return dx


def dropout(x: torch.Tensor, p: float, seed: int, offset: int, out: Optional[torch.Tensor] = None,
            dev_base: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dev_base: optional int64[1] device tensor added to `offset` on the device (per-step base of a graphed step)."""
    _chk(x, out, dev_base)
    _contig(x, "dropout input")
    y = out if out is not None else torch.empty_like(x)
    _lib.call("mh_dropout", _p(x), _p(y), x.numel(), p, seed, offset, _p(dev_base), dt(x), dt(y), stream=_stream())
    return y


def gemm_w4(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(experiment) bf16 [M, N] = a [M, K] @ w[N, K]^T + bias on the four-wave 256 x 256 tile kernel (mh_gemm_w4)."""
    _chk(a, w, bias, out)
    M, Kd = a.shape
    N = w.shape[0]
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == Kd
    y = out if out is not None else torch.empty((M, N), device=a.device, dtype=torch.bfloat16)
    _lib.call("mh_gemm_w4", _p(a), _p(w), _p(y), _p(bias), M, N, Kd, a.stride(0), w.stride(0), y.stride(0), stream=_stream())
    return y


def noise_draws(n_uniform: int, n_normal: int, seed: int, offset: int, dev_base: Optional[torch.Tensor], device) -> torch.Tensor:
    """f32 [n_uniform + n_normal]: uniform [0, 1) then standard normal draws on the dropout stream (mh_noise_draws); n_uniform % 4 == 0."""
    _chk(dev_base)
    assert n_uniform % 4 == 0 and offset % 4 == 0
    out = torch.empty((n_uniform + n_normal,), device=device, dtype=torch.float32)
    _chk(out)
    _lib.call("mh_noise_draws", _p(out), n_uniform, n_normal, seed, offset, _p(dev_base), stream=_stream())
    return out


def dropout_add(a: torch.Tensor, x: torch.Tensor, p: float, seed: int, offset: int, dev_base: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a + dropout(x) as f32 (a f32, x f32 / bf16, same shape, numel % 4 == 0)."""
    _chk(a, x, dev_base)
    _contig(a, "dropout_add a"), _contig(x, "dropout_add x")
    assert a.dtype == torch.float32 and a.shape == x.shape and x.numel() % 4 == 0
    y = torch.empty_like(a)
    _lib.call("mh_dropout_add", _p(a), _p(x), _p(y), x.numel(), p, seed, offset, _p(dev_base), dt(x), stream=_stream())
    return y


def dropout_lite_colsum(x: torch.Tensor, p: float, seed: int, offset: int, dev_base, out: torch.Tensor, db: torch.Tensor) -> torch.Tensor:
    """out (bf16) = dropout-backward of the f32 gradient x [.., N] on the lite stream, db [N] += column sums of out (mh_dropout_lite_colsum)."""
    _chk(x, dev_base, out, db)
    N = x.shape[-1]
    if not (x.dtype == torch.float32 and out.dtype == torch.bfloat16 and x.is_contiguous() and out.is_contiguous() and out.shape == x.shape
            and db.dtype == torch.float32 and db.is_contiguous() and db.numel() == N and offset % 8 == 0):
        raise MirrorHipError("dropout_lite_colsum: contiguous f32 gradient, bf16 output of the same shape, f32 [N] bias gradient")
    _lib.call("mh_dropout_lite_colsum", _p(x), _p(out), x.numel() // N, N, p, seed, offset, _p(dev_base), _p(db), stream=_stream())
    return out


def dropout_lite_colsum_ok(N: int) -> bool:
    return N % 8 == 0 and 8 <= N <= 2048 and 256 % (N // 8) == 0


def timestamp(slot: torch.Tensor) -> None:
    """*slot (int64, one element) = the device wall clock when the current stream reaches this point (mh_timestamp: profiling aid)."""
    assert slot.dtype == torch.int64 and slot.numel() == 1 and slot.is_cuda
    _lib.call("mh_timestamp", _p(slot), stream=_stream())


def dropout_lite(x: torch.Tensor, p: float, seed: int, offset: int, dev_base: Optional[torch.Tensor] = None, *, add_to: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dropout(x) (or add_to + dropout(x) as f32) on the lite stream (mh_dropout_lite: Philox4x32-7, 16 bits per element);
    numel % 8 == 0, offset % 8 == 0.  The DROPADD projection epilogue draws the same masks."""
    _chk(x, dev_base, add_to, out)
    _contig(x, "dropout_lite x")
    if x.numel() % 8 or offset % 8:
        raise MirrorHipError("dropout_lite: numel and offset must be multiples of 8")
    if add_to is not None:
        _contig(add_to, "dropout_lite add_to")
        assert add_to.dtype == torch.float32 and add_to.shape == x.shape
        out = torch.empty_like(add_to) if out is None else out
    elif out is None:
        out = torch.empty_like(x)
    assert out.is_contiguous() and out.numel() == x.numel()
    _lib.call("mh_dropout_lite", _p(add_to), _p(x), _p(out), x.numel(), p, seed, offset, _p(dev_base), dt(x), dt(out), stream=_stream())
    return out


def colsum(x2d: torch.Tensor, out: torch.Tensor) -> None:
    """out[c] += sum_r x2d[r, c]; x2d may be row-strided."""
    _chk(x2d, out)
    rows, cols = x2d.shape
    if x2d.stride(1) != 1 and cols > 1:
        raise MirrorHipError("colsum: unit column stride required")
    assert out.dtype == torch.float32 and out.numel() == cols
    _lib.call("mh_colsum", _p(x2d), _p(out), rows, cols, x2d.stride(0) if rows > 1 else cols, dt(x2d), stream=_stream())


def exp_fwd(x: torch.Tensor) -> torch.Tensor:
    _chk(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise MirrorHipError("exp_fwd: contiguous f32")
    y = torch.empty_like(x)
    _lib.call("mh_exp_fwd", _p(x), _p(y), x.numel(), stream=_stream())
    return y


def exp_bwd(dy: torch.Tensor, y: torch.Tensor, dx: torch.Tensor, accumulate: bool) -> None:
    _chk(dy, y, dx)
    if not all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == y.numel() for t in (dy, y, dx)):
        raise MirrorHipError("exp_bwd: contiguous f32 tensors of one size")
    _lib.call("mh_exp_bwd", _p(dy), _p(y), _p(dx), y.numel(), int(accumulate), stream=_stream())


def reparam_fwd(mu, logstd, eps) -> torch.Tensor:
    _chk(mu, logstd, eps)
    z = torch.empty_like(mu)
    _lib.call("mh_reparam_fwd", _p(_contig(mu, "mu")), _p(_contig(logstd, "logstd")), _p(_contig(eps, "eps")), _p(z),
              mu.numel(), stream=_stream())
    return z


def reparam_bwd(logstd, eps, dz, add_mu=None, add_ls=None):
    """(dmu, dlogstd) of z = mu + eps * exp(0.5 * logstd); add_mu / add_ls: gradients of mu / logstd from elsewhere, summed in."""
    _chk(logstd, eps, dz, add_mu, add_ls)
    for a in (add_mu, add_ls):
        if a is not None and (a.dtype != torch.float32 or not a.is_contiguous() or a.numel() != logstd.numel()):
            raise MirrorHipError("reparam_bwd: bad addend")
    dmu, dls = torch.empty_like(logstd), torch.empty_like(logstd)
    _lib.call("mh_reparam_bwd", _p(logstd), _p(eps), _p(_contig(dz, "dz")), _p(add_mu), _p(add_ls), _p(dmu), _p(dls), logstd.numel(),
              stream=_stream())
    return dmu, dls


# ----------------------------------------------------------------------------- losses / step glue
def ce_rows_fwd(G, scale, scale_mul, label_off, coef, out, loss_rows=None):
    _chk(G, scale, out, loss_rows)
    R, Cc = G.shape
    assert G.dtype == torch.float32 and G.stride(1) == 1
    lse = torch.empty((R,), device=G.device, dtype=torch.float32)
    _lib.call("mh_ce_rows_fwd", _p(G), G.stride(0), _p(scale), scale_mul, R, Cc, label_off, coef, _p(loss_rows), _p(lse),
              _p(out), stream=_stream())
    return lse


def ce_rows_bwd(G, scale, scale_mul, lse, g, g_per_row, gcoef, dscale, label_off):
    _chk(G, scale, lse, g, dscale)
    R, Cc = G.shape
    dG = torch.empty((R, Cc), device=G.device, dtype=torch.float32)
    _lib.call("mh_ce_rows_bwd", _p(G), G.stride(0), _p(scale), scale_mul, _p(lse), _p(g), int(g_per_row), gcoef, _p(dG),
              _p(dscale), R, Cc, label_off, stream=_stream())
    return dG


def _tgt_rows(tgt: torch.Tensor, rows: int, D: int):
    """(rows_per_batch, batch stride in elements) of a target that is contiguous or a row window [B, R, D] of a larger buffer."""
    if tgt.is_contiguous():
        return rows, rows * D
    if tgt.dim() == 3 and tgt.stride(2) == 1 and tgt.stride(1) == D and tgt.shape[2] == D:
        return tgt.shape[1], tgt.stride(0)
    raise MirrorHipError(f"masked-MSE target must be contiguous or a row window, got strides {tgt.stride()}")


def mse_masked_fwd(pred, tgt, mask, acc, rows, D):
    _chk(pred, tgt, mask, acc)
    assert pred.is_contiguous() and tgt.numel() == pred.numel()
    rpb, tbs = _tgt_rows(tgt, rows, D)
    _lib.call("mh_mse_masked_fwd", _p(pred), _p(tgt), _p(mask), _p(acc), rows, D, rpb, tbs, dt(pred), dt(tgt), stream=_stream())


MSE_CS_BLOCKS = 1024      # blocks (= rows of the column-sum table) of mse_masked_bwd(colsum_ws=...)


def mse_masked_bwd_colsum_ok(pred, tgt, dpred, D: int) -> bool:
    return (D in (256, 512, 768, 1024) and pred.dtype == torch.bfloat16 and tgt.dtype == torch.float32 and dpred.dtype == torch.bfloat16
            and tgt.stride(0) % 4 == 0 and tgt.data_ptr() % 16 == 0 and pred.data_ptr() % 8 == 0 and dpred.data_ptr() % 8 == 0)


def mse_masked_bwd(pred, tgt, mask, acc, g, dpred, dtgt, rows, D, gmul: float = 1.0, colsum_ws=None):
    """dtgt None: the target's gradient (-dpred) is not materialised.  The upstream is g[0] * gmul.
    colsum_ws [blocks, D] f32: every block's column sums of dpred (colsum(colsum_ws, db) is then the bias gradient in front of pred)."""
    _chk(pred, tgt, mask, acc, g, dpred, *([] if dtgt is None else [dtgt]))
    assert pred.is_contiguous() and dpred.is_contiguous() and (dtgt is None or (dtgt.is_contiguous() and dtgt.dtype == tgt.dtype))
    rpb, tbs = _tgt_rows(tgt, rows, D)
    nb = 0
    if colsum_ws is not None:
        _chk(colsum_ws)
        assert colsum_ws.dtype == torch.float32 and colsum_ws.is_contiguous() and colsum_ws.dim() == 2 and colsum_ws.shape[1] == D
        nb = colsum_ws.shape[0]
    _lib.call("mh_mse_masked_bwd", _p(pred), _p(tgt), _p(mask), _p(acc), _p(g), float(gmul), _p(dpred), _p(dtgt), rows, D, rpb, tbs, dt(pred),
              dt(tgt), dt(dpred), _p(colsum_ws), nb, stream=_stream())


LOSS_TERMS_BMAX = 32          # alignment block of mh_loss_terms: B <= 32 and 2 B (D + 1) + B (B + 1) floats <= 150 KiB of LDS


def loss_terms_ok(B: int, D: int) -> bool:
    return 1 <= B <= LOSS_TERMS_BMAX and 4 * (2 * B * (D + 1) + B * (B + 1)) <= 150 * 1024


def _loss_terms_desc(weights, t: dict) -> "_lib.LossTermsDesc":
    """t: name -> f32 contiguous device tensor (or None) for the fields of mh_loss_terms; shapes are read off the operands."""
    d = _lib.LossTermsDesc()
    for k, v in t.items():
        if v is not None:
            if not (v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()):
                raise MirrorHipError(f"loss_terms: {k} must be a contiguous f32 device tensor")
            setattr(d, k, v.data_ptr())
    d.has_align = int(t.get("wsi_emb") is not None)
    if d.has_align:
        d.B, d.D = t["wsi_emb"].shape
        if tuple(t["rna_emb"].shape) != (d.B, d.D) or not loss_terms_ok(d.B, d.D):
            raise MirrorHipError(f"loss_terms: alignment embeddings {tuple(t['wsi_emb'].shape)} / {tuple(t['rna_emb'].shape)} not supported")
    d.n_rna = t["rna_pred"].numel()
    if t["rna_tgt"].numel() != d.n_rna or t["rna_mask"].numel() != d.n_rna:
        raise MirrorHipError("loss_terms: rna prediction / target / mask sizes differ")
    d.n_wstyle, d.n_rstyle = t["w_mu"].numel(), t["r_mu"].numel()
    d.rows_wstyle, d.rows_rstyle = t["w_mu"].shape[0], t["r_mu"].shape[0]
    if t["w_logstd"].numel() != d.n_wstyle or t["r_logstd"].numel() != d.n_rstyle:
        raise MirrorHipError("loss_terms: mu / logstd sizes differ")
    d.Bc, d.P = t["w_score"].shape
    if tuple(t["r_score"].shape) != (d.Bc, d.P):
        raise MirrorHipError("loss_terms: prototype score shapes differ")
    for i, w in enumerate(weights):
        d.weight[i] = float(w)
    return d


def loss_terms_fwd(weights, t: dict) -> None:
    """mh_loss_terms_fwd: t carries the operands plus scratch [8] (zeroed), save [B*B + 2B], out [8], wsi_acc (or None)."""
    _lib.call("mh_loss_terms_fwd", C.byref(_loss_terms_desc(weights, t)), stream=_stream())


def loss_terms_bwd(weights, t: dict) -> None:
    """mh_loss_terms_bwd: t additionally carries g_total (and g_terms) and the d_* outputs."""
    _lib.call("mh_loss_terms_bwd", C.byref(_loss_terms_desc(weights, t)), stream=_stream())


def fanout_bwd(gfull, x, alpha: float, c, B: int, T: int, D: int) -> torch.Tensor:
    """dE[b,t] = gfull[b,t] + alpha * x[b,t-1] (t >= 1) + (t == 0 ? c[b] : 0); any of the three may be None."""
    ref = next(t for t in (gfull, x, c) if t is not None)
    _chk(*[t for t in (gfull, x, c) if t is not None])
    assert gfull is None or (gfull.is_contiguous() and gfull.dtype == torch.float32 and gfull.numel() == B * T * D)
    assert x is None or (x.is_contiguous() and x.numel() == B * (T - 1) * D)
    assert c is None or (c.is_contiguous() and c.dtype == torch.float32 and c.numel() == B * D)
    dE = torch.empty((B, T, D), device=ref.device, dtype=torch.float32)
    _lib.call("mh_fanout_bwd", _p(gfull), _p(x), float(alpha), _p(c), _p(dE), B, T, D, dt(x) if x is not None else _lib.MH_F32,
              stream=_stream())
    return dE


def weighted_sum(terms, weights) -> torch.Tensor:
    """sum_i weights[i] * terms[i] for up to 6 f32 scalars living in separate tensors -> 0-d tensor."""
    n = len(terms)
    assert 1 <= n <= 6 and len(weights) == n
    _chk(*terms)
    ts = [t.reshape(1) for t in terms]
    for t in ts:
        assert t.dtype == torch.float32
    out = torch.empty((1,), device=ts[0].device, dtype=torch.float32)
    ptrs = [_p(t) for t in ts] + [None] * (6 - n)
    ws = [float(w) for w in weights] + [0.0] * (6 - n)
    _lib.call("mh_weighted_sum", *ptrs, *ws, n, _p(out), stream=_stream())
    return out.reshape(())


def weighted_sum_bwd(g: torch.Tensor, weights) -> torch.Tensor:
    n = len(weights)
    _chk(g)
    out = torch.empty((n,), device=g.device, dtype=torch.float32)
    ws = [float(w) for w in weights] + [0.0] * (6 - n)
    _lib.call("mh_weighted_sum_bwd", _p(g), *ws, n, _p(out), stream=_stream())
    return out


def kl_fwd(mu, ls, out, coef):
    _chk(mu, ls, out)
    _lib.call("mh_kl_fwd", _p(mu), _p(ls), _p(out), mu.numel(), coef, stream=_stream())


def kl_bwd(mu, ls, g, coef):
    _chk(mu, ls, g)
    dmu, dls = torch.empty_like(mu), torch.empty_like(ls)
    _lib.call("mh_kl_bwd", _p(mu), _p(ls), _p(g), _p(dmu), _p(dls), mu.numel(), coef, stream=_stream())
    return dmu, dls


def symkl_fwd(w, r, out, coef):
    _chk(w, r, out)
    B, P = w.shape
    _lib.call("mh_symkl_fwd", _p(w), _p(r), _p(out), B, P, coef, stream=_stream())


def symkl_bwd(w, r, g, coef):
    _chk(w, r, g)
    B, P = w.shape
    dw, dr = torch.empty_like(w), torch.empty_like(r)
    _lib.call("mh_symkl_bwd", _p(w), _p(r), _p(g), _p(dw), _p(dr), B, P, coef, stream=_stream())
    return dw, dr


def rownorm_(w: torch.Tensor, eps: float = 1e-12, shadow: Optional[torch.Tensor] = None) -> None:
    """shadow: optional bf16 tensor of w's shape that receives the rounded result in the same launch."""
    _chk(w, shadow)
    assert w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 2
    if shadow is not None and (shadow.dtype != torch.bfloat16 or not shadow.is_contiguous() or tuple(shadow.shape) != tuple(w.shape)):
        raise MirrorHipError("rownorm_: the shadow is a contiguous bf16 tensor of w's shape")
    _lib.call("mh_rownorm_", _p(w), _p(shadow), w.shape[0], w.shape[1], eps, stream=_stream())


def clamp_(x: torch.Tensor, lo: float, hi: float) -> None:
    _chk(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    _lib.call("mh_clamp_", _p(x), x.numel(), lo, hi, stream=_stream())


def grad_clip(g: torch.Tensor, grad_scale: float, max_norm: float, dev_state: torch.Tensor) -> None:
    """Global-L2-norm clipping factor into dev_state[4] (and the norm into dev_state[5]); mh_adam applies it."""
    _chk(g, dev_state)
    assert g.dtype == torch.float32 and g.is_contiguous() and dev_state.numel() == 6
    scratch = torch.empty(1, device=g.device, dtype=torch.float32)
    _lib.call("mh_grad_clip", _p(g), g.numel(), grad_scale, max_norm, _p(scratch), _p(dev_state), stream=_stream())


def adam(p, g, m, v, shadow, lr, b1, b2, eps, bc1, bc2, grad_scale=1.0, dev_state: Optional[torch.Tensor] = None,
         clamp: Optional[tuple] = None, counter: Optional[torch.Tensor] = None, counter_add: int = 0, tick: bool = True,
         hole: Optional[tuple] = None) -> None:
    """dev_state: optional f32[6] device tensor {t, 1-b1^t, 1-b2^t, lr, clip, |g|}; when given the step count / bias
    corrections / lr live on the device (advanced by the launch itself), lr, bc1, bc2 are ignored and the gradient is
    also scaled by dev_state[4] (the factor grad_clip left there, else 1).  clamp = (index, lo, hi): that one parameter is clamped
    behind its update (master and shadow); counter (int64[1]) += counter_add in the same launches.  tick=False: dev_state is read,
    not advanced (tick="early": advanced, by the first launch of a two-launch step); hole = (lo, hi): those elements are left alone (the other launch of a two-launch step updates them)."""
    _chk(p, g, m, v, shadow, dev_state, counter)
    assert counter is None or (counter.dtype == torch.int64 and counter.numel() == 1)
    ci, clo, chi = (-1, 0.0, 0.0) if clamp is None else (int(clamp[0]), float(clamp[1]), float(clamp[2]))
    if ci >= p.numel():
        raise MirrorHipError("adam: clamp index outside the arena")
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    assert dev_state is None or (dev_state.dtype == torch.float32 and dev_state.numel() == 6 and dev_state.is_contiguous())
    _lib.call("mh_adam", _p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), lr, b1, b2, eps, bc1, bc2, grad_scale,
              _p(dev_state), ci, clo, chi, _p(counter), int(counter_add), 2 if tick == "early" else int(bool(tick)), *((0, 0) if hole is None else (int(hole[0]), int(hole[1]))),
              stream=_stream())
