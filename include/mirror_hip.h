/* libmirror_hip.so — C ABI of the MI355X (gfx950) kernels behind the MIRROR pre-training hot path.
 *
 * The reference (TianyiFranklinWang/MIRROR) contains no native code: every entry point below
 * replaces the ATen ops that one reference call site dispatches (file:line are relative to the
 * reference tree; [3P] = arithmetic that lives in the pip packages nystrom_attention~=0.0.14 /
 * timm~=1.0.15, called from the cited line).  The Python host (mirror_amd/) binds these with
 * ctypes and wraps them in torch.autograd.Function; INTEGRATION.md shows the binding.
 *
 * Contract: plain device pointers + sizes, explicit dtypes, caller-owned memory
 * and workspaces, a hipStream_t per call, int status return (0 ok, <0 error, text via
 * mh_last_error()); no allocation, no synchronisation, no exceptions; re-entrant across streams.
 */
#ifndef MIRROR_HIP_H
#define MIRROR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_OK 0
#define MH_EINVAL (-1)
#define MH_EUNSUP (-2)
#define MH_EHIP (-3)

#define MH_F32 0
#define MH_BF16 1

#define MH_ACT_NONE 0
#define MH_ACT_RELU 1
#define MH_ACT_GELU 2

typedef void* mh_stream; /* hipStream_t */

const char* mh_last_error(void);
int mh_version(void);
int mh_exp_build(void);      /* 1: built with -DMH_EXP (timing-experiment switches compiled in); the shipped build returns 0 */
/* number of visible HIP devices whose arch is gfx950; <=0 means the library cannot run here */
int mh_device_ok(void);

/* ---------------------------------------------------------------- MFMA GEMM (all contractions)
 * C[z] (+)= act(alpha * op(A[z]) x op(B[z]) + diag*I + bias + rcoef*R[z])   z = (b1, b2) two-level batch
 *   A(m,k) at A[m*lda + k] if a_kc else A[k*lda + m];  B(k,n) at B[n*ldb + k] if b_kc else B[k*ldb + n]
 *   mma = MH_F32  -> v_mfma_f32_32x32x2_f32 (exact fp32, parity mode; dtA=dtB=dtC=f32)
 *   mma = MH_BF16 -> v_mfma_f32_32x32x16_bf16, fp32 accumulate; f32 operands are rounded to bf16
 *                    while being staged into LDS (dtA must equal dtB)
 * Replaces: nn.Linear fwd/bwd (models/mirror.py:346, :70-74, :470-495, :594-605, :823-827),
 * [3P] NystromAttention's to_qkv / einsum similarities / attn@v / pinv matmuls / to_out
 * (models/mirror.py:312), ClipLoss logits (losses/mirror_loss.py:39-40).                      */
/* Fused epilogues of the 256 x 256-tile projection GEMM (mh_gemm_desc.epi; NULL = none).  Each replaces an elementwise pass
 * (and its HBM round trip) that the reference runs as a separate op behind an nn.Linear; the call fails with MH_EINVAL when
 * the shape is not on that kernel (bf16 operands, N % 256 == 0, K % 64 == 0, M > 256, no split-K / accumulate / R / diag), and
 * the host then runs the composed ops.  The Linear's result is rounded to bf16 (the activation dtype autocast gives it)
 * before the fused op, so fused == composed bit for bit.
 *   MH_EPI_DROPADD: C[f32] = resid + Dropout_p(A W^T + bias)         [3P] to_out = Sequential(Linear, Dropout) and the residual
 *                   add of TransLayer.forward, models/mirror.py:312-313.  The mask is mh_dropout_lite's: element i of C (flat,
 *                   C contiguous: ldc == N) uses 16-bit field i & 7 of the Philox4x32-7 block with counter (offset + i) >> 3.
 *   MH_EPI_MASKPOS: C[f32] = (t >= first && mask[b, t - first] ? token : A W^T + bias) + pos[t]      retention_embed followed by
 *                   random_masking's mask-token select and `+ retention_gene_embed`, models/mirror.py:636-643, :691-693;
 *                   flat row r of C is (b, t) = (r / rows_per_batch, r % rows_per_batch).
 *   MH_EPI_SQERR:   C[bf16] = A W^T + bias as usual, and sq[0] += sum over rows with mask[b, t] != 0 of mean_D (C - tgt)^2,
 *                   sq[1] += the number of such rows: mh_mse_masked_fwd's accumulator, filled by the projection that produces the
 *                   prediction (tgt f32 at tgt + b * tgt_bs + t * D + col; rows_per_batch % 256 == 0)
 *                   retention_head + the masked MSE of MIRRORLoss.forward, losses/mirror_loss.py:98-103. */
#define MH_EPI_NONE 0
#define MH_EPI_DROPADD 1
#define MH_EPI_MASKPOS 2
#define MH_EPI_SQERR 3
typedef struct {
    int32_t kind;
    const float* resid;           /* DROPADD: residual stream, C's shape and layout */
    float p; uint64_t seed, offset; const uint64_t* dev_base;      /* DROPADD: as mh_dropout_lite */
    const float* mask;            /* MASKPOS: [batches, rows_per_batch - first]; SQERR: [batches, rows_per_batch] (f32, != 0 = masked) */
    const float* token;           /* MASKPOS: [N] */
    const float* pos;             /* MASKPOS: [rows_per_batch, N] */
    int32_t rows_per_batch, first;
    const float* tgt; int64_t tgt_bs;      /* SQERR */
    float* sq;                    /* SQERR: 2 f32, accumulated with atomics (caller zeroes) */
} mh_gemm_epi;

typedef struct {
    const void* A; const void* B; void* C;
    const float* bias;            /* [N] f32 or NULL */
    int32_t M, N, K;
    int64_t lda, ldb, ldc;
    int32_t a_kc, b_kc;
    int32_t dtA, dtB, dtC, mma;
    int32_t batch1, batch2;
    int64_t sA1, sA2, sB1, sB2, sC1, sC2;   /* element strides of the two batch levels */
    float alpha, diag;
    int32_t act;                  /* MH_ACT_* (needs split_k == 1) */
    int32_t accumulate;           /* 0: C = r, 1: C += r */
    int32_t split_k;              /* >1: K split over workgroups, f32 atomics into C (needs accumulate=1, dtC=f32).
                                     A batch whose sC1 = sC2 = 0 with accumulate=1 also reduces into C with atomics. */
    const void* R; float rcoef;   /* optional addend: C (+)= ... + rcoef * R; R has C's dtype, ldc and batch strides
                                     (lets the pinv polynomial 15I - 7P + P.P come out of one launch) */
    float* workspace;             /* optional scratch for reductions into one C (split-K, batch broadcast): with at least
                                     mh_gemm_workspace_bytes(d) BYTES = parts * M * N * 2 (parts = splits * batch) the large-tile kernel
                                     writes plain partial tiles there and a fold pass adds them to C in f32.  The partial tiles are
                                     **bf16**: every K-slice's f32 accumulator is rounded ONCE on its way to the workspace, so an element
                                     of C carries an extra error <= parts * 2^-9 * max_slice |partial| (relative to the largest partial,
                                     not to the final sum: slices that cancel keep their rounding).  NULL / too small: f32 atomics (exact
                                     f32 partial sums, summation order not reproducible) — the caller's precision policy chooses
                                     (mirror_amd: kernels.SPLITK_PARTIALS, env MIRROR_SPLITK_PARTIALS=f32) */
    int64_t workspace_floats;     /* size of `workspace` in 4-byte units (the buffer is typed float* for alignment only) */
    void* C2;                     /* optional bf16 copy of the final C (C's ldc and batch strides), written by the same epilogue:
                                     the next product's operand without a cast launch.  192 x 384 tile kernel only (bf16
                                     operands, M % 192 == 0, N % 384 == 0, K % 64 == 0, no bias / activation / split-K) */
    int32_t r_bf16;               /* 1: R is bf16 although C is f32; 2 (v118): R is f32 although C is bf16 — a sum kept in f32 whose
                                   * next consumer reads bf16 leaves as bf16 without the f32 read-modify-write (same tile kernel only) */
    int32_t k_segments;           /* > 1: C = sum_s A_s B_s over k_segments operand pairs of K each, A_s = A + s * sA_seg, B_s = B + s * sB_seg
                                     (elements): a sum of products in one accumulator, e.g. dX = sum_k dP_k z_k^T of the pinv reverse
                                     mode without an f32 read-modify-write of C per term.  Same tile kernel only; 0 / 1: one pair */
    int64_t sA_seg, sB_seg;
    int32_t row_softmax;          /* 1: C = softmax over each row of alpha * A B, written as bf16 (N == 384 = one tile row of the same
                                     tile kernel; no R / accumulate / diag / C2): sim1 = q k_l^T of the template geometry (m = 384
                                     landmarks, models/mirror.py:312 [3P] `attn1 = sim1.softmax(dim=-1)`) without the f32 logits round trip;
                                     2: its backward, C = P o (alpha A B - rowsum(P o alpha A B)) with the probabilities P given as R (bf16) */
    const mh_gemm_epi* epi;       /* optional fused epilogue (above); NULL: none */
    int32_t a_rows_per_batch, a_row_skip;   /* > 0: A's M rows are `a_rows_per_batch`-row windows of larger batches: flat row r lives at
                                     A row r + (r / a_rows_per_batch) * a_row_skip (a_kc = 1 only).  Lets `to_out(out)[:, -n:]` and
                                     `retention_head(x)[:, 1:]` run as ONE flat problem instead of a batched one with a ragged tile per slide */
    int32_t shared_chip;          /* hint: != 0 when another stream's kernel shares the chip with this launch (the half-chip pinv chain):
                                     the persistent kernel (one workgroup per CU for the whole launch) would keep the other kernel's
                                     workgroups from being scheduled until it ends, so the one-workgroup-per-tile kernel is used */
    int32_t c_rows_per_batch, c_row_skip;   /* > 0 (bf16 C, with or without a_rows_per_batch, no `epi`): flat row r of the result is written to
                                     C row r + (r / c_rows_per_batch) * c_row_skip.  [3P] NystromAttention pads the sequence in FRONT
                                     (`F.pad(x, (0, 0, padding, 0))`, called at models/mirror.py:312): to_qkv of the pad rows is zero (no
                                     bias), so only the B x n real rows are multiplied, as one flat problem from and into the padded buffers
                                     (its data gradient likewise) */
    int32_t window_batches;       /* > 0: the row windows above cover `window_batches` batches only; flat rows past them follow the last
                                     window without a gap (row r lives at r + min(r / rows_per_batch, window_batches - 1) * row_skip).
                                     The landmark rows of a Nystrom layer sit behind the padded sequences of all slides in to_qkv's
                                     operand / result buffers and are multiplied by the same launch.  0: every flat row is window row */
} mh_gemm_desc;
int mh_gemm(const mh_gemm_desc* d, mh_stream s);
/* Tuning switch for A/B timing in one process: which main loop the 256 x 256-tile launches use (2 = persistent direct-to-LDS
 * ping-pong kernel, the default; 1 = the same, one workgroup per tile; 0 = register-staged kernel; env MH_GEMM_PP selects one
 * at start; mode < 0 only queries).  Results are identical up to the f32 summation order of split-K.  Returns the previous value. */
/* round-5 experiment (verdict item "GEMM structure"): C bf16 [M, N] = A [M, K] . B[N, K]^T + bias on the FOUR-wave form of the
 * 256 x 256 x 64 tile (128 x 128 per wave, one wave per SIMD, fragment reads of the next k-half issued in front of the current MFMAs);
 * whole tiles only (M, N % 256 == 0, K % 64 == 0), K-contiguous operands.  Measured against mh_gemm in tools/exp/time_w4.py. */
int mh_gemm_w4(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
               mh_stream s);
int mh_gemm_select_pp(int mode);
/* The kernel instance the calling thread's last mh_gemm call launched (e.g. "gemm_pq_kernel<float,false,false,part>",
 * "gemm_kernel<1,bf16,bf16,float,true,false,2,2,1>"): lets a profiler name launches without restating the dispatch rules. */
const char* mh_gemm_variant_name(void);
/* Bytes of `workspace` with which this call reduces through plain **bf16** partial tiles (parts * M * N * 2) + an f32 fold pass
 * instead of f32 atomics (0: the call has no use for one — no split-K / batch broadcast, fewer than 8 parts, or the shape is not
 * on the large-tile kernel). */
int64_t mh_gemm_workspace_bytes(const mh_gemm_desc* d);

/* ---------------------------------------------------------------- skinny-M linears (RNA encoder / style heads: every
 * tensor is [B, D], models/mirror.py:77-102, :217-224, :845-857): weight-streaming kernels, bf16 operands.
 * y[M<=32, N] = act(x[M,K] W[N,K]^T + bias + addend); K % 32 == 0 with 16-byte aligned rows streams 16-byte fragments, any other K / row
 * stride (the template's 10234 genes and 1975-wide MLP, configs/pretrain/mirror.template.yaml:27-46) the same kernel element-wise
 * (v116+); the data gradient is the same call on the W^T shadow, and
 * `addend` (nullable f32 [M, N], row stride ldadd) is how the data gradients of two linears that read the same x are summed
 * without a launch of their own (style_mu / style_logstd, models/mirror.py:845-857). */
int mh_skinny_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, const float* addend, int64_t ldadd,
                  void* y, int64_t ldy, int M, int N, int K, int act, int dt_x, int dt_y, mh_stream s);      /* x bf16, or f32 rounded to bf16 on load */
/* dW[N,K] (+)= dy[M,N]^T x[M,K]  (f32, plain read-modify-write: one block owns each output tile) */
int mh_skinny_wgrad(const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, int64_t lddw, float* db, int M, int N,
                    int K, int accumulate, int dt_dy, int dt_x, mh_stream s);   /* db (optional, [N] f32): db += column sums of dy (the bias gradient) */
/* Up to MH_SKINNY_MANY_MAX of those weight gradients in ONE launch (always accumulating: dW += dy^T x, db += column sums of dy):
 * the backward of the [B, D]-row linears queues its (dy, x) pairs and one grid covers the tiles of all of them. */
#define MH_SKINNY_MANY_MAX 32
typedef struct {
    const void* dy; int64_t lddy;     /* [M, N] bf16 (or f32, rounded to bf16 on load: dt_dy) */
    const void* x; int64_t ldx;       /* [M, K] bf16 (or f32: dt_x) */
    float* dw; int64_t lddw;          /* [N, K] f32, 16-byte aligned */
    float* db;                        /* [N] f32 or NULL */
    int32_t M, N, K;
    int32_t dt_dy, dt_x;              /* MH_BF16 / MH_F32 */
} mh_skinny_wgrad_item;
int mh_skinny_wgrad_many(const mh_skinny_wgrad_item* items, int n, mh_stream s);
/* bf16 [R,C] -> [C,R]; the batched form walks table[i] = {src_off, dst_off, R, C} (int64 element offsets).
 * vec_ok != 0: every entry has R % 8 == 0, C % 8 == 0 and offsets that are multiples of 8 (16-byte accesses) */
int mh_transpose_bf16(const void* in, void* out, int R, int C, mh_stream s);
int mh_transpose_bf16_many(const void* src, void* dst, const int64_t* table, int n, int max_r, int max_c, int vec_ok,
                           mh_stream s);

/* ---------------------------------------------------------------- LayerNorm (models/mirror.py:298, :350, :604, :210)
 * rows are addressed as (b, i): x + b*x_bs + i*D, y + b*y_bs + i*D, i < rows_per_batch (lets the
 * output land inside the front-zero-padded Nystrom buffer, or skip the square-pad rows :679). */
int mh_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                     int batches, int rows_per_batch, int D, int64_t x_bs, int64_t y_bs, float eps,
                     int dt_x, int dt_y, mh_stream s);
/* f32 output y plus a bf16 copy y_bf16 (same row addressing) in one pass: the encoder's final norm (models/mirror.py:679) feeds
 * f32 consumers (retention target :699, cls row :684) and the bf16 operand of retention_embed (:690).  D % 4 == 0, D <= 2048. */
int mh_layernorm_fwd_dual(const float* x, const float* gamma, const float* beta, float* y, void* y_bf16, float* mean, float* rstd,
                          int batches, int rows_per_batch, int D, int64_t x_bs, int64_t y_bs, float eps, mh_stream s);
/* the same with an e4m3 copy of the output beside the bf16 one (x f32, y bf16; q8 has y's row addressing, one byte per element):
 * delayed per-tensor scaling as in mh_quant_fp8_delayed (ring: 3 uint32 of this call site, tick: device step counter), scale[0] =
 * the dequantisation factor.  Config 5: the fp8 forward of [3P] to_qkv reads q8, its weight gradient the bf16 copy. */
int mh_layernorm_fwd_q8(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                        int batches, int rows_per_batch, int D, int64_t x_bs, int64_t y_bs, float eps, void* q8, unsigned* ring,
                        const float* tick, float margin, float* scale, mh_stream s);
/* dx = d/dx, dgamma/dbeta accumulated (+=, f32; caller zeroes them). dy uses y's addressing.
 * workspace (optional, f32, ws_floats >= 2*D): per-block dgamma/dbeta partials are written there and folded by a second
 * small launch instead of thousands of same-address atomics; size it 2*D*min(rows/16, 1024) floats for full speed. */
int64_t mh_layernorm_bwd_workspace_bytes(int64_t rows, int D);      /* full-speed size of `workspace` below */
int mh_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                     void* dx, float* dgamma, float* dbeta,
                     int batches, int rows_per_batch, int D, int64_t x_bs, int64_t y_bs,
                     int dt_x, int dt_dy, int dt_dx, int accumulate_dx, float* workspace, int64_t ws_floats, mh_stream s);

/* LayerNorm of a Nystrom layer that also leaves the landmark means of its output (models/mirror.py:298 + [3P] NystromAttention's
 * front padding and `q_landmarks = reduce(q, '... (n l) d -> ... n d', 'sum') / l`): x f32 [batches, >= rows, D] (x_bs elements per
 * batch) -> y bf16 [batches, pad + rows, D] (pad zero rows first, written here) and xpm f32 [batches, (pad + rows) / l, D] =
 * the mean of each group of l consecutive rows of y.  to_qkv is linear and bias-free, so the landmarks are to_qkv(xpm)[:, :2D].
 * mh_layernorm_bwd_lm: mh_layernorm_bwd whose dy rows also receive gadd[b, (i + pad) / l] / l (gadd, in dy's dtype, = d loss / d xpm).
 * xpm (f32) may be NULL when only the bf16 means are wanted (they are the landmark rows behind the sequence in to_qkv's operand). */
int mh_layernorm_fwd_lm(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, void* xpm,
                        void* xpm_bf16, int batches, int rows, int D, int64_t x_bs, int pad, int l, float eps, const float* row_mask,
                        const float* lm_scale, mh_stream s);
                        /* row_mask (nullable, f32 [batches, pad + rows]; BASELINE config 4): the key-padding mask, front-padded like the
                           sequence — rows with a zero entry leave as zero rows ([3P] `q, k, v = t * mask[..., None]` through the bias-free
                           to_qkv) and stay out of their group's sum: the landmark rows are masked sums / l (the caller scales them by
                           l / valid count — or passes those factors as lm_scale, f32 [batches, (pad + rows) / l], nullable: the landmark rows then leave as
                           the masked MEANS); mh_layernorm_bwd_lm takes the same mask and scale */
                        /* xpm_bf16 (optional): the bf16 rounding of xpm, the B operand of the landmark projection's weight gradient */
/* mh_layernorm_bwd for a LayerNorm output that fans out (the WSI encoder's final norm: decoder input, retention target = rows 1..,
 * cls row; models/mirror.py:684-700, :833): f32 x / dx, dy f32 or bf16 (dt_dy; round 5: the decoder's data gradient arrives in
 * bf16, as under autocast); dy row i >= 1 of batch b also receives fan_alpha * fan_bf16[b, i - 1]
 * (fan_bf16 [batches, rows_per_batch - 1, D]: the masked MSE hands its target gradient over as -dpred) and row 0 fan_cls[b]
 * (f32 [batches, D], may be NULL) — the three-way sum mh_fanout_bwd writes is formed while this launch reads its operands.
 * Needs the workspace form (D % 4 == 0, 16-byte aligned buffers, >= 64 rows, mh_layernorm_bwd_workspace_bytes). */
int mh_layernorm_bwd_fan(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                         void* dx, float* dgamma, float* dbeta, int batches, int rows_per_batch, int D, int64_t x_bs, int64_t y_bs,
                         int dt_dy, int accumulate_dx, float* workspace, int64_t ws_floats, const void* fan_bf16, float fan_alpha,
                         const float* fan_cls, mh_stream s);
/* mh_layernorm_bwd (optionally with mh_layernorm_bwd_fan's two extra gradients: fan_bf16 may be NULL) for a LayerNorm whose input x is the
 * output of `resid + Dropout_p(Linear(core))` ([3P] to_out = Sequential(Linear, Dropout) and TransLayer's residual add,
 * models/mirror.py:312): this launch's dx is that Dropout's upstream gradient, so drop_out [batches * rows, D] bf16 receives
 * mask * dx / (1 - p) on the lite Philox stream (the masks mh_gemm_epi DROPADD drew for (seed, offset + *dev_base, element)) — the operand of
 * to_out's two gradient products — and drop_db [D] += its column sums (to_out's bias gradient): mh_dropout_lite_colsum's pass over dx is
 * not launched.  f32 x / dx; dy f32 or bf16; the workspace form with >= 3 D floats per block; D % 8 == 0, D <= 1024. */
int mh_layernorm_bwd_drop(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                          void* dx, float* dgamma, float* dbeta, int batches, int rows_per_batch, int D, int64_t x_bs, int64_t y_bs,
                          int dt_dy, int accumulate_dx, float* workspace, int64_t ws_floats, const void* fan_bf16, float fan_alpha,
                          const float* fan_cls, void* drop_out, float drop_p, uint64_t drop_seed, uint64_t drop_offset,
                          const uint64_t* drop_base, float* drop_db, int drop_rows_per_batch, mh_stream s);
                          /* drop_rows_per_batch >= rows_per_batch: rows per batch of the Dropout's tensor [batches, ., D] (drop_out has its
                             shape); a norm over the first rows of a square-padded sequence leaves the rows behind them to the caller (zeros) */
int mh_layernorm_bwd_lm(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                        void* dx, float* dgamma, float* dbeta, int batches, int rows_per_batch, int D, int64_t x_bs, int64_t y_bs,
                        int dt_x, int dt_dy, int dt_dx, int accumulate_dx, float* workspace, int64_t ws_floats,
                        const void* gadd, int pad, int l, void* relu_out, int relu_first, int relu_rows, float* relu_db,
                        const float* row_mask, const float* lm_scale, mh_stream s);
                        /* relu_out (nullable, bf16 [batches, relu_rows, D]; f32 x): x rows [relu_first, relu_first + relu_rows) of every batch
                           are the output of a ReLU (`_fc1 = Linear + ReLU`, models/mirror.py:346, :652-654, feeds layer 1's norm): their
                           total gradient is written HERE as bf16 (x > 0 ? dx : 0), the operand of _fc1's weight gradient, and NOT as f32 dx
                           (no separate ReLU-backward pass); the other rows (cls) keep their f32 dx.
                           relu_db (nullable, f32 [D], needs relu_out and a workspace of >= 3 D floats): += the column sums of relu_out as
                           stored, i.e. _fc1's bias gradient, as a third partial row beside dgamma / dbeta (no mh_colsum pass over relu_out) */

/* ---------------------------------------------------------------- fp8 forward projections (BASELINE config 5)
 * mh_quant_fp8: q[i] = e4m3(x[i] * 448 / max|x|) for a whole tensor (x f32 / bf16, n % 4 == 0), scale[0] = max|x| / 448
 *               (amax_scratch: one uint32 of device scratch).
 * mh_gemm_fp8 : C[z] = act(scale_a * scale_b * A[z] B^T + bias), v_mfma_f32_32x32x16_fp8_fp8, f32 accumulate.  A [batch][M, K] and
 *               B [N, K] are e4m3 bytes with K contiguous (lda, ldb, a_bs in bytes); C f32 / bf16 (ldc, c_bs in elements);
 *               K % 64 == 0, N % 128 == 0, M ragged.  Replaces the forward of nn.Linear at models/mirror.py:346 (`_fc1`), [3P]
 *               to_qkv / to_out and the retention embed / head (:595-607) under the `fp8` precision policy. */
int mh_quant_fp8(const void* x, int64_t n, void* q, float* scale, unsigned* amax_scratch, int dt, mh_stream s);
/* Delayed scaling, ONE pass: the scale is `margin` x this tensor's max |x| of the PREVIOUS step (ring: 3 uint32 of device state
 * per call site, zero-initialised; tick: device f32 holding the step counter, e.g. TrainEngine's Adam state), this step's
 * max is gathered for the next step.  Values past the e4m3 range saturate.  The first two steps of a call site (ring still
 * empty) must go through mh_quant_fp8. */
int mh_quant_fp8_delayed(const void* x, int64_t n, void* q, float* scale, unsigned* ring, const float* tick, float margin,
                         int dt, mh_stream s);
int mh_gemm_fp8(const void* A, int64_t lda, int64_t a_bs, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t c_bs, int batch,
                const float* scale_a, const float* scale_b, const float* bias, int act, int M, int N, int K, int dt_c, mh_stream s);

/* ---------------------------------------------------------------- row softmax ([3P] sim.softmax(-1); Attention :95)
 * x/y: rows x cols, row stride ld (elements). In place allowed when dtypes match. */
int mh_softmax_fwd(const void* x, void* y, int64_t rows, int cols, int64_t ldx, int64_t ldy,
                   int dt_x, int dt_y, mh_stream s);
/* dx = y * (dy - sum(dy*y)) ; in place on dy allowed */
int mh_softmax_bwd(const void* y, const void* dy, void* dx, int64_t rows, int cols, int64_t ldy, int64_t lddy,
                   int64_t lddx, int dt_y, int dt_dy, int dt_dx, mh_stream s);

/* Key-padding-mask variants ([3P] NystromAttention.forward(x, mask=...): `sim.masked_fill_(~(rowmask & colmask), -finfo.max)`
 * before each of the three softmaxes; BASELINE config 4).  x, y: contiguous [batches, h, R, cols]; rowmask [batches, R] and
 * colmask [batches, cols] hold 0 / 1 floats.  A fully masked row comes out uniform, as in the package.  bwd: dx = 0 at
 * filled entries, y (dy - sum y dy) elsewhere; in place on dy allowed. */
int mh_softmax_masked_fwd(const void* x, void* y, const float* rowmask, const float* colmask, int64_t batches, int h, int R,
                          int cols, int dt_x, int dt_y, mh_stream s);
int mh_softmax_masked_bwd(const void* y, const void* dy, void* dx, const float* rowmask, const float* colmask, int64_t batches,
                          int h, int R, int cols, int dt_y, int dt_d, mh_stream s);
/* y[r, :] = x[r, :] * scale[r] (masked rows zeroed before the bias-free to_qkv; masked-mean landmarks = sum / (count + 1e-8)) */
int mh_row_scale(const void* x, const float* scale, void* y, int64_t rows, int D, int dt, mh_stream s);
/* Key-padding plan of one Nystrom layer in one launch (BASELINE config 4; the `mask` argument of [3P] NystromAttention.forward).
 * mask: [B, n_src] bool bytes (non-zero = real patch).  The layer's sequence is [pad zeros | lead ones (cls) | mask |
 * mask[:, :wrap] (the square-pad rows of models/mirror.py:357-360)], pad + lead + n_src + wrap = m * l.  Out, all f32:
 * mrow [B, m*l] row mask, mlm [B, m] = (valid rows of the group > 0), lscale [B, m] = l * (1 / (count + 1e-8)). */
int mh_keymask_plan(const unsigned char* mask, float* mrow, float* mlm, float* lscale, int64_t B, int64_t n_src, int lead, int wrap,
                    int pad, int l, mh_stream s);

/* ---------------------------------------------------------------- Nystrom pieces ([3P], called at models/mirror.py:312)
 * qkv: [B, n_p, 3D] (q | k | v column blocks, heads are dh-wide column slices).
 * landmarks: lm[b, j, c] = mean_{t<l} qkv[b, j*l+t, c], c < 2D  -> lm [B, m, 2D]            */
int mh_landmark_fwd(const void* qkv, void* lm, int B, int n_p, int D, int l, int dt, mh_stream s);
/* dqkv[b, r, c] += dlm[b, r/l, c] / l for c < 2D */
int mh_landmark_bwd(const void* dlm, void* dqkv, int B, int n_p, int D, int l, int dt, mh_stream s);
/* out[b,t,h*dh+d] += sum_j w[h][j] * v[b, t+j-K/2, h*dh+d]; v = qkv[..., 2D:3D] (ldv = 3D), out ld = ldo.
 * transpose=1 applies the adjoint (data gradient).                                           */
int mh_resconv_fwd(const void* v, int64_t ldv, int64_t v_bs, const float* w, void* out, int64_t ldo, int64_t o_bs,
                   int B, int n_p, int heads, int dh, int taps, int transpose, int accumulate,
                   int dt_v, int dt_o, mh_stream s);
/* dw[h][j] += sum_{b,t,d} dout[b,t,h,d] * v[b,t+j-K/2,h,d]  (f32 atomics) */
int mh_resconv_wgrad(const void* v, int64_t ldv, int64_t v_bs, const void* dout, int64_t ldo, int64_t o_bs,
                     float* dw, int B, int n_p, int heads, int dh, int taps, int dt_v, int dt_o, mh_stream s);
/* Both gradients of res_conv in ONE pass over dout ([3P] `out += self.res_conv(v)`, called at models/mirror.py:312):
 *   dv[b,t,h*dh+d] += sum_j w[h][j] * dout[b, t-j+K/2, h*dh+d]      (= mh_resconv_fwd(dout -> dv, transpose=1, accumulate=1))
 *   dw[h][j]       += sum_{b,t,d} dout[b,t,h,d] * v[b,t+j-K/2,h,d]   (= mh_resconv_wgrad)
 * bf16 operands with dh = 64, 33 taps and `workspace` of mh_resconv_bwd_workspace_bytes(..) bytes: one launch over 128-row tiles
 * (the dout tile of a head sits in LDS for the adjoint conv; the tap gradient is 8 more MFMAs per 32 rows against v rows read
 * beside it, its per-tile sums go through `workspace`) + a fold launch; anything else runs the two passes above. */
int64_t mh_resconv_bwd_workspace_bytes(int B, int n_p, int heads, int dh, int taps);
int mh_resconv_bwd(const void* dout, int64_t ldo, int64_t o_bs, const void* v, int64_t ldv, int64_t v_bs, const float* w, void* dv,
                   int64_t lddv, int64_t dv_bs, float* dw, float* workspace, int64_t ws_floats, int B, int n_p, int heads, int dh,
                   int taps, int dt_v, int dt_o, mh_stream s);
/* moore_penrose_iter_pinv initial scaling: stats[0]=max_i sum_j|x|, stats[1]=max_j sum_i|x| over the
 * WHOLE [BH,m,m] tensor (couples the batch), with arg positions packed in stats64. */
int mh_pinv_absmax(const float* x, uint64_t* stats64, int BH, int m, mh_stream s);
/* z0[bh,i,j] = x[bh,j,i] / (c*r) */
int mh_pinv_z0(const float* x, const uint64_t* stats64, float* z0, int BH, int m, mh_stream s);
/* dx += dz0^T/(c r) + sub-gradients through the two max(); z0 may be NULL (m % 64 == 0): z0 = x^T / (c r) is then formed on the fly.
 * scratch1 = one device float the kernels reduce into; scratch_zeroed != 0: the caller hands it over holding 0 (no memset here) */
int mh_pinv_z0_bwd(const float* x, const float* z0, const float* dz0, const uint64_t* stats64, float* dx,
                   float* scratch1, int scratch_zeroed, int BH, int m, mh_stream s);
/* attn2's backward tail in one pass (m = 256, x = p = the softmax output of [3P] sim2, no key-padding mask): on entry dx holds the
 * chain's d loss / d x; on exit dx = d loss / d sim2-logits = softmax_bwd(p, dx + dz0^T / (c r) + max sub-gradients).  Equals
 * mh_pinv_z0_bwd(x = p, z0 = NULL) followed by mh_softmax_bwd up to the rounding of the row sums (the row maximum's constant cancels
 * in a softmax backward; the column maximum's is applied as a rank-one correction of the one matrix that holds it). */
int mh_pinv_s2_bwd(const float* p, const float* dz0, const uint64_t* stats64, float* dx, float* scratch1, int scratch_zeroed, int BH,
                   int m, const float* mlm, int heads, mh_stream s);
                   /* mlm (nullable, f32 [BH / heads, m]; BASELINE config 4): p came out of masked_fill + softmax (mh_nys_sim2(mlm) /
                      mh_softmax_masked_fwd); entries of an invalid row or column landmark receive no gradient (dx = 0 there) */
/* The whole iteration as ONE launch per pass (bf16 policy, m = 256; other sizes return MH_EINVAL and the caller
 * composes mh_gemm): one 256-thread workgroup per (b,h) walks the chain of m x m products; a wave owns 64 columns of
 * every product, its B operand never leaves the register file, A is one LDS image (pinv_panel.hip).
 * Chain-private matrices are "panel native" (PN): PN(M)[jblk][T][lane][e] = M[16T + 4hl + (e&3) + 8(e>>2)][32 jblk + c],
 * c = lane & 31, hl = lane >> 5 — the 16 bytes one lane feeds to one MFMA k-step, lanes consecutive (coalesced).
 *   prep : x [BH,m,m] f32 (attn2) + stats64 -> z0 = x^T/(c r) f32 row-major (as mh_pinv_z0), xp = PN(x),
 *          z0p = PN(z0) (the caller places it at saved[0][0])
 *   pack : dz [BH,m,m] f32 row-major (d z_iters) -> up = PN(dz^T), the backward's input
 *   fwd  : XP = xp; saved [iters][4][BH][m][m] bf16 = PN{z_k, P_k, T2_k, T3_k}; zfT[j][i] = z_iters[i][j] (column-major,
 *          i.e. the row-major transpose of the pseudo-inverse)
 *   bwd  : dzf = up; work like saved (scratch); dX (f32, row-major) = sum_k dP_k z_k^T, dz0 (f32, row-major) = d z_0 */
int mh_pinv_chain_prep(const float* x, const uint64_t* stats64, float* z0, void* xp, void* z0p, int BH, int m, mh_stream s);
int mh_pinv_chain_pack(const float* dz, void* up, int BH, int m, mh_stream s);
int mh_pinv_chain_fwd(const void* XP, void* saved, void* zfT, int BH, int m, int iters, const float* z0f, const uint64_t* stats64,
                      int z0_rowmajor, mh_stream s);     /* z0f != NULL: z_0 = z0f / (c r) (mh_nys_sim2's unscaled panel-native attn2^T, stats64 = its maxima),
                                           rounded to bf16 and written to saved[0] here; NULL: saved[0] is pre-filled (mh_pinv_chain_prep) */
/* attn2 = softmax(scale q_l k_l^T) of [3P] NystromAttention (m = 256 landmarks, dh = 64) and what moore_penrose_iter_pinv's start
 * needs from it, one launch: lm bf16 [B, m, 2 D] (q | k landmarks) -> a2 f32 [B h, m, m] row-major, xp = panel-native bf16 a2 (the
 * chain's X), z0f = panel-native f32 a2^T (unscaled z_0), stats64[0 / 1] = packed (value << 32 | flat index) maxima of the row /
 * column abs sums as mh_pinv_absmax leaves them (caller zeroes stats64).  Replaces mh_gemm + mh_softmax_fwd + mh_pinv_absmax +
 * mh_pinv_chain_prep on the fused path. */
int mh_nys_sim2(const void* lm, float* a2, void* xp, float* z0f, uint64_t* stats64, int B, int m, int D, int heads, float scale,
                int64_t lm_ld, const float* mlm, mh_stream s);
                /* lm_ld: row stride of lm in elements; 0 = contiguous [B, m, 2D] (see mh_nys_attn1_fwd).  mlm (nullable, f32 [B, m]; BASELINE
                   config 4): valid-landmark flags of the key-padding mask — entries of an invalid row or column are filled with -FLT_MAX in
                   front of the softmax ([3P] sim2.masked_fill_), z0f must then be NULL (mh_pinv_chain_fwd(z0_rowmajor) forms z_0) */
/* The two small products that open NystromAttention's backward around the chain, one launch (m = 256, dh = 64; dw2, av f32
 * [BH, m, dh], zfT bf16 [BH, m, m] = the chain's column-major output): up = PN((dw2 av^T)^T) bf16, the input of mh_pinv_chain_bwd
 * (what mh_gemm + mh_pinv_chain_pack produce), dav = Z^T dw2 bf16 [BH, m, dh].  delta3 (f32 [BH, m], may be NULL) receives
 * sum_d dav[., d] av[., d] of the rounded dav: hand it to mh_nys_attn3_bwd with av = NULL and that call skips its first launch. */
int mh_nys_dz_dav(const float* dw2, const float* av, const void* zfT, void* up, void* dav, float* delta3, int BH, int m, int dh,
                  mh_stream s);
int mh_pinv_chain_bwd(const void* XP, const void* saved, const void* dzf, void* work, float* dX, float* dz0, int BH, int m,
                      int iters, mh_stream s);
/* Bytes of the chain's caller-allocated buffers: which = 0: `saved` (forward output, backward input: [iters, 4, BH, m, m]
 * bf16 = the iterates z_k, P_k, T2_k, T3_k); which = 1: `work` (backward scratch: [iters + 4, BH, m, m] bf16 = W_k of every
 * iteration for dX, and four slots of per-iteration temporaries). */
int64_t mh_pinv_chain_workspace_bytes(int BH, int m, int iters, int which);
/* Fused attention sides of the Nystrom core (bf16 policy, dh = 64, m = 256 landmarks; anything else returns
 * MH_EINVAL and the caller composes mh_gemm + mh_softmax).  The [n_p x m] / [m x n_p] similarity matrices stay in MFMA
 * accumulators; only row statistics reach HBM.  Replaces, in [3P] NystromAttention.forward (called at
 * models/mirror.py:312): sim1/sim3 einsum + softmax + the two products with them, and their autograd.
 *   qkv [B,n_p,3D] bf16 (D = 64 h, heads are 64-wide column slices), lm [B,m,2D] bf16 = q_l | k_l,
 *   w2 [B,h,m,64] bf16, out/dout [B,n_p,D] bf16, av [B,h,m,64] f32, dav [B,h,m,64] bf16,
 *   lse1/delta1 [B,h,n_p] f32, lse3 [B,h,m] f32, dqkv like qkv, dw2 [B,h,m,64] f32, dlm [B,m,2D] f32.
 * attn1_fwd: out[:, :, head] (+)= softmax_m(scale q k_l^T) w2 (accumulate = 1 adds to what is there, e.g. the res_conv
 *            term computed while the pinv chain was running), lse1 = row logsumexp.
 * attn3_fwd: av = softmax_n(scale q_l k^T) v, lse3.
 * attn1_bwd: two parts, selected by `which` (1, 2 or 3 = both, in this order):
 *   1: ADDS (f32 atomics) dw2 = P1^T dO and dk_l = dS1^T q into dw2 and the k_l half of dlm, and WRITES delta1[b, h, n] =
 *      sum_l P1 dP1 = sum_d dO[n, d] o1[n, d] — o1 = attn1's own output rows as mh_nys_attn1_fwd(o1 = ..) saved them (the
 *      flash-attention identity).  dw2 is all the pinv chain's backward waits for, so this part runs first.
 *   2: writes the q block of dqkv = dS1 k_l from delta1 (a single pass: 3 products); nothing the chain needs — it can run beside it.
 * attn3_bwd: writes the k and v blocks of dqkv and delta3 [B,h,m] f32 (scratch); ADDS into the q_l half of dlm. */
/* mrow [B, n_p] / mlm [B, m] (f32 0 / 1, both or neither): the package's key-padding mask — valid sequence rows and landmark
 * groups that contain a valid row.  A logit whose row or landmark is invalid is masked_fill'ed before the softmax (a fully
 * masked row comes out uniform, as in the package) and gets no gradient.  NULL: no mask. */
/* lm_ld (all mh_nys_attn*): row stride of `lm` in elements, 0 = 2 D (a contiguous [B, m, 2D]).  3 D when the landmarks are rows of
 * to_qkv's output buffer behind the sequence (landmarks = to_qkv(group means): [3P] landmarks are means over l consecutive
 * positions of q and k, and to_qkv is linear and bias-free); batch b's landmarks start at lm + b * m * lm_ld. */
/* o1 (nullable, bf16 [B, n_p, D]): attn1's own rows softmax(scale q k_l^T) w2, WITHOUT whatever `out` held under accumulate = 1
 * (res_conv(v)): what mh_nys_attn1_bwd part 1 takes delta1 from. */
int mh_nys_attn1_fwd(const void* qkv, const void* lm, const void* w2, void* out, float* lse1, const float* mrow, const float* mlm,
                     int B, int h, int n_p, int m, int dh, float scale, int accumulate, int64_t lm_ld, void* o1, mh_stream s);
/* unmasked mh_nys_attn1_fwd that also writes the e4m3 copy q8 [B, n_p, D] bytes of `out` (delayed scaling: ring / tick / margin as
 * in mh_quant_fp8_delayed, q8_scale[0] = dequantisation factor): the fp8 forward of [3P] to_out needs no quantisation pass */
int mh_nys_attn1_fwd_q8(const void* qkv, const void* lm, const void* w2, void* out, float* lse1, int B, int h, int n_p, int m, int dh,
                        float scale, int accumulate, void* q8, unsigned* ring, const float* tick, float margin, float* q8_scale, void* o1,
                        mh_stream s);
/* attn3_fwd cuts the sequence into ranges (one workgroup each) when B*h alone would not fill the chip; the partial results
 * live in `workspace` (mh_nys_attn3_ws_floats(B, h, n_p) floats; NULL / too small: one workgroup per (b, h)). */
int64_t mh_nys_attn3_ws_floats(int B, int h, int n_p);
int64_t mh_nys_attn3_workspace_bytes(int B, int h, int n_p);         /* the same in bytes */
/* rc_w / rc_out (both or neither): [3P] `out += self.res_conv(v)` computed by the same launch — rc_w = the 33-tap filters [h][33]
 * (Conv2d(h, h, (33, 1), groups=h, bias=False), models/mirror.py:299-309), rc_out [B, n_p, D] bf16 receives res_conv(v) (every row
 * written; mh_nys_attn1_fwd(accumulate=1) then adds its product): the conv reads the v tile this kernel stages anyway. */
int mh_nys_attn3_fwd(const void* qkv, const void* lm, float* av, float* lse3, float* workspace, int64_t ws_floats,
                     const float* mrow, const float* mlm, int B, int h, int n_p, int m, int dh, float scale, int64_t lm_ld,
                     const float* rc_w, void* rc_out, mh_stream s);
int mh_nys_attn1_bwd(const void* qkv, const void* lm, const void* w2, const void* dout, const float* lse1, const void* o1, float* delta1,
                     void* dqkv, float* dw2, float* dlm, const float* mrow, const float* mlm, int B, int h, int n_p, int m, int dh,
                     float scale, int64_t lm_ld, int which, mh_stream s);
/* av == NULL: delta3 already holds sum_d dav av (mh_nys_dz_dav wrote it); otherwise it is scratch this call fills first */
/* one_pass != 0 (round 5): dk, dv and dq_l from ONE kernel (every logit / dP / exponential computed once, k and v read once: the
 * landmark-owning waves hand P and dS to the key-owning role through two LDS images); 0: the dk / dv kernel + the dq_l kernel */
int mh_nys_attn3_bwd(const void* qkv, const void* lm, const float* av, const void* dav, const float* lse3, float* delta3,
                     void* dqkv, float* dlm, const float* mrow, const float* mlm, int B, int h, int n_p, int m, int dh,
                     float scale, int64_t lm_ld, int one_pass, mh_stream s);
/* out[r, 0:cols] = bf16(a[r] + b[r]) (f32 [rows, cols], b may be NULL) at row stride out_ld, out[r, cols:cols + zero_cols] = 0: the
 * landmark gradient (q_l | k_l halves from the attention kernels + sim2's products) written as rows [dq_l | dk_l | 0] of to_qkv's
 * output gradient, where its data / weight gradient products pick it up together with the sequence rows. */
int mh_lm_merge(const float* a, const float* b, void* out, int64_t rows, int cols, int64_t out_ld, int zero_cols, mh_stream s);
/* T = d*I - P  (batched [BH,m,m] f32) */
int mh_eye_minus(const float* P, float* T, float d, int BH, int m, mh_stream s);

/* ---------------------------------------------------------------- TransMIL glue (models/mirror.py:657-665, :317-331)
 * seq [B, n, D]: row 0 <- cls, rows 1+N.. <- copies of rows 1..add (square pad). */
int mh_seq_finish(void* seq, const float* cls, int B, int N, int add, int D, int dt, mh_stream s);
/* dseq[b,1+i] += dseq[b,1+N+i] (i<add); dcls[c] += sum_b dseq[b,0,c] */
int mh_seq_finish_bwd(void* dseq, float* dcls, int B, int N, int add, int D, int dt, mh_stream s);
/* merged[49][D] = w7 + embed(w5) + embed(w3) + centre 1 ; bsum[D] = b7+b5+b3 */
int mh_ppeg_merge(const float* w7, const float* w5, const float* w3, const float* b7, const float* b5,
                  const float* b3, float* merged, float* bsum, int D, mh_stream s);
/* depthwise 7x7 on tokens 1..S*S of seq [B, 1+S*S, D]; cls row copied. flip=1 -> adjoint (no bias). */
int mh_ppeg_fwd(const void* x, void* y, const float* merged, const float* bsum, int B, int S, int D,
                int flip, int dt_x, int dt_y, mh_stream s);
/* dmerged[tap][c] += sum dout*x(shifted); dbsum[c] += sum dout  (f32 atomics) */
int mh_ppeg_wgrad(const void* x, const void* dout, float* dmerged, float* dbsum, int B, int S, int D,
                  int dt_x, int dt_o, mh_stream s);
/* dw7 [D,1,7,7] += dmerged^T, dw5 [D,1,5,5] += its centre 5x5, dw3 [D,1,3,3] += its centre 3x3, db7 / db5 / db3 [D] += dbsum:
 * the gradients of the three depthwise kernels PPEG sums (models/mirror.py:324-331) from the merged kernel's gradient
 * (dmerged is tap-major [49, D] as mh_ppeg_wgrad writes it).  Accumulates: the caller owns the zeroing. */
int mh_ppeg_grad_scatter(const float* dmerged, const float* dbsum, float* dw7, float* dw5, float* dw3, float* db7,
                         float* db5, float* db3, int D, mh_stream s);

/* ---------------------------------------------------------------- masking (models/mirror.py:624-649, :510-533)
 * mask[b,i] = 1 if rank(noise[b,i]) >= len_keep (rank by ascending noise, ties by index) */
int mh_rank_mask(const float* noise, float* mask, int B, int N, int len_keep, mh_stream s);
/* y [B, T, D] (dt_y): rows t>=first take `token` where mask[b,t-first] != 0, else x (dt_x); then + pos[t]  (pos [T,D]) */
int mh_mask_apply_fwd(const void* x, void* y, const float* mask, const float* token, const float* pos, int B, int T, int D,
                      int first, int token_scalar, int dt_x, int dt_y, mh_stream s);   /* y may be x (in place, same dtype) */
/* dx (dt_dx) = dy*(1-mask) (dx may be dy when the dtypes agree); dtoken += sum mask*dy ; dpos[t] += sum_b dy.
 * dbias [D] f32 (may be NULL; only where mh_mask_apply_bwd_dbias_ok): += the column sums of dx, i.e. the bias gradient of the Linear
 * whose output the forward masked (retention_embed, models/mirror.py:636-643) — the pass already holds both sums it is the
 * difference of, so mh_colsum over dx is not launched. */
int mh_mask_apply_bwd_dbias_ok(const void* dy, const void* dx, const float* dpos, int D, int dt_dy, int dt_dx);
int mh_mask_apply_bwd(const void* dy, void* dx, const float* mask, float* dtoken, float* dpos, int B, int T, int D,
                      int first, int token_scalar, int dt_dy, int dt_dx, float* dbias, mh_stream s);

/* ---------------------------------------------------------------- RNA encoder pieces (models/mirror.py:77-102)
 * qkv [B, 3D] -> softmax over the HEADS axis -> out[b, d*H + h]; attn [B,H,H] saved for backward */
/* One pre-norm Block of the RNA transformer (reference: Block.forward models/mirror.py:149-152, Attention.forward :77-102,
 * [3P] timm Mlp fc1 -> GELU -> drop -> fc2 -> drop) on [B <= 32, D] rows, forward and backward, ONE call per direction:
 *   x1 = x + drop(proj(headattn(qkv(LN1(x)))));  y = x1 + drop(fc2(drop(gelu(fc1(LN2(x1))))))
 * LayerNorm, bias, GELU, the three dropouts (Philox, regenerated in the backward) and the residual adds ride in the
 * prologues / epilogues of the weight-streaming GEMM kernels (5 launches forward, 6 backward; csrc/rna_block.hip).
 * Needs D % 32 == 0, Hh % 32 == 0, D % H == 0.  Weights are bf16 [N, K] row-major; wt_* are their transposes [K, N] (the
 * data gradients stream them like forward weights).  Saved tensors are caller-allocated in the forward and handed back
 * unchanged in the backward; gradients ACCUMULATE into the f32 d* buffers. */
typedef struct {
    int32_t B, D, Hh, H;          /* rows, model width, MLP hidden width, attention heads */
    float eps, p_drop;            /* LayerNorm eps; dropout probability (0 = eval mode) */
    uint64_t seed, offset;        /* Philox streams: proj output at `offset`, fc1 activation at + q4(B D), fc2 output at + q4(B Hh) more (q4 = round up to 4) */
    const uint64_t* dev_base;     /* optional device-side base added to `offset` (per-step base of a replayed HIP graph) */
    const void *w_qkv, *w_proj, *w_fc1, *w_fc2;         /* bf16 [3D, D], [D, D], [Hh, D], [D, Hh] */
    const void *wt_qkv, *wt_proj, *wt_fc1, *wt_fc2;     /* bf16 transposes (backward only) */
    const float *b_qkv, *b_proj, *b_fc1, *b_fc2;        /* biases (b_qkv may be NULL) */
    const float *g1, *be1, *g2, *be2;                   /* LayerNorm 1 / 2 weight and bias */
    const float* x;               /* [B, D] f32 block input */
    float* y;                     /* [B, D] f32 block output (forward) */
    float* stats;                 /* saved: [4, B] mean1, rstd1, mean2, rstd2 */
    void* qkv;                    /* saved: bf16 [B, 3D] */
    float* attn;                  /* saved: f32 [B, H, H] attention probabilities */
    void* o;                      /* saved: bf16 [B, D] attention output */
    float* x1;                    /* saved: f32 [B, D] stream after the attention half */
    void* u;                      /* saved: bf16 [B, Hh] fc1 output before GELU */
    void* f;                      /* saved: bf16 [B, Hh] fc2 input */
    const float* dy;              /* backward: [B, D] f32 gradient of y */
    float* dx;                    /* backward: [B, D] f32 gradient of x (written) */
    float *dw_qkv, *dw_proj, *dw_fc1, *dw_fc2, *db_qkv, *db_proj, *db_fc1, *db_fc2, *dg1, *dbe1, *dg2, *dbe2;
    void* scratch;                /* backward: mh_rna_block_workspace_bytes(B, D, Hh) bytes */
} mh_rna_block;
int mh_rna_block_fwd(const mh_rna_block* blk, mh_stream s);
int mh_rna_block_bwd(const mh_rna_block* blk, mh_stream s);
int64_t mh_rna_block_workspace_bytes(int B, int D, int Hh);

int mh_headattn_fwd(const void* qkv, void* out, float* attn, int B, int H, int hd, int dt, mh_stream s);
int mh_headattn_bwd(const void* qkv, const float* attn, const void* dout, void* dqkv, int B, int H, int hd,
                    int dt, mh_stream s);

/* ---------------------------------------------------------------- elementwise / reductions */
int mh_add(const void* a, const void* b, void* y, int64_t n, int dt_a, int dt_b, int dt_y, mh_stream s);
int mh_cast(const void* x, void* y, int64_t n, int dt_x, int dt_y, mh_stream s);
int mh_gelu_fwd(const void* x, void* y, int64_t n, int dt_x, int dt_y, mh_stream s);
int mh_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dt_x, int dt_dy, int dt_dx, mh_stream s);
/* dx = dy * (y > 0); `batches` blocks of n_per_batch contiguous elements at the given batch strides */
int mh_relu_bwd(const void* y, const void* dy, void* dx, int64_t n_per_batch, int batches, int64_t y_bs, int64_t dy_bs,
                int64_t dx_bs, int dt_y, int dt_dy, int dt_dx, mh_stream s);
/* The four random draws of a training step in one launch (models/mirror.py:630 `torch.rand(B, N)`, :516 `torch.rand(B, D)`, :832-833
 * `torch.randn_like` twice): out[0, n_uniform) uniform in [0, 1) on 24 random bits, out[n_uniform, n_uniform + n_normal) standard normal
 * (Box-Muller); Philox4x32-10 on the dropout stream: element i = word (i & 3) of block (offset + *dev_base + i) >> 2 under `seed`.
 * n_uniform and offset are multiples of 4.  The VALUES differ from torch's generator (as any seed change would); callers that pin the
 * draws (parity tests) pass them in instead. */
int mh_noise_draws(float* out, int64_t n_uniform, int64_t n_normal, uint64_t seed, uint64_t offset, const uint64_t* dev_base, mh_stream s);
/* y = x * keep/(1-p); keep from Philox4x32-10(seed, offset + *dev_base + i)  ([3P] nn.Dropout in to_out; :75, :142).
 * dev_base (nullable, device memory): per-step base offset, so that a captured graph draws fresh masks at every replay */
int mh_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, uint64_t offset, const uint64_t* dev_base,
               int dt_x, int dt_y, mh_stream s);
/* y = a + dropout(x): the residual add behind the Dropout of [3P] NystromAttention.to_out (models/mirror.py:312), one pass.
 * Same mask as mh_dropout for the same (seed, offset [+ *dev_base]).  n % 4 == 0, quad-aligned buffers; a, y f32. */
int mh_dropout_add(const float* a, const void* x, float* y, int64_t n, float p, uint64_t seed, uint64_t offset,
                   const uint64_t* dev_base, int dt_x, mh_stream s);
/* The "lite" dropout stream of the [B, n, D]-sized WSI dropouts ([3P] to_out[1], models/mirror.py:312): Philox4x32-7, 16 random
 * bits per element (8 elements per block: element i = 16-bit field i & 7 of the block with counter (offset + i) >> 3), kept when
 * field >= round(p * 65536), scaled by the exact 1 / P(keep).  y = dropout(x) (a NULL) or a + dropout(x) (y f32);
 * n % 8 == 0, offset % 8 == 0.  The DROPADD projection epilogue (mh_gemm_epi) draws the same stream. */
int mh_dropout_lite(const float* a, const void* x, void* y, int64_t n, float p, uint64_t seed, uint64_t offset,
                    const uint64_t* dev_base, int dt_x, int dt_y, mh_stream s);
/* Timeline probe: *dst = the device's constant-rate wall clock (100 MHz ticks) when this one-thread launch runs on stream s.
 * Profiling aid (tools/exp/probe_timeline.py: where the branches of a replayed step really start and end, without a tracer
 * attached); nothing on the product path calls it. */
int mh_timestamp(uint64_t* dst, mh_stream s);
/* mh_dropout_lite's backward use on a [rows, N] f32 gradient -> bf16, which also adds the column sums of what it wrote to db [N]:
 * the bias gradient of [3P] to_out[0] (models/mirror.py:312) without mh_colsum's pass over the bf16 gradient. */
int mh_dropout_lite_colsum(const float* x, void* y, int64_t rows, int N, float p, uint64_t seed, uint64_t offset,
                           const uint64_t* dev_base, float* db, mh_stream s);
/* out[c] += sum_r x[r*ld + c]  (bias gradients; f32 atomics) */
int mh_colsum(const void* x, float* out, int64_t rows, int cols, int64_t ld, int dt, mh_stream s);
/* y[r] = x[r*x_rs .. +D] / max(||.||, eps) (F.normalize, models/mirror.py:540, :683); norm[r] saved */
int mh_l2norm_fwd(const void* x, void* y, float* nrm, int rows, int D, int64_t x_rs, float eps, int dt_x, int dt_y,
                  mh_stream s);
int mh_l2norm_bwd(const void* y, const float* nrm, const void* dy, void* dx, int rows, int D, int64_t dx_rs,
                  float eps, int dt_y, int dt_dy, int dt_dx, int accumulate, mh_stream s);
/* y = exp(x) on a small f32 tensor (`logit_scale.exp()`, models/mirror.py:911); backward dx (+)= dy * y (accumulate != 0: summed
   into dx, the parameter's gradient slot) */
int mh_exp_fwd(const float* x, float* y, int64_t n, mh_stream s);
int mh_exp_bwd(const float* dy, const float* y, float* dx, int64_t n, int accumulate, mh_stream s);
/* z = mu + eps*exp(0.5*logstd) (models/mirror.py:830-833); backward: dmu = dz + add_mu, dlogstd = dz*eps*0.5*exp(0.5*logstd) +
   add_logstd (add_* nullable: what reached mu / logstd from their other consumer, the KL term of losses/mirror_loss.py:112-117) */
int mh_reparam_fwd(const float* mu, const float* logstd, const float* eps, float* z, int64_t n, mh_stream s);
int mh_reparam_bwd(const float* logstd, const float* eps, const float* dz, const float* add_mu, const float* add_logstd,
                   float* dmu, float* dlogstd, int64_t n, mh_stream s);

/* ---------------------------------------------------------------- losses (losses/mirror_loss.py, losses/info_nce.py)
 * cross-entropy of rows of scale*G against label (label_off + row); G [R x C] f32.
 * out[0] += coef * sum_r (lse_r - scale*G[r,label]) ; loss_rows[r] = coef * (...) ; lse [R] saved.
 * scale is read from device memory (no host sync). */
int mh_ce_rows_fwd(const float* G, int64_t ldg, const float* scale, float scale_mul, int R, int C, int label_off,
                   float coef, float* loss_rows, float* lse, float* out, mh_stream s);
/* dG[r,c] = gcoef * g[0 or r] * scale * (softmax - onehot); dscale += sum dG_unscaled*G  */
int mh_ce_rows_bwd(const float* G, int64_t ldg, const float* scale, float scale_mul, const float* lse, const float* g,
                   int g_per_row, float gcoef, float* dG, float* dscale, int R, int C, int label_off, mh_stream s);
/* acc[0] += sum_r mask[r] * mean_D (p-t)^2 ; acc[1] += sum_r mask[r]   (losses/mirror_loss.py:98-103)
 * pred [rows, D] contiguous; target row r at tgt + (r / rows_per_batch) * tgt_bs + (r % rows_per_batch) * D elements
 * (a row window of a larger buffer: the WSI target is encoder_output[:, 1:], models/mirror.py:700) */
int mh_mse_masked_fwd(const void* pred, const void* tgt, const float* mask, float* acc, int64_t rows, int D,
                      int64_t rows_per_batch, int64_t tgt_bs, int dt_p, int dt_t, mh_stream s);
/* dpred[rows, D] (dt_dp) = g[0] * gmul * 2*mask*(p-t)/(D*acc[1]) (gmul: the term's loss weight, host constant); dtgt[rows, D] (dt_t) = -dpred, not written when NULL.
 * colsum_ws [cs_blocks, D] f32 (may be NULL; bf16 pred / f32 target / bf16 dpred, D a multiple of 256 up to 1024): the launch runs cs_blocks blocks and block i
 * leaves the column sums of the dpred rows it wrote in row i (every row is written, nothing to zero): mh_colsum over that table is the
 * bias gradient of the Linear that produced pred (retention_head, models/mirror.py:698-699) without a second pass over dpred. */
int mh_mse_masked_bwd(const void* pred, const void* tgt, const float* mask, const float* acc, const float* g, float gmul,
                      void* dpred, void* dtgt, int64_t rows, int D, int64_t rows_per_batch, int64_t tgt_bs, int dt_p, int dt_t,
                      int dt_dp, float* colsum_ws, int cs_blocks, mh_stream s);
/* Data feed (datasets/dataset_pretrain.py:150-167, `wsi_feature[sampled_indices]`): out[r, :] = src[row[r], :] for R rows of F
 * elements; src is the bank of all slides' patch features back to back ([src_rows, F]); row holds GLOBAL row indices
 * (slide offset + sampled index; clamped to the bank). */
int mh_gather_rows(const void* src, const int64_t* row, void* out, int64_t R, int64_t F, int64_t src_rows, int dt, mh_stream s);
/* Gradient of the WSI encoder output E [B, T, D] (f32), which three consumers read (models/mirror.py:684, :690, :700, :833):
 *   dE[b, t] = gfull[b, t] + alpha * x[b, t - 1] (t >= 1) + (t == 0 ? c[b] : 0)
 * gfull f32 [B, T, D], x [B, T-1, D] in dt_x, c f32 [B, D]; each may be NULL (taken as zero). */
int mh_fanout_bwd(const float* gfull, const void* x, float alpha, const float* c, float* dE, int B, int T, int D, int dt_x,
                  mh_stream s);
/* total = sum_i w_i * term_i over up to 6 separate f32 scalars (losses/mirror_loss.py:121-127); bwd: dterms[i] = w_i * g[0] */
int mh_weighted_sum(const float* t0, const float* t1, const float* t2, const float* t3, const float* t4, const float* t5,
                    float w0, float w1, float w2, float w3, float w4, float w5, int n, float* out, mh_stream s);
int mh_weighted_sum_bwd(const float* g, float w0, float w1, float w2, float w3, float w4, float w5, int n, float* dterms,
                        mh_stream s);
/* out[0] += coef * sum (exp(ls) + mu^2 - 1 - ls)   (losses/mirror_loss.py:105-112) */
int mh_kl_fwd(const float* mu, const float* ls, float* out, int64_t n, float coef, mh_stream s);
int mh_kl_bwd(const float* mu, const float* ls, const float* g, float* dmu, float* dls, int64_t n, float coef,
              mh_stream s);
/* out[0] += coef * sum_b sum_k (p_r - p_w)(log p_r - log p_w)  (losses/mirror_loss.py:114-119) */
int mh_symkl_fwd(const float* w, const float* r, float* out, int B, int P, float coef, mh_stream s);
int mh_symkl_bwd(const float* w, const float* r, const float* g, float* dw, float* dr, int B, int P, float coef,
                 mh_stream s);

/* MIRRORLoss's small terms in one launch each way (losses/mirror_loss.py:74-135, everything but the WSI retention MSE):
 * the rank-local CLIP alignment term (:37-52: logits s W R^T, both cross-entropies, and in the backward the gradient
 * products d W = P R, d R = P^T W and d s), the RNA retention MSE (:98-103), both style KLs (:105-112), the cluster KL
 * (:114-119) and the weighted total (:121-127).  All tensors f32, contiguous.  weight[] = {alignment, wsi retention,
 * rna retention, style (wsi), style (rna), cluster}.
 *   forward : out[8] = {total, alignment, wsi retention, rna retention, style_w + style_r, cluster, style_w, style_r};
 *             wsi_acc = the {num, den} pair mh_mse_masked_fwd accumulated on the same stream BEFORE this call (NULL: 0);
 *             scratch = 8 floats zeroed by the caller (kept for the backward), save = B*B + 2B floats (kept as well).
 *   backward: the upstream of term i is weight[i] * g_total[0] + (g_terms ? g_terms[i] : 0); every d_* of a term that is
 *             present is overwritten.  The WSI retention gradient stays mh_mse_masked_bwd(g = g_total, gmul = weight[1]).
 * has_align = 0: the alignment term was computed by the caller over the gathered batch (align_ext, one float, may be
 * NULL = 0) and the backward writes its upstream to d_align_ext.  Alignment in here needs B <= 32 and
 * 2 B (D + 1) + B (B + 1) floats of LDS <= 150 KiB. */
typedef struct {
    const float* wsi_emb;      /* [B, D] L2-normalised alignment embeddings */
    const float* rna_emb;      /* [B, D] */
    const float* logit_scale;  /* one float: exp(logit_scale parameter) */
    const float* align_ext;
    int32_t B, D, has_align;
    const float* rna_pred;     /* [n_rna] reconstructed expression, target, mask (1 = masked gene) */
    const float* rna_tgt;
    const float* rna_mask;
    int64_t n_rna;
    const float* w_mu;         /* style posteriors: [rows, latent] each */
    const float* w_logstd;
    const float* r_mu;
    const float* r_logstd;
    int64_t n_wstyle, n_rstyle;
    int32_t rows_wstyle, rows_rstyle;
    const float* w_score;      /* prototype scores [Bc, P] (logits; the kernel takes the softmax) */
    const float* r_score;
    int32_t Bc, P;
    const float* wsi_acc;
    float weight[6];
    float* scratch;
    float* save;
    float* out;
    const float* g_total;      /* backward only from here */
    const float* g_terms;
    float* d_wsi_emb;
    float* d_rna_emb;
    float* d_logit_scale;
    float* d_align_ext;
    float* d_rna_pred;
    float* d_rna_tgt;          /* = -d_rna_pred; NULL: not written (the target is data) */
    float* d_w_mu;
    float* d_w_logstd;
    float* d_r_mu;
    float* d_r_logstd;
    float* d_w_score;
    float* d_r_score;
} mh_loss_terms;
int mh_loss_terms_fwd(const mh_loss_terms* d, mh_stream s);
int mh_loss_terms_bwd(const mh_loss_terms* d, mh_stream s);

/* ---------------------------------------------------------------- step glue (train_mirror.py:1133-1136, :1230, :1254-1255) */
/* w[r] /= max(||w[r]||, eps) in place (the prototype renormalisation, train_mirror.py:1133-1136); shadow_bf16 (nullable, [rows, D]
 * contiguous) receives the bf16 copy of the result in the same pass */
int mh_rownorm_(float* w, void* shadow_bf16, int rows, int D, float eps, mh_stream s);
int mh_clamp_(float* x, int64_t n, float lo, float hi, mh_stream s);
/* torch.optim.Adam (wd=0): flat f32 params/grads/moments; optional bf16 shadow copy of the params.
 * dev_state (nullable, 6 device floats {t, 1-b1^t, 1-b2^t, lr, clip, |g|}): when given, t is advanced and the bias
 * corrections are refreshed ON THE DEVICE before the update, lr / bias_c1 / bias_c2 arguments are ignored and the
 * gradient is additionally scaled by dev_state[4] (1, or the factor mh_grad_clip left there) — nothing step-dependent
 * is a launch argument, so the whole step can be captured in a HIP graph (train_mirror.py:1254 optimizer.step()).
 * Step glue that rides along instead of costing launches of its own: clamp_index >= 0 clamps that one parameter to
 * [clamp_lo, clamp_hi] right behind its update, master and shadow (`logit_scale.clamp_(0, ln 100)`, train_mirror.py:1255; -1 =
 * none); counter (nullable) += counter_add on the device (the dropout streams' per-step base).
 * One optimizer step as TWO launches (round 5: the RNA encoder's parameters, 80 % of the arena, have their gradients 2 ms before the
 * step's last one): tick = 0 reads dev_state without advancing it (the other launch of the step did), and elements [hole_lo, hole_hi)
 * (quad-aligned, not holding clamp_index) are left untouched — the range the other launch updates.  tick = 1, hole_lo = hole_hi = 0:
 * the whole arena in one launch, as before.  tick = 2: like 1, for the EARLY launch of such a pair (it runs as `adam_range_kernel`, so that
 * profiling tools that cut a kernel trace into steps at `adam_kernel` keep working). */
int mh_adam(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
            float beta2, float eps, float bias_c1, float bias_c2, float grad_scale, float* dev_state, int64_t clamp_index,
            float clamp_lo, float clamp_hi, int64_t* counter, int64_t counter_add, int tick, int64_t hole_lo, int64_t hole_hi,
            mh_stream s);

/* clip-grad "norm" mode (train_mirror.py:1206-1230): dev_state[5] = ||grad_scale * g||_2, dev_state[4] =
 * min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0); scratch1 = one device float. */
int mh_grad_clip(const float* g, int64_t n, float grad_scale, float max_norm, float* scratch1, float* dev_state, mh_stream s);

#ifdef __cplusplus
}
#endif
#endif
