// Fused Nystrom attention sides ([3P] NystromAttention.forward, called at models/mirror.py:312) for the TransMIL
// geometry dh = 64, m = 256 landmarks, bf16 operands.
//
// The two big similarity matrices  sim1 = q k_l^T  [n_p x m]  and  sim3 = q_l k^T  [m x n_p]  are never written to
// HBM.  As separate GEMM / softmax / GEMM launches each of them costs ~1.7 GB of traffic per layer at B = 16
// (f32 logits out, softmax in/out, probabilities in again) for 36 GFLOP of work; fused, the probabilities live in
// MFMA accumulators from the first product to the second and the traffic is q, k, v, out (~0.3 GB).
//
//   attn1 side:  out = softmax_m(scale q k_l^T) w2                (w2 = pinv(attn2) (attn3 v), [m x dh])
//   attn3 side:  av  = softmax_n(scale q_l k^T) v                 (online softmax over the n_p keys)
//
// Register-only data flow: v_mfma_f32_32x32x16_bf16 leaves C[i][j] with j = lane & 31 and i in registers
// (i = 8 (r >> 2) + 4 (lane >> 5) + (r & 3)).  That is exactly the operand layout of a matrix whose CONTRACTION index
// is i, so a tile of probabilities is fed straight back as an operand: registers 8t..8t+7 form the bf16x8 fragment of
// k-step t, and the other operand is read from LDS in the same permuted k order (rows kb + 4 hl + {0..3} and
// kb + 8 + 4 hl + {0..3}) with ds_read_b64_tr_b16.  Which index must be contracted decides the orientation:
//   N kernels (one 128-row tile of the sequence per workgroup, n in lanes, all 256 landmarks in registers):
//       attn1 fwd, attn1 bwd dq (+ delta), attn3 bwd dk/dv
//   L kernels (wave owns 64 landmark columns in lanes and walks the sequence, n in registers):
//       attn3 fwd (online softmax), attn1 bwd dw2/dk_l, attn3 bwd dq_l
// Layout (SURVEY.md §8 / DESIGN.md §4): qkv [B, n_p, 3D] bf16, heads are 64-wide column slices; landmarks lm
// [B, m, 2D] = q_l | k_l; w2, av, dav [B, h, m, 64]; out / dout [B, n_p, D].
#include "gemm_kernel.h"

namespace {

constexpr int NM = 256;   // landmarks
constexpr int ND = 64;    // head dim
constexpr int NP = 72;    // LDS pitch in bf16 of every [rows][64] image: ds_read_b128 fragments conflict free
constexpr int NT = 256;   // threads per workgroup (4 waves)
constexpr int TR = 128;   // sequence rows per tile

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef short s16x8 __attribute__((ext_vector_type(8)));

// A/B fragment, contraction index contiguous in the image: row (row0 + lane&31), k = k0 + 8 hl + {0..7}
__device__ __forceinline__ bf16x8 frag_kc(const bf16_t* img, int row0, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + (row0 + (lane & 31)) * NP + k0 + 8 * (lane >> 5));
}
// A/B fragment, contraction index = image ROW, free index = image column (col0 + lane&31), in the accumulator's k
// order: element j <-> image row kb + 4 hl + j (j < 4), kb + 8 + 4 hl + (j - 4) (j >= 4)
__device__ __forceinline__ bf16x8 frag_tr(const bf16_t* img, int col0, int kb, int lane) {
    const int g16 = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const bf16_t* a0 = img + (kb + 4 * (g16 >> 1) + q) * NP + col0 + 16 * (g16 & 1) + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 8 * NP));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// fragment straight from global memory: 8 consecutive bf16 of a row
__device__ __forceinline__ bf16x8 frag_g(const bf16_t* rowptr, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(rowptr + k0 + 8 * (lane >> 5));
}
// accumulator registers 8t..8t+7 -> bf16x8 operand fragment of k-step t
template <int T>
__device__ __forceinline__ bf16x8 pack8(const f32x16& a) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = (__bf16)a[8 * T + j];
    return r;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0.f;
    return z;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// ROWS x 64 bf16 (row stride ld) -> pitch-NP image
template <int ROWS>
__device__ __forceinline__ void stage_rows(bf16_t* img, const bf16_t* __restrict__ src, long ld, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; i++) {
        const int cid = tid + i * NT, r = cid >> 3, c = cid & 7;
        *reinterpret_cast<u32x4*>(img + r * NP + c * 8) = *reinterpret_cast<const u32x4*>(src + (long)r * ld + c * 8);
    }
}
// the same in two halves so the next tile can wait in registers while the current one is consumed
template <int ROWS>
__device__ __forceinline__ void tile_load(u32x4 (&regs)[ROWS * 8 / NT], const bf16_t* __restrict__ src, long ld, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; i++) {
        const int cid = tid + i * NT, r = cid >> 3, c = cid & 7;
        regs[i] = *reinterpret_cast<const u32x4*>(src + (long)r * ld + c * 8);
    }
}
template <int ROWS>
__device__ __forceinline__ void tile_store(const u32x4 (&regs)[ROWS * 8 / NT], bf16_t* img, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; i++) {
        const int cid = tid + i * NT, r = cid >> 3, c = cid & 7;
        *reinterpret_cast<u32x4*>(img + r * NP + c * 8) = regs[i];
    }
}
// accumulator rows 8g + 4hl + {0..3} of a [d][n]-oriented result -> 4 consecutive bf16 of the n-th row
__device__ __forceinline__ void store_row4(bf16_t* p, const f32x16& a, int g) {
    u32x2 w;
    w[0] = (unsigned)f2bf(a[4 * g + 0]) | ((unsigned)f2bf(a[4 * g + 1]) << 16);
    w[1] = (unsigned)f2bf(a[4 * g + 2]) | ((unsigned)f2bf(a[4 * g + 3]) << 16);
    *reinterpret_cast<u32x2*>(p) = w;
}

struct Geo {
    int h, n_p, D;
    float scale;
    int accumulate;     // attn1 forward: add to `out` (the res_conv term is already there) instead of overwriting it
};

// ============================================================================ attn1 forward (N kernel)
// grid (n_p / 128, B h).  out[b, n, hd*64 + d] = sum_l softmax_l(scale q k_l^T)[n, l] w2[l, d];  lse1 = row logsumexp
__global__ __launch_bounds__(NT) void nys_a1_fwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                        const bf16_t* __restrict__ w2, bf16_t* __restrict__ out,
                                                        float* __restrict__ lse1, Geo g) {
    __shared__ __attribute__((aligned(16))) bf16_t s_kl[NM * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_w2[NM * NP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    stage_rows<NM>(s_kl, lm + (long)b * NM * 2 * D + D + hd * ND, 2 * D, tid);
    stage_rows<NM>(s_w2, w2 + (long)bh * NM * ND, ND, tid);
    const long row = (long)blockIdx.x * TR + wave * 32 + c;
    const bf16_t* qrow = qkv + ((long)b * g.n_p + row) * 3 * D + hd * ND;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) qf[ks] = frag_g(qrow, 16 * ks, lane);
    __syncthreads();
    // online softmax over the 8 landmark blocks: 16 logits live at a time instead of 128 (two waves per SIMD fit)
    float mrun = -INFINITY, lrun = 0.f;
    f32x16 o[2] = {zero16(), zero16()};   // O^T[d][q row]
#pragma unroll
    for (int blk = 0; blk < 8; blk++) {
        f32x16 sb = zero16();             // S^T[landmark 32 blk ..][q row]
#pragma unroll
        for (int ks = 0; ks < 4; ks++) sb = MFMA(frag_kc(s_kl, 32 * blk, 16 * ks, lane), qf[ks], sb);
        float mx = sb[0];
#pragma unroll
        for (int r = 1; r < 16; r++) mx = fmaxf(mx, sb[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx * g.scale);
        const float alpha = __expf(mrun - mnew);
        mrun = mnew;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float pv = __expf(sb[r] * g.scale - mnew);
            sb[r] = pv;
            sum += pv;
        }
        lrun = lrun * alpha + sum;        // per lane half; the halves are joined once at the end
        o[0] *= alpha;
        o[1] *= alpha;
        const bf16x8 p0 = pack8<0>(sb), p1 = pack8<1>(sb);
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
            o[nb] = MFMA(frag_tr(s_w2, 32 * nb, 32 * blk, lane), p0, o[nb]);
            o[nb] = MFMA(frag_tr(s_w2, 32 * nb, 32 * blk + 16, lane), p1, o[nb]);
        }
    }
    const float sum = lrun + __shfl_xor(lrun, 32, 64);
    if (hl == 0) lse1[(long)bh * g.n_p + row] = mrun + __logf(sum);
    const float inv = 1.f / sum;
    bf16_t* orow = out + ((long)b * g.n_p + row) * D + hd * ND;
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
        o[nb] *= inv;
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            bf16_t* p = orow + 32 * nb + 8 * gq + 4 * hl;
            if (g.accumulate) {
                const u32x2 old = *reinterpret_cast<const u32x2*>(p);
                o[nb][4 * gq + 0] += __uint_as_float(old[0] << 16);
                o[nb][4 * gq + 1] += __uint_as_float(old[0] & 0xffff0000u);
                o[nb][4 * gq + 2] += __uint_as_float(old[1] << 16);
                o[nb][4 * gq + 3] += __uint_as_float(old[1] & 0xffff0000u);
            }
            store_row4(p, o[nb], gq);
        }
    }
}

// ============================================================================ attn1 backward, dq + delta (N kernel)
// dS1 = P1 o (dO w2^T - delta) scale, delta[n] = sum_l P1 dP1;  dq = dS1 k_l
__global__ __launch_bounds__(NT) void nys_a1_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                           const bf16_t* __restrict__ w2, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ lse1, float* __restrict__ delta1,
                                                           bf16_t* __restrict__ dqkv, Geo g) {
    __shared__ __attribute__((aligned(16))) bf16_t s_kl[NM * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_w2[NM * NP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    stage_rows<NM>(s_kl, lm + (long)b * NM * 2 * D + D + hd * ND, 2 * D, tid);
    stage_rows<NM>(s_w2, w2 + (long)bh * NM * ND, ND, tid);
    const long row = (long)blockIdx.x * TR + wave * 32 + c;
    const bf16_t* qrow = qkv + ((long)b * g.n_p + row) * 3 * D + hd * ND;
    const bf16_t* grow = dout + ((long)b * g.n_p + row) * D + hd * ND;
    bf16x8 qf[4], gf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        qf[ks] = frag_g(qrow, 16 * ks, lane);
        gf[ks] = frag_g(grow, 16 * ks, lane);
    }
    const float lse = lse1[(long)bh * g.n_p + row];
    __syncthreads();
    f32x16 s[8];   // P1^T[landmark][q row]
#pragma unroll
    for (int blk = 0; blk < 8; blk++) {
        s[blk] = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ks++) s[blk] = MFMA(frag_kc(s_kl, 32 * blk, 16 * ks, lane), qf[ks], s[blk]);
#pragma unroll
        for (int r = 0; r < 16; r++) s[blk][r] = __expf(s[blk][r] * g.scale - lse);
    }
    // pass 1: delta = sum_l P dP   (dP^T[landmark][q row] = w2 dO^T, recomputed in pass 2 instead of held)
    float del = 0.f;
#pragma unroll
    for (int blk = 0; blk < 8; blk++) {
        f32x16 dp = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ks++) dp = MFMA(frag_kc(s_w2, 32 * blk, 16 * ks, lane), gf[ks], dp);
#pragma unroll
        for (int r = 0; r < 16; r++) del += s[blk][r] * dp[r];
    }
    del += __shfl_xor(del, 32, 64);
    if (hl == 0) delta1[(long)bh * g.n_p + row] = del;
    f32x16 dq[2] = {zero16(), zero16()};   // dq^T[d][q row]
#pragma unroll
    for (int blk = 0; blk < 8; blk++) {
        f32x16 dp = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ks++) dp = MFMA(frag_kc(s_w2, 32 * blk, 16 * ks, lane), gf[ks], dp);
#pragma unroll
        for (int r = 0; r < 16; r++) dp[r] = s[blk][r] * (dp[r] - del) * g.scale;
        const bf16x8 d0 = pack8<0>(dp), d1 = pack8<1>(dp);
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
            dq[nb] = MFMA(frag_tr(s_kl, 32 * nb, 32 * blk, lane), d0, dq[nb]);
            dq[nb] = MFMA(frag_tr(s_kl, 32 * nb, 32 * blk + 16, lane), d1, dq[nb]);
        }
    }
    bf16_t* drow = dqkv + ((long)b * g.n_p + row) * 3 * D + hd * ND;
#pragma unroll
    for (int nb = 0; nb < 2; nb++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) store_row4(drow + 32 * nb + 8 * gq + 4 * hl, dq[nb], gq);
}

// accumulator [rows in registers][cols in lanes] -> f32 atomics into dst[row * ld + col]
__device__ __forceinline__ void atomic_tile(float* dst, long ld, const f32x16& a, int hl, int c) {
#pragma unroll
    for (int r = 0; r < 16; r++) atomicAdd(dst + (long)(8 * (r >> 2) + 4 * hl + (r & 3)) * ld + c, a[r]);
}

// ============================================================================ attn1 backward, dw2 + dk_l (L kernel)
// grid (splits, B h); wave w owns landmarks [64 w, 64 w + 64).  dw2 = P1^T dO,  dk_l = dS1^T q  (f32 atomics)
__global__ __launch_bounds__(NT) void nys_a1_bwd_dw_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                           const bf16_t* __restrict__ w2, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ lse1, const float* __restrict__ delta1,
                                                           float* __restrict__ dw2, float* __restrict__ dlm, Geo g,
                                                           int tiles_per_wg) {
    __shared__ __attribute__((aligned(16))) bf16_t s_q[TR * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_g[TR * NP];
    __shared__ __attribute__((aligned(16))) float s_lse[TR];
    __shared__ __attribute__((aligned(16))) float s_del[TR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    const int ntiles = g.n_p / TR;
    const int t0 = blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, ntiles);
    if (t0 >= t1) return;
    const bf16_t* klb = lm + (long)b * NM * 2 * D + D + hd * ND;
    const bf16_t* w2b = w2 + (long)bh * NM * ND;
    bf16x8 klf[2][4], w2f[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int l = 64 * wave + 32 * j + c;
            klf[j][ks] = frag_g(klb + (long)l * 2 * D, 16 * ks, lane);
            w2f[j][ks] = frag_g(w2b + (long)l * ND, 16 * ks, lane);
        }
    const bf16_t* qb = qkv + (long)b * g.n_p * 3 * D + hd * ND;
    const bf16_t* gb = dout + (long)b * g.n_p * D + hd * ND;
    f32x16 adw[2][2], adk[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int nb = 0; nb < 2; nb++) adw[j][nb] = adk[j][nb] = zero16();
    u32x4 rq[TR * 8 / NT], rg[TR * 8 / NT];
    float rl = 0.f, rd = 0.f;
    tile_load<TR>(rq, qb + (long)t0 * TR * 3 * D, 3 * D, tid);
    tile_load<TR>(rg, gb + (long)t0 * TR * D, D, tid);
    if (tid < TR) {
        rl = lse1[(long)bh * g.n_p + (long)t0 * TR + tid];
        rd = delta1[(long)bh * g.n_p + (long)t0 * TR + tid];
    }
#pragma unroll 1
    for (int t = t0; t < t1; t++) {
        __syncthreads();
        tile_store<TR>(rq, s_q, tid);
        tile_store<TR>(rg, s_g, tid);
        if (tid < TR) {
            s_lse[tid] = rl;
            s_del[tid] = rd;
        }
        __syncthreads();
        if (t + 1 < t1) {
            tile_load<TR>(rq, qb + (long)(t + 1) * TR * 3 * D, 3 * D, tid);
            tile_load<TR>(rg, gb + (long)(t + 1) * TR * D, D, tid);
            if (tid < TR) {
                rl = lse1[(long)bh * g.n_p + (long)(t + 1) * TR + tid];
                rd = delta1[(long)bh * g.n_p + (long)(t + 1) * TR + tid];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {   // 32 q rows at a time
            f32x4 lr[4], dr[4];
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                lr[gq] = *reinterpret_cast<const f32x4*>(s_lse + 32 * i + 8 * gq + 4 * hl);
                dr[gq] = *reinterpret_cast<const f32x4*>(s_del + 32 * i + 8 * gq + 4 * hl);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                f32x16 s = zero16(), dp = zero16();   // S[q row][landmark], dP[q row][landmark]
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    s = MFMA(frag_kc(s_q, 32 * i, 16 * ks, lane), klf[j][ks], s);
                    dp = MFMA(frag_kc(s_g, 32 * i, 16 * ks, lane), w2f[j][ks], dp);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float p = __expf(s[r] * g.scale - lr[r >> 2][r & 3]);
                    s[r] = p;
                    dp[r] = p * (dp[r] - dr[r >> 2][r & 3]) * g.scale;
                }
                const bf16x8 p0 = pack8<0>(s), p1 = pack8<1>(s), d0 = pack8<0>(dp), d1 = pack8<1>(dp);
#pragma unroll
                for (int nb = 0; nb < 2; nb++) {
                    adw[j][nb] = MFMA(p0, frag_tr(s_g, 32 * nb, 32 * i, lane), adw[j][nb]);
                    adw[j][nb] = MFMA(p1, frag_tr(s_g, 32 * nb, 32 * i + 16, lane), adw[j][nb]);
                    adk[j][nb] = MFMA(d0, frag_tr(s_q, 32 * nb, 32 * i, lane), adk[j][nb]);
                    adk[j][nb] = MFMA(d1, frag_tr(s_q, 32 * nb, 32 * i + 16, lane), adk[j][nb]);
                }
            }
        }
    }
    float* dwb = dw2 + (long)bh * NM * ND;
    float* dkb = dlm + (long)b * NM * 2 * D + D + hd * ND;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
            const int l0 = 64 * wave + 32 * j;
            atomic_tile(dwb + (long)l0 * ND + 32 * nb, ND, adw[j][nb], hl, c);
            atomic_tile(dkb + (long)l0 * 2 * D + 32 * nb, 2 * D, adk[j][nb], hl, c);
        }
}

// ============================================================================ attn3 forward (L kernel, online softmax)
// grid (B h).  av[bh, l, d] = sum_n softmax_n(scale q_l k^T)[l, n] v[n, d];  lse3[bh, l]
__global__ __launch_bounds__(NT) void nys_a3_fwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                        float* __restrict__ av, float* __restrict__ lse3, Geo g) {
    __shared__ __attribute__((aligned(16))) bf16_t s_k[TR * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_v[TR * NP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.x, b = bh / g.h, hd = bh % g.h, D = g.D;
    const int ntiles = g.n_p / TR;
    const bf16_t* qlb = lm + (long)b * NM * 2 * D + hd * ND;
    bf16x8 qlf[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qlf[j][ks] = frag_g(qlb + (long)(64 * wave + 32 * j + c) * 2 * D, 16 * ks, lane);
    const bf16_t* kb = qkv + (long)b * g.n_p * 3 * D + D + hd * ND;
    const bf16_t* vb = kb + D;
    float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};
    f32x16 o[2][2];   // O^T[d (nb)][landmark (j)]
#pragma unroll
    for (int nb = 0; nb < 2; nb++)
#pragma unroll
        for (int j = 0; j < 2; j++) o[nb][j] = zero16();
    u32x4 rk[TR * 8 / NT], rv[TR * 8 / NT];
    tile_load<TR>(rk, kb, 3 * D, tid);
    tile_load<TR>(rv, vb, 3 * D, tid);
#pragma unroll 1
    for (int t = 0; t < ntiles; t++) {
        __syncthreads();
        tile_store<TR>(rk, s_k, tid);
        tile_store<TR>(rv, s_v, tid);
        __syncthreads();
        if (t + 1 < ntiles) {
            tile_load<TR>(rk, kb + (long)(t + 1) * TR * 3 * D, 3 * D, tid);
            tile_load<TR>(rv, vb + (long)(t + 1) * TR * 3 * D, 3 * D, tid);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            f32x16 s[4];   // S3^T[key][landmark]
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                s[i] = zero16();
#pragma unroll
                for (int ks = 0; ks < 4; ks++) s[i] = MFMA(frag_kc(s_k, 32 * i, 16 * ks, lane), qlf[j][ks], s[i]);
#pragma unroll
                for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[i][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(mrun[j], mx * g.scale);
            const float alpha = __expf(mrun[j] - mnew);
            mrun[j] = mnew;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float p = __expf(s[i][r] * g.scale - mnew);
                    s[i][r] = p;
                    sum += p;
                }
            lrun[j] = lrun[j] * alpha + sum;   // per lane half; the halves are joined once at the end
#pragma unroll
            for (int nb = 0; nb < 2; nb++) o[nb][j] *= alpha;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bf16x8 p0 = pack8<0>(s[i]), p1 = pack8<1>(s[i]);
#pragma unroll
                for (int nb = 0; nb < 2; nb++) {
                    o[nb][j] = MFMA(frag_tr(s_v, 32 * nb, 32 * i, lane), p0, o[nb][j]);
                    o[nb][j] = MFMA(frag_tr(s_v, 32 * nb, 32 * i + 16, lane), p1, o[nb][j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const float l = lrun[j] + __shfl_xor(lrun[j], 32, 64);
        const float inv = 1.f / l;
        const int lq = 64 * wave + 32 * j + c;
        if (hl == 0) lse3[(long)bh * NM + lq] = mrun[j] + __logf(l);
        float* arow = av + ((long)bh * NM + lq) * ND;
#pragma unroll
        for (int nb = 0; nb < 2; nb++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; e++) w[e] = o[nb][j][4 * gq + e] * inv;
                *reinterpret_cast<f32x4*>(arow + 32 * nb + 8 * gq + 4 * hl) = w;
            }
    }
}

// delta3[bh, l] = sum_d dav[l, d] av[l, d]: once per (b, h) instead of once per 128-row tile (34x the reads)
__global__ __launch_bounds__(256) void nys_delta3_kernel(const float* __restrict__ av, const bf16_t* __restrict__ dav,
                                                         float* __restrict__ delta3) {
    const long row = (long)blockIdx.x * NM + threadIdx.x;
    const float* ar = av + row * ND;
    const bf16_t* gr = dav + row * ND;
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < ND; e += 8) {
        const u32x4 gv = *reinterpret_cast<const u32x4*>(gr + e);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ar + e), a1 = *reinterpret_cast<const f32x4*>(ar + e + 4);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float lo = __uint_as_float(gv[w] << 16), hi = __uint_as_float(gv[w] & 0xffff0000u);
            const float x0 = w < 2 ? a0[2 * w] : a1[2 * w - 4], x1 = w < 2 ? a0[2 * w + 1] : a1[2 * w - 3];
            d += lo * x0 + hi * x1;
        }
    }
    delta3[row] = d;
}

// ============================================================================ attn3 backward, dk + dv (N kernel)
// grid (n_p / 128, B h).  P3 = exp(scale q_l k^T - lse3), dv = P3^T dav, dS3 = P3 o (dav v^T - delta3) scale, dk = dS3^T q_l
__global__ __launch_bounds__(NT) void nys_a3_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                            const float* __restrict__ delta3, const bf16_t* __restrict__ dav,
                                                            const float* __restrict__ lse3, bf16_t* __restrict__ dqkv, Geo g) {
    __shared__ __attribute__((aligned(16))) bf16_t s_ql[NM * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_g[NM * NP];
    __shared__ __attribute__((aligned(16))) float s_lse[NM];
    __shared__ __attribute__((aligned(16))) float s_del[NM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    stage_rows<NM>(s_ql, lm + (long)b * NM * 2 * D + hd * ND, 2 * D, tid);
    stage_rows<NM>(s_g, dav + (long)bh * NM * ND, ND, tid);
    s_del[tid] = delta3[(long)bh * NM + tid];       // thread = landmark
    s_lse[tid] = lse3[(long)bh * NM + tid];
    const long row = (long)blockIdx.x * TR + wave * 32 + c;
    const bf16_t* krow = qkv + ((long)b * g.n_p + row) * 3 * D + D + hd * ND;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        kf[ks] = frag_g(krow, 16 * ks, lane);
        vf[ks] = frag_g(krow + D, 16 * ks, lane);
    }
    __syncthreads();
    f32x16 adv[2] = {zero16(), zero16()}, adk[2] = {zero16(), zero16()};   // dv^T[d][key], dk^T[d][key]
#pragma unroll
    for (int blk = 0; blk < 8; blk++) {
        f32x16 s = zero16(), dp = zero16();   // S3[landmark][key], dP3[landmark][key]
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            s = MFMA(frag_kc(s_ql, 32 * blk, 16 * ks, lane), kf[ks], s);
            dp = MFMA(frag_kc(s_g, 32 * blk, 16 * ks, lane), vf[ks], dp);
        }
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            const f32x4 lr = *reinterpret_cast<const f32x4*>(s_lse + 32 * blk + 8 * gq + 4 * hl);
            const f32x4 dr = *reinterpret_cast<const f32x4*>(s_del + 32 * blk + 8 * gq + 4 * hl);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float p = __expf(s[4 * gq + e] * g.scale - lr[e]);
                s[4 * gq + e] = p;
                dp[4 * gq + e] = p * (dp[4 * gq + e] - dr[e]) * g.scale;
            }
        }
        const bf16x8 p0 = pack8<0>(s), p1 = pack8<1>(s), d0 = pack8<0>(dp), d1 = pack8<1>(dp);
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
            adv[nb] = MFMA(frag_tr(s_g, 32 * nb, 32 * blk, lane), p0, adv[nb]);
            adv[nb] = MFMA(frag_tr(s_g, 32 * nb, 32 * blk + 16, lane), p1, adv[nb]);
            adk[nb] = MFMA(frag_tr(s_ql, 32 * nb, 32 * blk, lane), d0, adk[nb]);
            adk[nb] = MFMA(frag_tr(s_ql, 32 * nb, 32 * blk + 16, lane), d1, adk[nb]);
        }
    }
    bf16_t* dkrow = dqkv + ((long)b * g.n_p + row) * 3 * D + D + hd * ND;
#pragma unroll
    for (int nb = 0; nb < 2; nb++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            store_row4(dkrow + 32 * nb + 8 * gq + 4 * hl, adk[nb], gq);
            store_row4(dkrow + D + 32 * nb + 8 * gq + 4 * hl, adv[nb], gq);
        }
}

// ============================================================================ attn3 backward, dq_l (L kernel)
// grid (splits, B h).  dq_l[l, d] += sum_n dS3[l, n] k[n, d]   (f32 atomics into the q_l half of dlm)
__global__ __launch_bounds__(NT) void nys_a3_bwd_dql_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                            const float* __restrict__ delta3, const bf16_t* __restrict__ dav,
                                                            const float* __restrict__ lse3, float* __restrict__ dlm, Geo g,
                                                            int tiles_per_wg) {
    __shared__ __attribute__((aligned(16))) bf16_t s_k[TR * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_v[TR * NP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    const int ntiles = g.n_p / TR;
    const int t0 = blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, ntiles);
    if (t0 >= t1) return;
    const bf16_t* qlb = lm + (long)b * NM * 2 * D + hd * ND;
    bf16x8 qlf[2][4], gf[2][4];
    float lsev[2], delv[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int lq = 64 * wave + 32 * j + c;
        const bf16_t* gr = dav + ((long)bh * NM + lq) * ND;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            qlf[j][ks] = frag_g(qlb + (long)lq * 2 * D, 16 * ks, lane);
            gf[j][ks] = frag_g(gr, 16 * ks, lane);
        }
        delv[j] = delta3[(long)bh * NM + lq];
        lsev[j] = lse3[(long)bh * NM + lq];
    }
    const bf16_t* kb = qkv + (long)b * g.n_p * 3 * D + D + hd * ND;
    const bf16_t* vb = kb + D;
    f32x16 acc[2][2];   // dq_l[landmark (j block, registers)][d (nb block, lanes)]
#pragma unroll
    for (int nb = 0; nb < 2; nb++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[nb][j] = zero16();
    u32x4 rk[TR * 8 / NT], rv[TR * 8 / NT];
    tile_load<TR>(rk, kb + (long)t0 * TR * 3 * D, 3 * D, tid);
    tile_load<TR>(rv, vb + (long)t0 * TR * 3 * D, 3 * D, tid);
#pragma unroll 1
    for (int t = t0; t < t1; t++) {
        __syncthreads();
        tile_store<TR>(rk, s_k, tid);
        tile_store<TR>(rv, s_v, tid);
        __syncthreads();
        if (t + 1 < t1) {
            tile_load<TR>(rk, kb + (long)(t + 1) * TR * 3 * D, 3 * D, tid);
            tile_load<TR>(rv, vb + (long)(t + 1) * TR * 3 * D, 3 * D, tid);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                f32x16 s = zero16(), dp = zero16();   // S3^T[key][landmark], dP3^T[key][landmark]
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    s = MFMA(frag_kc(s_k, 32 * i, 16 * ks, lane), qlf[j][ks], s);
                    dp = MFMA(frag_kc(s_v, 32 * i, 16 * ks, lane), gf[j][ks], dp);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) dp[r] = __expf(s[r] * g.scale - lsev[j]) * (dp[r] - delv[j]) * g.scale;
                // dS3^T in the accumulator layout IS dS3 as an A operand (row = landmark = lane, k = keys): the product
                // comes out as dq_l[landmark (registers)][d (lanes)], so the final atomics are 128-byte coalesced
                const bf16x8 d0 = pack8<0>(dp), d1 = pack8<1>(dp);
#pragma unroll
                for (int nb = 0; nb < 2; nb++) {
                    acc[nb][j] = MFMA(d0, frag_tr(s_k, 32 * nb, 32 * i, lane), acc[nb][j]);
                    acc[nb][j] = MFMA(d1, frag_tr(s_k, 32 * nb, 32 * i + 16, lane), acc[nb][j]);
                }
            }
    }
    float* dqb = dlm + (long)b * NM * 2 * D + hd * ND;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int nb = 0; nb < 2; nb++)
            atomic_tile(dqb + (long)(64 * wave + 32 * j) * 2 * D + 32 * nb, 2 * D, acc[nb][j], hl, c);
}

int check_geo(const char* fn, int B, int h, int n_p, int m, int dh) {
    MH_REQUIRE(m == NM && dh == ND, "%s: built for m = %d landmarks and dh = %d (got m=%d dh=%d); other shapes use mh_gemm + mh_softmax",
               fn, NM, ND, m, dh);
    MH_REQUIRE(B >= 0 && h >= 1 && n_p >= NM && n_p % NM == 0, "%s: n_p=%d must be a positive multiple of %d", fn, n_p, NM);
    return MH_OK;
}

}  // namespace

extern "C" int mh_nys_attn1_fwd(const void* qkv, const void* lm, const void* w2, void* out, float* lse1, int B, int h, int n_p,
                                int m, int dh, float scale, int accumulate, mh_stream s) {
    if (int e = check_geo("mh_nys_attn1_fwd", B, h, n_p, m, dh)) return e;
    if (B == 0) return MH_OK;
    const Geo g{h, n_p, h * ND, scale, accumulate};
    hipLaunchKernelGGL(nys_a1_fwd_kernel, dim3(n_p / TR, B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv, (const bf16_t*)lm,
                       (const bf16_t*)w2, (bf16_t*)out, lse1, g);
    MH_LAUNCH_CHECK("mh_nys_attn1_fwd");
    return MH_OK;
}

extern "C" int mh_nys_attn3_fwd(const void* qkv, const void* lm, float* av, float* lse3, int B, int h, int n_p, int m, int dh,
                                float scale, mh_stream s) {
    if (int e = check_geo("mh_nys_attn3_fwd", B, h, n_p, m, dh)) return e;
    if (B == 0) return MH_OK;
    const Geo g{h, n_p, h * ND, scale, 0};
    hipLaunchKernelGGL(nys_a3_fwd_kernel, dim3(B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv, (const bf16_t*)lm, av, lse3, g);
    MH_LAUNCH_CHECK("mh_nys_attn3_fwd");
    return MH_OK;
}

// splits: workgroups per (b, h) for the landmark-owner kernels (they flush with f32 atomics)
static int pick_splits(int BH, int ntiles) {
    int splits = 1;
    while (BH * splits < 512 && splits * 2 <= ntiles) splits *= 2;
    return splits;
}

extern "C" int mh_nys_attn1_bwd(const void* qkv, const void* lm, const void* w2, const void* dout, const float* lse1,
                                float* delta1, void* dqkv, float* dw2, float* dlm, int B, int h, int n_p, int m, int dh,
                                float scale, mh_stream s) {
    if (int e = check_geo("mh_nys_attn1_bwd", B, h, n_p, m, dh)) return e;
    if (B == 0) return MH_OK;
    const Geo g{h, n_p, h * ND, scale, 0};
    hipLaunchKernelGGL(nys_a1_bwd_dq_kernel, dim3(n_p / TR, B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv,
                       (const bf16_t*)lm, (const bf16_t*)w2, (const bf16_t*)dout, lse1, delta1, (bf16_t*)dqkv, g);
    MH_LAUNCH_CHECK("mh_nys_attn1_bwd(dq)");
    const int ntiles = n_p / TR, splits = pick_splits(B * h, ntiles), tpw = (ntiles + splits - 1) / splits;
    hipLaunchKernelGGL(nys_a1_bwd_dw_kernel, dim3(splits, B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv,
                       (const bf16_t*)lm, (const bf16_t*)w2, (const bf16_t*)dout, lse1, (const float*)delta1, dw2, dlm, g, tpw);
    MH_LAUNCH_CHECK("mh_nys_attn1_bwd(dw)");
    return MH_OK;
}

extern "C" int mh_nys_attn3_bwd(const void* qkv, const void* lm, const float* av, const void* dav, const float* lse3, float* delta3,
                                void* dqkv, float* dlm, int B, int h, int n_p, int m, int dh, float scale, mh_stream s) {
    if (int e = check_geo("mh_nys_attn3_bwd", B, h, n_p, m, dh)) return e;
    if (B == 0) return MH_OK;
    const Geo g{h, n_p, h * ND, scale, 0};
    hipLaunchKernelGGL(nys_delta3_kernel, dim3(B * h), dim3(NM), 0, (hipStream_t)s, av, (const bf16_t*)dav, delta3);
    hipLaunchKernelGGL(nys_a3_bwd_dkv_kernel, dim3(n_p / TR, B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv,
                       (const bf16_t*)lm, (const float*)delta3, (const bf16_t*)dav, lse3, (bf16_t*)dqkv, g);
    MH_LAUNCH_CHECK("mh_nys_attn3_bwd(dkv)");
    const int ntiles = n_p / TR, splits = pick_splits(B * h, ntiles), tpw = (ntiles + splits - 1) / splits;
    hipLaunchKernelGGL(nys_a3_bwd_dql_kernel, dim3(splits, B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv,
                       (const bf16_t*)lm, (const float*)delta3, (const bf16_t*)dav, lse3, dlm, g, tpw);
    MH_LAUNCH_CHECK("mh_nys_attn3_bwd(dql)");
    return MH_OK;
}
