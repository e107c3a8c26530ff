// One pre-norm Block of the RNA transformer (models/mirror.py:105-152 with Attention :50-102 and [3P] timm Mlp) behind ONE
// C-ABI call per direction: mh_rna_block_fwd / mh_rna_block_bwd.
//
//   x1 = x  + drop(proj(headattn(qkv(LN1(x)))))        x2 = x1 + drop(fc2(drop(gelu(fc1(LN2(x1))))))
//
// Every tensor is [B, D] with B = the per-GPU batch (<= 32): the four Linears are weight-streaming problems (6.3 MB of
// bf16 weights per block at D = 512, Hh = 2048) and everything between them is B x D elementwise work.  The un-fused
// path ran a block as ~12 forward and ~20 backward launches (LayerNorm, GELU, three dropouts, residual adds, casts, one
// launch per bias / weight gradient).  Here the elementwise work rides in the prologues / epilogues of the GEMM kernels:
//
//   forward  (5 launches)  [LN1 -> qkv + bias]  [heads attention]  [proj + bias + dropout + residual]
//                          [LN2 -> fc1 + bias -> GELU -> dropout]  [fc2 + bias + dropout + residual]
//   backward (6 launches)  [dropout' -> fc2 dgrad -> GELU' dropout'  |  fc2 wgrad + bias grad]
//                          [fc1 dgrad                                |  fc1 wgrad (LN2 recomputed) + bias grad]
//                          [LN2' + residual -> dropout' -> proj dgrad |  proj wgrad + bias grad | LN2 gamma / beta grads]
//                          [heads attention']  [qkv dgrad | qkv wgrad (LN1 recomputed) + bias grad]  [LN1' + residual, LN1 grads]
//
// A data-gradient launch and the weight-gradient launch of the same Linear are ONE grid (role by blockIdx): both read the
// same B x N gradient matrix, which every workgroup rebuilds from its source (incoming gradient, dropout mask regenerated
// from Philox, LayerNorm backward) in its prologue instead of a separate kernel materialising it.
// Stage boundaries are kernel boundaries on purpose: every stage needs ALL columns of the previous one, and a dependent
// launch costs ~1.5-1.9 us on this chip against 4-7 us for an in-kernel grid barrier (MI355X_MICROARCH.md price list).
//
// GEMM core: v_mfma_f32_16x16x32_bf16, the B x K operand staged once in LDS (bf16), weights HBM -> VGPR in B-fragment layout
// (one 16-byte load of a weight row per lane and k-step), one workgroup per 16 output columns, the 4 waves split K.
#include "common.h"

typedef __bf16 rb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float rb_f4 __attribute__((ext_vector_type(4)));
typedef unsigned rb_u4 __attribute__((ext_vector_type(4)));
typedef unsigned rb_u2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int PADK = 8;       // bf16 elements of padding per LDS operand row (16 bytes: rows land on different banks)

struct Drop {                  // one Philox dropout stream: element i -> word (i & 3) of block (offset + i) >> 2
    float p;
    unsigned long long seed, offset;
};

// keep-scale factors of the four elements i .. i + 3 (i % 4 == 0) of stream d
__device__ __forceinline__ rb_f4 drop4(const Drop& d, long i) {
    if (d.p <= 0.f) return rb_f4{1.f, 1.f, 1.f, 1.f};
    const float scale = 1.f / (1.f - d.p);
    const uint32_t thr = (uint32_t)fminf(d.p * 4294967296.f, 4294967295.f);
    const uint64_t blk = (d.offset + (uint64_t)i) >> 2;
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u};
    philox4x32_10(ctr, (uint32_t)d.seed, (uint32_t)(d.seed >> 32));
    return rb_f4{ctr[0] >= thr ? scale : 0.f, ctr[1] >= thr ? scale : 0.f, ctr[2] >= thr ? scale : 0.f, ctr[3] >= thr ? scale : 0.f};
}

__device__ __forceinline__ rb_u2 pack4(rb_f4 v) {
    return rb_u2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
}
__device__ __forceinline__ rb_f4 unpack4(rb_u2 u) {
    return rb_f4{__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u), __uint_as_float(u[1] << 16), __uint_as_float(u[1] & 0xffff0000u)};
}

// ------------------------------------------------------------------------------------------------------ GEMM core
// acc[t] (16 x 16 per 16-row tile t) = sA[16 t .., :] . W[n0 .., :]^T over K, K split over the 4 waves, reduced into
// red[4][MT][16][17].  sA: bf16 [MT * 16][K + PADK] in LDS.
// The first eight weight fragments of a wave (k-steps wave, wave + 4, ...): issued at the very top of a kernel, before its
// prologue, so that the weight stream's first HBM round trip overlaps the prologue's instead of following it.
struct WPre {
    rb_bf16x8 b[8];
};
__device__ __forceinline__ void wfrag_batch(WPre& p, const bf16_t* wp, int steps, int s) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int su = s + 4 * u;
        p.b[u] = *reinterpret_cast<const rb_bf16x8*>(wp + 32 * min(su, steps - 1));
        if (su >= steps) p.b[u] = __builtin_bit_cast(rb_bf16x8, rb_u4{0u, 0u, 0u, 0u});      // k-steps past the end multiply by zero
    }
}
__device__ __forceinline__ const bf16_t* wfrag_ptr(const bf16_t* __restrict__ w, long ldw, int n0, int N) {
    const int lane = threadIdx.x & 63;
    return w + (long)min(n0 + (lane & 15), N - 1) * ldw + 8 * (lane >> 4);      // ragged last column group: re-read row N-1, never stored
}
template <int MT>
__device__ __forceinline__ void skinny_core(const bf16_t* sA, int K, const bf16_t* wp, WPre& pre, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int pitch = K + PADK;
    rb_f4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = rb_f4{0.f, 0.f, 0.f, 0.f};
    const int steps = K / 32;
    for (int s = wave; s < steps; s += 32) {
        if (s != wave) wfrag_batch(pre, wp, steps, s);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int su = min(s + 4 * u, steps - 1);
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const rb_bf16x8 a = *reinterpret_cast<const rb_bf16x8*>(sA + (16 * t + col) * pitch + 32 * su + 8 * kq);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pre.b[u], acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[((wave * MT + t) * 16 + kq * 4 + r) * 17 + col] = acc[t][r];   // C/D map: row = 4 kq + r, col
}
template <int MT>
__device__ __forceinline__ rb_f4 red_quad(const float* red, int t, int rr, int c4) {
    rb_f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] += red[((w * MT + t) * 16 + rr) * 17 + c4 + e];
    return v;
}

// ------------------------------------------------------------------------------------------------------ forward
struct FwdArgs {
    const void* a;            // operand rows [B, K]: f32 when a_f32 (LayerNorm input or plain cast), else bf16
    int a_f32, ln;
    const float *gamma, *beta;
    float eps;
    float* stats;             // ln: mean [B], rstd [B] written by workgroup 0 (saved for the backward)
    const bf16_t* w;          // [N, K]
    const float* bias;
    void* out;                // [B, N]
    int out_f32;
    bf16_t* preact;           // optional: bf16 copy of the result before the activation
    const float* res;         // optional f32 [B, N] residual: out = res + dropout(act(...))
    int act;
    Drop drop;
    const unsigned long long* dev_base;     // optional per-step base offset kept on the device (graph replays)
    int B, N, K;
};

template <int MT, int QPL>      // QPL: 16-byte quads per lane and row in one pass of the prologue (256 QPL columns)
__global__ __launch_bounds__(256) void rna_fwd_kernel(FwdArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sA = reinterpret_cast<bf16_t*>(smem);
    if (g.dev_base) g.drop.offset += *g.dev_base & ~3ull;
    const int K = g.K, pitch = K + PADK;
    float* red = reinterpret_cast<float*>(smem + (size_t)MT * 16 * pitch * 2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16;
    const bf16_t* wp = wfrag_ptr(g.w, K, n0, g.N);
    WPre pre;
    wfrag_batch(pre, wp, K / 32, wave);
    // ---- operand image: (LayerNorm of) the B input rows as bf16; rows past B are zero.  A wave owns rows wave, wave + 4, ...;
    //      all loads of four rows (8 quads per lane and row: a 2048-column chunk) are issued before anything waits on them —
    //      a row-by-row loop is a chain of dependent L2 round trips (~1 us each) in front of a ~2 us GEMM
    const int nchunk = (K + 256 * QPL - 1) / (256 * QPL);       // LayerNorm rows fit one chunk (host check: K <= 2048 when ln)
    for (int i0 = 0; i0 < MT * 4; i0 += 4) {
        for (int c = 0; c < nchunk; c++) {
            if (g.a_f32) {
                rb_f4 v[4][QPL], gm[QPL], bt[QPL];
#pragma unroll
                for (int j = 0; j < QPL; j++) {
                    const int k = min(256 * QPL * c + 4 * (lane + 64 * j), K - 4);
                    if (g.ln) { gm[j] = *reinterpret_cast<const rb_f4*>(g.gamma + k); bt[j] = *reinterpret_cast<const rb_f4*>(g.beta + k); }
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        v[i][j] = *reinterpret_cast<const rb_f4*>(reinterpret_cast<const float*>(g.a) + (long)min(wave + 4 * (i0 + i), g.B - 1) * K + k);
                }
                float mean[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {1.f, 1.f, 1.f, 1.f};
                if (g.ln) {          // two passes over the registers: mean, then sum (x - mean)^2 (what mh_layernorm_fwd computes)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        float s1 = 0.f;
#pragma unroll
                        for (int j = 0; j < QPL; j++)
                            if (4 * (lane + 64 * j) < K) s1 += v[i][j][0] + v[i][j][1] + v[i][j][2] + v[i][j][3];
                        mean[i] = wave_sum(s1) / (float)K;
                        float s2 = 0.f;
#pragma unroll
                        for (int j = 0; j < QPL; j++)
                            if (4 * (lane + 64 * j) < K) {
                                const rb_f4 d = v[i][j] - mean[i];
                                s2 += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
                            }
                        rstd[i] = rsqrtf(wave_sum(s2) / (float)K + g.eps);
                        const int m = wave + 4 * (i0 + i);
                        if (blockIdx.x == 0 && lane == 0 && g.stats && m < g.B) { g.stats[m] = mean[i]; g.stats[g.B + m] = rstd[i]; }
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < QPL; j++) {
                        const int m = wave + 4 * (i0 + i), k = 256 * QPL * c + 4 * (lane + 64 * j);
                        if (k >= K) continue;
                        rb_f4 o = v[i][j];
                        if (g.ln) o = (o - mean[i]) * rstd[i] * gm[j] + bt[j];
                        if (m >= g.B) o = rb_f4{0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<rb_u2*>(sA + m * pitch + k) = pack4(o);
                    }
            } else {
                rb_u2 v[4][QPL];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < QPL; j++) {
                        const int m = min(wave + 4 * (i0 + i), g.B - 1), k = min(256 * QPL * c + 4 * (lane + 64 * j), K - 4);
                        v[i][j] = *reinterpret_cast<const rb_u2*>(reinterpret_cast<const bf16_t*>(g.a) + (long)m * K + k);
                    }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < QPL; j++) {
                        const int m = wave + 4 * (i0 + i), k = 256 * QPL * c + 4 * (lane + 64 * j);
                        if (k >= K) continue;
                        *reinterpret_cast<rb_u2*>(sA + m * pitch + k) = m < g.B ? v[i][j] : rb_u2{0u, 0u};
                    }
            }
        }
    }
    __syncthreads();
    skinny_core<MT>(sA, K, wp, pre, red);
    __syncthreads();
    // ---- epilogue: one quad of columns per thread
    for (int i = threadIdx.x; i < MT * 64; i += 256) {
        const int t = i >> 6, rr = (i >> 2) & 15, c4 = 4 * (i & 3);
        const int m = 16 * t + rr, n = n0 + c4;
        if (m >= g.B || n >= g.N) continue;
        rb_f4 v = red_quad<MT>(red, t, rr, c4);
        if (g.bias) v += *reinterpret_cast<const rb_f4*>(g.bias + n);
        const long idx = (long)m * g.N + n;
        if (g.preact) *reinterpret_cast<rb_u2*>(g.preact + idx) = pack4(v);
        if (g.act == MH_ACT_GELU) v = rb_f4{gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])};
        else if (g.act == MH_ACT_RELU) v = rb_f4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        v *= drop4(g.drop, idx);
        if (g.res) v += *reinterpret_cast<const rb_f4*>(g.res + idx);
        if (g.out_f32) *reinterpret_cast<rb_f4*>(reinterpret_cast<float*>(g.out) + idx) = v;
        else *reinterpret_cast<rb_u2*>(reinterpret_cast<bf16_t*>(g.out) + idx) = pack4(v);
    }
}

size_t fwd_lds(int MT, int K) { return (size_t)MT * 16 * (K + PADK) * 2 + (size_t)4 * MT * 16 * 17 * 4; }

int launch_fwd(const FwdArgs& a, hipStream_t s) {
    const int MT = a.B <= 16 ? 1 : 2;
    const size_t lds = fwd_lds(MT, a.K);
    dim3 grid(mh_cdiv(a.N, 16));
    static const bool attr = [] {       // operand images above 64 KiB (K = 2048 at B = 16) need the opt-in
#define RNA_A(MT_, Q_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rna_fwd_kernel<MT_, Q_>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024)
        RNA_A(1, 2); RNA_A(1, 4); RNA_A(1, 8); RNA_A(2, 2); RNA_A(2, 4); RNA_A(2, 8);
#undef RNA_A
        return true;
    }();
    (void)attr;
    if (lds > 158 * 1024) { mh_set_error("rna block: operand image of %zu bytes exceeds the 160 KiB of LDS", lds); return MH_EINVAL; }
    // the prologue handles 256 QPL columns per pass: pick the smallest instance that covers K in one pass (compact code,
    // no predicated-off loads), 8 (2048 columns per pass) beyond
#define RNA_F(MT_, Q_) hipLaunchKernelGGL((rna_fwd_kernel<MT_, Q_>), grid, dim3(256), lds, s, a)
    const int q = a.K <= 512 ? 2 : (a.K <= 1024 ? 4 : 8);
    if (MT == 1) { if (q == 2) RNA_F(1, 2); else if (q == 4) RNA_F(1, 4); else RNA_F(1, 8); }
    else { if (q == 2) RNA_F(2, 2); else if (q == 4) RNA_F(2, 4); else RNA_F(2, 8); }
#undef RNA_F
    return 0;
}

// ------------------------------------------------------------------------------------------------------ backward
// One Linear y = x W^T + b of the block, both gradients in one grid.  The gradient matrix G [B, M] (M = the Linear's output
// width) is rebuilt by every workgroup from its source:
//   src 0: G = g_in (bf16)                                   src 1: G = r * dropmask                 (r f32 [B, M])
//   src 2: G = (r + LayerNorm'(dh; xs, stats, gamma)) * dropmask — the pre-norm residual join; workgroup 0 also writes the
//          joined f32 gradient to r_out and accumulates the LayerNorm's gamma / beta gradients
// dgrad role (blockIdx < n_dgrad): dx[:, k0 .. k0 + 16) = G . W over M, via W^T [Kin, M]; epilogue 0: f32, 1: * GELU'(u) *
//          dropmask2 -> bf16, 2: bf16
// wgrad role: dW[n0 .. n0 + 64, k0 .. k0 + 256) += G^T X, db += column sums of G; X = x_in (bf16) or LayerNorm(xl) recomputed
struct BwdArgs {
    int src;
    const bf16_t* g_in;
    const float *r, *dh, *xs, *stats, *gamma;
    float *r_out, *dgamma, *dbeta;
    Drop drop;
    int B, M;
    const bf16_t* wt;         // [Kin, M]
    int Kin, epi;
    void* dx;
    const bf16_t* u;
    Drop drop2;
    int xsrc;
    const bf16_t* x_in;
    const float *xl, *stats_l, *gamma_l, *beta_l;
    float *dw, *db;
    const unsigned long long* dev_base;
    int n_dgrad, n_wgrad;     // + one more workgroup when src == 2: the LayerNorm's gamma / beta gradients
};

// four consecutive elements (n % 4 == 0) of row m of G; c12: LDS [2][32] row constants of the LayerNorm backward (src 2)
__device__ __forceinline__ rb_f4 gval4(const BwdArgs& g, const float* c12, int m, int n) {
    const long idx = (long)m * g.M + n;
    if (g.src == 0) return unpack4(*reinterpret_cast<const rb_u2*>(g.g_in + idx));
    rb_f4 v = *reinterpret_cast<const rb_f4*>(g.r + idx);
    if (g.src == 2) {
        const float mean = g.stats[m], rstd = g.stats[g.B + m];
        const rb_f4 xh = (*reinterpret_cast<const rb_f4*>(g.xs + idx) - mean) * rstd;
        const rb_f4 a = *reinterpret_cast<const rb_f4*>(g.dh + idx) * *reinterpret_cast<const rb_f4*>(g.gamma + n);
        v += rstd * (a - c12[m] - xh * c12[32 + m]);
    }
    return v;
}

template <int MT, int QPL>
__global__ __launch_bounds__(256) void rna_bwd_kernel(BwdArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float c12[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int M = g.M;
    if (g.dev_base) {
        const unsigned long long base = *g.dev_base & ~3ull;
        g.drop.offset += base;
        g.drop2.offset += base;
    }
    const int role = (int)blockIdx.x < g.n_dgrad ? 0 : ((int)blockIdx.x < g.n_dgrad + g.n_wgrad ? 1 : 2);
    if (role == 2) {         // src 2 only: dgamma[n] += sum_m dh xhat, dbeta[n] += sum_m dh — columns over threads, 16 rows in flight
        for (int n = threadIdx.x; n < M; n += 256) {
            float dg = 0.f, dbt = 0.f;
            for (int m0 = 0; m0 < g.B; m0 += 16) {
                float xv[16], dv[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const long idx = (long)min(m0 + i, g.B - 1) * M + n;
                    xv[i] = g.xs[idx];
                    dv[i] = g.dh[idx];
                }
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (m0 + i < g.B) {
                        dg += dv[i] * (xv[i] - g.stats[m0 + i]) * g.stats[g.B + m0 + i];
                        dbt += dv[i];
                    }
            }
            g.dgamma[n] += dg;
            g.dbeta[n] += dbt;
        }
        return;
    }
    const int k0 = blockIdx.x * 16;
    const bf16_t* wp = nullptr;
    WPre pre;
    if (role == 0) {         // the weight stream's first round trip overlaps the prologue's
        wp = wfrag_ptr(g.wt, M, k0, g.Kin);
        wfrag_batch(pre, wp, M / 32, wave);
    }
    if (g.src == 2) {        // row constants of LayerNorm': c1 = mean(dh gamma), c2 = mean(dh gamma xhat); M <= 2048 (host check)
        for (int i0 = 0; i0 < MT * 4; i0 += 4) {
            rb_f4 xv[4][QPL], dv[4][QPL], gm[QPL];
#pragma unroll
            for (int j = 0; j < QPL; j++) {
                const int n = min(4 * (lane + 64 * j), M - 4);
                gm[j] = *reinterpret_cast<const rb_f4*>(g.gamma + n);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const long idx = (long)min(wave + 4 * (i0 + i), g.B - 1) * M + n;
                    xv[i][j] = *reinterpret_cast<const rb_f4*>(g.xs + idx);
                    dv[i][j] = *reinterpret_cast<const rb_f4*>(g.dh + idx);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = min(wave + 4 * (i0 + i), g.B - 1);
                const float mean = g.stats[m], rstd = g.stats[g.B + m];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < QPL; j++)
                    if (4 * (lane + 64 * j) < M) {
                        const rb_f4 xh = (xv[i][j] - mean) * rstd, a = dv[i][j] * gm[j];
                        s1 += a[0] + a[1] + a[2] + a[3];
                        s2 += a[0] * xh[0] + a[1] * xh[1] + a[2] * xh[2] + a[3] * xh[3];
                    }
                s1 = wave_sum(s1) / (float)M;
                s2 = wave_sum(s2) / (float)M;
                if (lane == 0 && wave + 4 * (i0 + i) < g.B) { c12[m] = s1; c12[32 + m] = s2; }
            }
        }
        __syncthreads();
    }
    if (role == 0) {
        // ------------------------------------------------ data gradient: G (bf16 image) . W^T rows
        bf16_t* sG = reinterpret_cast<bf16_t*>(smem);
        const int pitch = M + PADK;
        float* red = reinterpret_cast<float*>(smem + (size_t)MT * 16 * pitch * 2);
        // four rows x 8 quads per lane in flight (a 2048-column chunk), as in the forward prologue
        const int nchunk = (M + 256 * QPL - 1) / (256 * QPL);
        for (int i0 = 0; i0 < MT * 4; i0 += 4)
            for (int c = 0; c < nchunk; c++) {
                rb_f4 v[4][QPL];
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < QPL; j++) {
                        const int m = min(wave + 4 * (i0 + i), g.B - 1), n = min(256 * QPL * c + 4 * (lane + 64 * j), M - 4);
                        v[i][j] = gval4(g, c12, m, n);
                    }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < QPL; j++) {
                        const int m = wave + 4 * (i0 + i), n = 256 * QPL * c + 4 * (lane + 64 * j);
                        if (n >= M) continue;
                        rb_f4 o = {0.f, 0.f, 0.f, 0.f};
                        if (m < g.B) o = v[i][j] * drop4(g.drop, (long)m * M + n);
                        *reinterpret_cast<rb_u2*>(sG + m * pitch + n) = pack4(o);
                        // src 2: the joined f32 gradient r + LayerNorm'(dh) leaves through the workgroups that own its columns
                        if (g.src == 2 && g.r_out && m < g.B && (n >> 4) == (int)blockIdx.x) *reinterpret_cast<rb_f4*>(g.r_out + (long)m * M + n) = v[i][j];
                    }
            }
        __syncthreads();
        skinny_core<MT>(sG, M, wp, pre, red);
        __syncthreads();
        for (int i = threadIdx.x; i < MT * 64; i += 256) {
            const int t = i >> 6, rr = (i >> 2) & 15, c4 = 4 * (i & 3);
            const int m = 16 * t + rr, k = k0 + c4;
            if (m >= g.B || k >= g.Kin) continue;
            rb_f4 v = red_quad<MT>(red, t, rr, c4);
            const long idx = (long)m * g.Kin + k;
            if (g.epi == 1) {
                const rb_f4 uu = unpack4(*reinterpret_cast<const rb_u2*>(g.u + idx));
                v *= drop4(g.drop2, idx) * rb_f4{gelu_grad_f(uu[0]), gelu_grad_f(uu[1]), gelu_grad_f(uu[2]), gelu_grad_f(uu[3])};
            }
            if (g.epi == 0) *reinterpret_cast<rb_f4*>(reinterpret_cast<float*>(g.dx) + idx) = v;
            else *reinterpret_cast<rb_u2*>(reinterpret_cast<bf16_t*>(g.dx) + idx) = pack4(v);
        }
        return;
    }
    // ---------------------------------------------------- weight gradient tile: 64 (n) x 256 (k), rank-B outer products
    float (*sdy)[64] = reinterpret_cast<float (*)[64]>(smem);
    float (*sx)[256] = reinterpret_cast<float (*)[256]>(smem + 32 * 64 * 4);
    const int tiles_n = (M + 63) / 64;
    const int t = blockIdx.x - g.n_dgrad;
    const int n0 = (t % tiles_n) * 64, kt0 = (t / tiles_n) * 256;
    const int Bn = g.B;
    {   // G tile [B][64] and X tile [B][256] as f32: all loads issued before the first use
        rb_f4 gv[MT], xv[MT * 4];
#pragma unroll
        for (int u = 0; u < MT; u++) {              // MT * 16 rows x 16 quads = MT * 256 items
            const int i = threadIdx.x + 256 * u, m = min(i >> 4, Bn - 1), c = 4 * (i & 15);
            gv[u] = gval4(g, c12, m, min(n0 + c, M - 4));
        }
#pragma unroll
        for (int u = 0; u < MT * 4; u++) {
            const int i = threadIdx.x + 256 * u, m = min(i >> 6, Bn - 1), k = min(kt0 + 4 * (i & 63), g.Kin - 4);
            if (g.xsrc == 0) xv[u] = unpack4(*reinterpret_cast<const rb_u2*>(g.x_in + (long)m * g.Kin + k));
            else xv[u] = *reinterpret_cast<const rb_f4*>(g.xl + (long)m * g.Kin + k);
        }
#pragma unroll
        for (int u = 0; u < MT; u++) {
            const int i = threadIdx.x + 256 * u, m = i >> 4, c = 4 * (i & 15);
            rb_f4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < Bn && n0 + c < M) {      // M % 4 == 0: a quad is inside or outside as a whole
                v = gv[u] * drop4(g.drop, (long)m * M + n0 + c);
                v = unpack4(pack4(v));                       // the data gradient consumes G rounded to bf16: same operand here
            }
            *reinterpret_cast<rb_f4*>(&sdy[m][c]) = v;
        }
#pragma unroll
        for (int u = 0; u < MT * 4; u++) {
            const int i = threadIdx.x + 256 * u, m = i >> 6, c = 4 * (i & 63), k = kt0 + c;
            rb_f4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < Bn && k < g.Kin) {
                v = xv[u];
                if (g.xsrc != 0) {
                    const float mean = g.stats_l[m], rstd = g.stats_l[Bn + m];
                    v = (v - mean) * rstd * *reinterpret_cast<const rb_f4*>(g.gamma_l + k) + *reinterpret_cast<const rb_f4*>(g.beta_l + k);
                    v = unpack4(pack4(v));                   // what the forward GEMM multiplied by
                }
            }
            *reinterpret_cast<rb_f4*>(&sx[m][c]) = v;
        }
    }
    __syncthreads();
    if (g.db && kt0 == 0 && threadIdx.x < 64 && n0 + (int)threadIdx.x < M) {
        float sacc = 0.f;
        for (int m = 0; m < Bn; m++) sacc += sdy[m][threadIdx.x];
        g.db[n0 + threadIdx.x] += sacc;
    }
    const int tn = threadIdx.x >> 5, tk = threadIdx.x & 31;
    float acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = 0.f;
    for (int m = 0; m < Bn; m++) {
        const rb_f4 a0 = *reinterpret_cast<const rb_f4*>(&sdy[m][8 * tn]), a1 = *reinterpret_cast<const rb_f4*>(&sdy[m][8 * tn + 4]);
        const rb_f4 b0 = *reinterpret_cast<const rb_f4*>(&sx[m][8 * tk]), b1 = *reinterpret_cast<const rb_f4*>(&sx[m][8 * tk + 4]);
        const float a[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const float b[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] += a[i] * b[j];
    }
    if (kt0 + 8 * tk >= g.Kin) return;         // Kin % 8 == 0: an 8-wide strip is inside or outside as a whole
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int n = n0 + 8 * tn + i;
        if (n >= M) continue;
        float* dst = g.dw + (long)n * g.Kin + kt0 + 8 * tk;
        rb_f4 o0 = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]}, o1 = {acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
        o0 += *reinterpret_cast<const rb_f4*>(dst);
        o1 += *reinterpret_cast<const rb_f4*>(dst + 4);
        *reinterpret_cast<rb_f4*>(dst) = o0;
        *reinterpret_cast<rb_f4*>(dst + 4) = o1;
    }
}

int launch_bwd(BwdArgs a, hipStream_t s) {
    const int MT = a.B <= 16 ? 1 : 2;
    a.n_dgrad = a.dx ? mh_cdiv(a.Kin, 16) : 0;
    a.n_wgrad = a.dw ? mh_cdiv(a.M, 64) * mh_cdiv(a.Kin, 256) : 0;
    const int n_wgrad = a.n_wgrad + (a.src == 2 ? 1 : 0);
    const size_t lds_d = (size_t)MT * 16 * (a.M + PADK) * 2 + (size_t)4 * MT * 16 * 17 * 4, lds_w = 32 * 64 * 4 + 32 * 256 * 4;
    const size_t lds = lds_d > lds_w ? lds_d : lds_w;
    dim3 grid(a.n_dgrad + n_wgrad);
    if (grid.x == 0) return 0;
    static const bool attr = [] {
#define RNA_A(MT_, Q_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rna_bwd_kernel<MT_, Q_>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024)
        RNA_A(1, 2); RNA_A(1, 4); RNA_A(1, 8); RNA_A(2, 2); RNA_A(2, 4); RNA_A(2, 8);
#undef RNA_A
        return true;
    }();
    (void)attr;
    if (lds > 158 * 1024) { mh_set_error("rna block: operand image of %zu bytes exceeds the 160 KiB of LDS", lds); return MH_EINVAL; }
#define RNA_B(MT_, Q_) hipLaunchKernelGGL((rna_bwd_kernel<MT_, Q_>), grid, dim3(256), lds, s, a)
    const int q = a.M <= 512 ? 2 : (a.M <= 1024 ? 4 : 8);
    if (MT == 1) { if (q == 2) RNA_B(1, 2); else if (q == 4) RNA_B(1, 4); else RNA_B(1, 8); }
    else { if (q == 2) RNA_B(2, 2); else if (q == 4) RNA_B(2, 4); else RNA_B(2, 8); }
#undef RNA_B
    return 0;
}

// dx = r + LayerNorm'(dh; x, stats, gamma); dgamma += sum_m dh xhat; dbeta += sum_m dh.  One workgroup per 64-column slab;
// every workgroup recomputes the 2 B row constants from the whole [B, D] operands (all loads of four rows in flight).
template <int QPL>
__global__ __launch_bounds__(256) void rna_ln_bwd_res_kernel(const float* __restrict__ r, const float* __restrict__ dh, const float* __restrict__ x,
                                                             const float* __restrict__ stats, const float* __restrict__ gamma, float* __restrict__ dx,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int D) {
    __shared__ float c12[64];
    __shared__ float part[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i0 = 0; i0 < 8; i0 += 4) {
        if (wave + 4 * i0 >= B) break;
        rb_f4 xv[4][QPL], dv[4][QPL], gm[QPL];
#pragma unroll
        for (int j = 0; j < QPL; j++) {
            const int n = min(4 * (lane + 64 * j), D - 4);
            gm[j] = *reinterpret_cast<const rb_f4*>(gamma + n);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const long idx = (long)min(wave + 4 * (i0 + i), B - 1) * D + n;
                xv[i][j] = *reinterpret_cast<const rb_f4*>(x + idx);
                dv[i][j] = *reinterpret_cast<const rb_f4*>(dh + idx);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = min(wave + 4 * (i0 + i), B - 1);
            const float mean = stats[m], rstd = stats[B + m];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < QPL; j++)
                if (4 * (lane + 64 * j) < D) {
                    const rb_f4 xh = (xv[i][j] - mean) * rstd, a = dv[i][j] * gm[j];
                    s1 += a[0] + a[1] + a[2] + a[3];
                    s2 += a[0] * xh[0] + a[1] * xh[1] + a[2] * xh[2] + a[3] * xh[3];
                }
            s1 = wave_sum(s1) / (float)D;
            s2 = wave_sum(s2) / (float)D;
            if (lane == 0 && wave + 4 * (i0 + i) < B) { c12[m] = s1; c12[32 + m] = s2; }
        }
    }
    __syncthreads();
    // slab: column n = 64 blockIdx + lane, rows wave, wave + 4, ...: partial column sums per wave, folded through LDS
    const int n = 64 * blockIdx.x + lane;
    float dg = 0.f, dbt = 0.f;
    if (n < D) {
        const float gmn = gamma[n];
#pragma unroll 8
        for (int m = wave; m < B; m += 4) {
            const long idx = (long)m * D + n;
            const float rstd = stats[B + m], xh = (x[idx] - stats[m]) * rstd, d = dh[idx];
            dg += d * xh;
            dbt += d;
            dx[idx] = (r ? r[idx] : 0.f) + rstd * (d * gmn - c12[m] - xh * c12[32 + m]);
        }
    }
    part[0][wave][lane] = dg;
    part[1][wave][lane] = dbt;
    __syncthreads();
    if (wave == 0 && n < D) {
        dgamma[n] += part[0][0][lane] + part[0][1][lane] + part[0][2][lane] + part[0][3][lane];
        dbeta[n] += part[1][0][lane] + part[1][1][lane] + part[1][2][lane] + part[1][3][lane];
    }
}

inline long q4(long n) { return (n + 3) / 4 * 4; }

}  // namespace

extern "C" int64_t mh_rna_block_workspace_bytes(int B, int D, int Hh) {
    // dg bf16 [B, Hh] | dh2 f32 [B, D] | dx1 f32 [B, D] | do bf16 [B, D] | dqkv bf16 [B, 3D] | dh1 f32 [B, D]; each 256-byte aligned
    auto al = [](long n) { return (n + 255) / 256 * 256; };
    return al(2L * B * Hh) + 3 * al(4L * B * D) + al(2L * B * D) + al(6L * B * D);
}

static int rna_check(const mh_rna_block* b, const char* who) {
    MH_REQUIRE(b, "%s: null descriptor", who);
    MH_REQUIRE(b->B >= 1 && b->B <= 32, "%s: B=%d (needs 1..32 rows)", who, b->B);
    MH_REQUIRE(b->D % 32 == 0 && b->Hh % 32 == 0 && b->D > 0 && b->Hh > 0, "%s: D=%d, Hh=%d must be multiples of 32", who, b->D, b->Hh);
    MH_REQUIRE(b->H >= 1 && b->H <= 64 && b->D % b->H == 0 && b->D <= 4096, "%s: H=%d does not divide D=%d", who, b->H, b->D);
    MH_REQUIRE(b->Hh <= 4096 && b->D <= 2048, "%s: D=%d, Hh=%d exceed the batched prologues (D <= 2048, Hh <= 4096)", who, b->D, b->Hh);
    MH_REQUIRE(b->p_drop >= 0.f && b->p_drop < 1.f && (b->offset & 3) == 0, "%s: bad dropout arguments", who);
    return MH_OK;
}

extern "C" int mh_rna_block_fwd(const mh_rna_block* b, mh_stream s) {
    if (int rc = rna_check(b, "mh_rna_block_fwd")) return rc;
    hipStream_t st = (hipStream_t)s;
    const int B = b->B, D = b->D, Hh = b->Hh;
    const unsigned long long base = b->offset;      // a device-side per-step base (dev_base) is added by the kernels themselves
    const Drop none{0.f, 0ull, 0ull};
    // dropout streams in the order of the un-fused path: proj output, fc1 activation, fc2 output
    Drop d1{b->p_drop, b->seed, base}, d2{b->p_drop, b->seed, base + q4((long)B * D)}, d3{b->p_drop, b->seed, base + q4((long)B * D) + q4((long)B * Hh)};
    FwdArgs a{};
    a.B = B;
    a.dev_base = (const unsigned long long*)b->dev_base;
    // [LN1 -> qkv]
    a.a = b->x; a.a_f32 = 1; a.ln = 1; a.gamma = b->g1; a.beta = b->be1; a.eps = b->eps; a.stats = b->stats;
    a.w = (const bf16_t*)b->w_qkv; a.bias = b->b_qkv; a.out = b->qkv; a.out_f32 = 0; a.preact = nullptr; a.res = nullptr; a.act = MH_ACT_NONE;
    a.drop = none; a.N = 3 * D; a.K = D;
    if (int rc = launch_fwd(a, st)) return rc;
    if (int rc = mh_headattn_fwd(b->qkv, b->o, b->attn, B, b->H, D / b->H, MH_BF16, s)) return rc;
    // [proj + dropout + residual]
    a.a = b->o; a.a_f32 = 0; a.ln = 0; a.stats = nullptr; a.w = (const bf16_t*)b->w_proj; a.bias = b->b_proj; a.out = b->x1; a.out_f32 = 1;
    a.res = b->x; a.drop = d1; a.N = D; a.K = D;
    if (int rc = launch_fwd(a, st)) return rc;
    // [LN2 -> fc1 -> GELU -> dropout]
    a.a = b->x1; a.a_f32 = 1; a.ln = 1; a.gamma = b->g2; a.beta = b->be2; a.stats = b->stats + 2 * B;
    a.w = (const bf16_t*)b->w_fc1; a.bias = b->b_fc1; a.out = b->f; a.out_f32 = 0; a.preact = (bf16_t*)b->u; a.res = nullptr; a.act = MH_ACT_GELU;
    a.drop = d2; a.N = Hh; a.K = D;
    if (int rc = launch_fwd(a, st)) return rc;
    // [fc2 + dropout + residual]
    a.a = b->f; a.a_f32 = 0; a.ln = 0; a.stats = nullptr; a.w = (const bf16_t*)b->w_fc2; a.bias = b->b_fc2; a.out = b->y; a.out_f32 = 1;
    a.preact = nullptr; a.res = b->x1; a.act = MH_ACT_NONE; a.drop = d3; a.N = D; a.K = Hh;
    if (int rc = launch_fwd(a, st)) return rc;
    MH_LAUNCH_CHECK("mh_rna_block_fwd");
    return MH_OK;
}

extern "C" int mh_rna_block_bwd(const mh_rna_block* b, mh_stream s) {
    if (int rc = rna_check(b, "mh_rna_block_bwd")) return rc;
    MH_REQUIRE(b->scratch && b->dy && b->dx, "mh_rna_block_bwd: dy, dx and scratch are required");
    hipStream_t st = (hipStream_t)s;
    const int B = b->B, D = b->D, Hh = b->Hh;
    auto al = [](long n) { return (n + 255) / 256 * 256; };
    char* ws = (char*)b->scratch;
    bf16_t* dg = (bf16_t*)ws; ws += al(2L * B * Hh);
    float* dh2 = (float*)ws; ws += al(4L * B * D);
    float* dx1 = (float*)ws; ws += al(4L * B * D);
    bf16_t* dO = (bf16_t*)ws; ws += al(2L * B * D);
    bf16_t* dqkv = (bf16_t*)ws; ws += al(6L * B * D);
    float* dh1 = (float*)ws;
    const Drop none{0.f, 0ull, 0ull};
    Drop d1{b->p_drop, b->seed, b->offset}, d2{b->p_drop, b->seed, b->offset + q4((long)B * D)},
        d3{b->p_drop, b->seed, b->offset + q4((long)B * D) + q4((long)B * Hh)};
    BwdArgs a{};
    a.dev_base = (const unsigned long long*)b->dev_base;
    a.B = B;
    // fc2: G = dy * mask3 [B, D];  dg = (G . W2) * mask2 * GELU'(u);  dW2 += G^T f
    a.src = 1; a.r = b->dy; a.drop = d3; a.M = D; a.wt = (const bf16_t*)b->wt_fc2; a.Kin = Hh; a.epi = 1; a.dx = dg; a.u = (const bf16_t*)b->u; a.drop2 = d2;
    a.xsrc = 0; a.x_in = (const bf16_t*)b->f; a.dw = b->dw_fc2; a.db = b->db_fc2;
    if (int rc = launch_bwd(a, st)) return rc;
    // fc1: G = dg [B, Hh];  dh2 = G . W1 (f32);  dW1 += G^T LN2(x1)
    a = BwdArgs{};
    a.dev_base = (const unsigned long long*)b->dev_base;
    a.B = B; a.src = 0; a.g_in = dg; a.drop = none; a.M = Hh; a.wt = (const bf16_t*)b->wt_fc1; a.Kin = D; a.epi = 0; a.dx = dh2; a.drop2 = none;
    a.xsrc = 1; a.xl = b->x1; a.stats_l = b->stats + 2 * B; a.gamma_l = b->g2; a.beta_l = b->be2; a.dw = b->dw_fc1; a.db = b->db_fc1;
    if (int rc = launch_bwd(a, st)) return rc;
    // proj: dx1 = dy + LN2'(dh2);  G = dx1 * mask1;  do = G . Wproj (bf16);  dWproj += G^T o
    a = BwdArgs{};
    a.dev_base = (const unsigned long long*)b->dev_base;
    a.B = B; a.src = 2; a.r = b->dy; a.dh = dh2; a.xs = b->x1; a.stats = b->stats + 2 * B; a.gamma = b->g2; a.r_out = dx1; a.dgamma = b->dg2; a.dbeta = b->dbe2;
    a.drop = d1; a.M = D; a.wt = (const bf16_t*)b->wt_proj; a.Kin = D; a.epi = 2; a.dx = dO; a.drop2 = none;
    a.xsrc = 0; a.x_in = (const bf16_t*)b->o; a.dw = b->dw_proj; a.db = b->db_proj;
    if (int rc = launch_bwd(a, st)) return rc;
    if (int rc = mh_headattn_bwd(b->qkv, b->attn, dO, dqkv, B, b->H, D / b->H, MH_BF16, s)) return rc;
    // qkv: G = dqkv [B, 3D];  dh1 = G . Wqkv (f32);  dWqkv += G^T LN1(x)
    a = BwdArgs{};
    a.dev_base = (const unsigned long long*)b->dev_base;
    a.B = B; a.src = 0; a.g_in = dqkv; a.drop = none; a.M = 3 * D; a.wt = (const bf16_t*)b->wt_qkv; a.Kin = D; a.epi = 0; a.dx = dh1; a.drop2 = none;
    a.xsrc = 1; a.xl = b->x; a.stats_l = b->stats; a.gamma_l = b->g1; a.beta_l = b->be1; a.dw = b->dw_qkv; a.db = b->db_qkv;
    if (int rc = launch_bwd(a, st)) return rc;
    // dx = dx1 + LN1'(dh1)
#define RNA_L(Q_) hipLaunchKernelGGL((rna_ln_bwd_res_kernel<Q_>), dim3(mh_cdiv(D, 64)), dim3(256), 0, st, (const float*)dx1, (const float*)dh1, b->x, (const float*)b->stats, b->g1, b->dx, b->dg1, b->dbe1, B, D)
    if (D <= 512) RNA_L(2); else if (D <= 1024) RNA_L(4); else RNA_L(8);
#undef RNA_L
    MH_LAUNCH_CHECK("mh_rna_block_bwd");
    return MH_OK;
}
