// Row-wise kernels: LayerNorm fwd/bwd, softmax fwd/bwd, L2-normalise fwd/bwd.
// One wave64 per row (shuffle reductions); long softmax rows use one 256-thread block per row.
#include "common.h"
#include <cstdlib>

#define LN_MAXPL 32  // values per lane kept in registers -> D <= 2048

template <typename TX, typename TY>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, TY* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            long rows, int rpb, int D, long x_bs, long y_bs, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long b = row / rpb, i = row % rpb;
    const TX* xr = x + b * x_bs + i * D;
    TY* yr = y + b * y_bs + i * D;
    float v[LN_MAXPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXPL; k++) {
        const int c = lane + 64 * k;
        v[k] = (c < D) ? ldf(xr + c) : 0.f;
        s += v[k];
    }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXPL; k++) {
        const int c = lane + 64 * k;
        const float d = (c < D) ? v[k] - mu : 0.f;
        q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int k = 0; k < LN_MAXPL; k++) {
        const int c = lane + 64 * k;
        if (c < D) stf(yr + c, (v[k] - mu) * rs * gamma[c] + beta[c]);
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat*mean(g*dy*xhat)); dgamma += dy*xhat; dbeta += dy
template <typename TX, typename TDY, typename TDX>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, TDX* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            long rows, int rpb, int D, long x_bs, long y_bs, int acc_dx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float pg[LN_MAXPL], pb[LN_MAXPL];
#pragma unroll
    for (int k = 0; k < LN_MAXPL; k++) { pg[k] = 0.f; pb[k] = 0.f; }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const long b = row / rpb, i = row % rpb;
        const TX* xr = x + b * x_bs + i * D;
        const TDY* dyr = dy + b * y_bs + i * D;
        TDX* dxr = dx + b * x_bs + i * D;
        const float mu = mean[row], rs = rstd[row];
        float xh[LN_MAXPL], gd[LN_MAXPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXPL; k++) {
            const int c = lane + 64 * k;
            if (c < D) {
                const float d = ldf(dyr + c);
                xh[k] = (ldf(xr + c) - mu) * rs;
                gd[k] = d * gamma[c];
                pg[k] += d * xh[k];
                pb[k] += d;
                s1 += gd[k];
                s2 += gd[k] * xh[k];
            } else { xh[k] = 0.f; gd[k] = 0.f; }
        }
        s1 = wave_sum(s1) / D;
        s2 = wave_sum(s2) / D;
#pragma unroll
        for (int k = 0; k < LN_MAXPL; k++) {
            const int c = lane + 64 * k;
            if (c < D) {
                float r = rs * (gd[k] - s1 - xh[k] * s2);
                if (acc_dx) r += ldf(dxr + c);
                stf(dxr + c, r);
            }
        }
    }
    // reduce the 4 waves' column partials through LDS, one atomic per column per block
    __shared__ float red[2][4][64];
#pragma unroll
    for (int k = 0; k < LN_MAXPL; k++) {
        if (64 * k >= D) break;
        __syncthreads();
        red[0][wave][lane] = pg[k];
        red[1][wave][lane] = pb[k];
        __syncthreads();
        if (wave == 0) {
            const int c = lane + 64 * k;
            if (c < D) {
                atomicAdd(dgamma + c, red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]);
                atomicAdd(dbeta + c, red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]);
            }
        }
    }
}

// 4-wide vector access: ld4 / st4 of common.h
typedef f4_t f4;

// Vectorised LayerNorm (D % 4 == 0): lane owns elements [256k + 4*lane, +4), k < LNV_CH (D <= 2048).
template <typename TX, typename TY, int LNV_CH>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const TX* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, TY* __restrict__ y,
                                                                float* __restrict__ mean, float* __restrict__ rstd,
                                                                long rows, int rpb, int D, long x_bs, long y_bs, float eps,
                                                                bf16_t* __restrict__ y2 = nullptr) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long b = row / rpb, i = row % rpb;
    const TX* xr = x + b * x_bs + i * D;
    TY* yr = y + b * y_bs + i * D;
    bf16_t* y2r = y2 ? y2 + b * y_bs + i * D : nullptr;      // mh_layernorm_fwd_dual: a bf16 copy beside the f32 output
    f4 v[LNV_CH];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LNV_CH; k++) {
        const int c = 256 * k + 4 * lane;
        if (c < D) { v[k] = ld4(xr + c); s += v[k][0] + v[k][1] + v[k][2] + v[k][3]; }
        else v[k] = (f4){0.f, 0.f, 0.f, 0.f};
    }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LNV_CH; k++) {
        const int c = 256 * k + 4 * lane;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; e++) { const float d = v[k][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int k = 0; k < LNV_CH; k++) {
        const int c = 256 * k + 4 * lane;
        if (c < D) {
            const f4 g = ld4(gamma + c), bt = ld4(beta + c);
            f4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (v[k][e] - mu) * rs * g[e] + bt[e];
            st4(yr + c, o);
            if (y2r) st4(y2r + c, o);
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// LayerNorm forward of a Nystrom layer (models/mirror.py:298 + the front zero padding of [3P] NystromAttention) that ALSO
// leaves the landmark means of its OUTPUT: xpm[b, g] (f32) = mean over the l consecutive padded positions g l .. g l + l - 1 of the
// (bf16-rounded) normalised rows, zero rows for the pad.  to_qkv is linear and bias-free, so the package's landmarks
// q_landmarks = reduce(q, '... (n l) d -> ... n d', 'sum') / l equal to_qkv(xpm)[:, :2D]: a [B m, D] x [D, 2D] product instead of
// a pass over the [B, n_p, 2D] q | k columns, and in the backward the landmark gradient reaches the rows through this
// LayerNorm's backward (mh_layernorm_bwd gadd) instead of a read-modify-write of dqkv.
// One wave per (batch, group): its l rows in chunks of LMU rows in flight; pad rows are written as zeros here.
#define LMU 2
template <int LNV_CH, int SPLIT>
__global__ __launch_bounds__(256) void layernorm_fwd_lm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                               float* __restrict__ mean, float* __restrict__ rstd,
                                                               float* __restrict__ xpm, bf16_t* __restrict__ xpm16, int groups, int m, int rows,
                                                               int D, long x_bs, int pad, int l, float eps, const float* __restrict__ rmask,
                                                               const float* __restrict__ lm_scale) {
    // lm_scale [batches, m] (may be null): the landmark row of group g leaves as (masked) sum / l * lm_scale[b, g] — l / (valid count) turns
    // it into the masked mean ([3P] `q_landmarks /= divisor`), so the caller needs no pass over the landmark rows
    // rmask [batches, pad + rows] (may be null; BASELINE config 4's key-padding mask, front-padded like the sequence): rows with a zero
    // entry leave as ZERO rows (to_qkv is bias-free: zero q / k / v, what the package's `t * mask[..., None]` makes of them) and add
    // nothing to their group's sum — the landmark rows are then masked SUMS / l, which the caller scales by l / (valid count)
    // SPLIT waves share a group (its rows in SPLIT consecutive runs, partial landmark sums met in LDS): with few long groups (config 4:
    // 2048 groups of 33 rows) one wave per group leaves 8 waves per CU walking rows one after the other (2.4 TB/s)
    __shared__ f4 part[SPLIT > 1 ? 4 : 1][LNV_CH][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = wave % SPLIT;
    const int grp = blockIdx.x * (4 / SPLIT) + wave / SPLIT;
    if (SPLIT == 1 && grp >= groups) return;          // SPLIT > 1: the host launches whole blocks only (groups % (4 / SPLIT) == 0)
    const int b = grp / m, g = grp - b * m;
    const long n_p = (long)pad + rows;
    f4 gm[LNV_CH], bt[LNV_CH], acc[LNV_CH];
#pragma unroll
    for (int k = 0; k < LNV_CH; k++) {
        const int c = 256 * k + 4 * lane;
        acc[k] = (f4){0.f, 0.f, 0.f, 0.f};
        gm[k] = c < D ? ld4(gamma + c) : acc[k];
        bt[k] = c < D ? ld4(beta + c) : acc[k];
    }
    // the rows of chunk c + 1 are requested before chunk c is reduced and stored: with one chunk in flight the kernel alternated
    // between a load phase and a compute / store phase (4.5 TB/s)
    const int per = (l + SPLIT - 1) / SPLIT;
    const int jbeg = g * l + sub * per;
    const int jend = min(jbeg + per, (g + 1) * l);
    auto load_chunk = [&](f4 (&v)[LMU][LNV_CH], float (&kp)[LMU], int j0) {
#pragma unroll
        for (int u = 0; u < LMU; u++) {
            const int j = j0 + u;
            const bool live = j < jend && j >= pad;
            const float* xr = x + b * x_bs + (long)(live ? j - pad : 0) * D;
            kp[u] = (rmask && live) ? rmask[b * n_p + j] : 1.f;        // asked for with the row, not when the row is normalised
#pragma unroll
            for (int k = 0; k < LNV_CH; k++) {
                const int c = 256 * k + 4 * lane;
                v[u][k] = (live && c < D) ? ld4(xr + c) : (f4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto do_chunk = [&](f4 (&v)[LMU][LNV_CH], float (&kp)[LMU], int j0) {
#pragma unroll
        for (int u = 0; u < LMU; u++) {
            const int j = j0 + u;
            if (j >= jend) break;                     // wave-uniform
            bf16_t* yr = y + (b * n_p + j) * D;
            if (j < pad) {                            // front padding: zero rows, no statistics
#pragma unroll
                for (int k = 0; k < LNV_CH; k++) {
                    const int c = 256 * k + 4 * lane;
                    if (c < D) st4(yr + c, (f4){0.f, 0.f, 0.f, 0.f});
                }
                continue;
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < LNV_CH; k++) s += v[u][k][0] + v[u][k][1] + v[u][k][2] + v[u][k][3];
            const float mu = wave_sum(s) / D;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < LNV_CH; k++) {
                const int c = 256 * k + 4 * lane;
                if (c < D) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { const float d = v[u][k][e] - mu; q += d * d; }
                }
            }
            const float rs = rsqrtf(wave_sum(q) / D + eps);
            const bool keep = kp[u] != 0.f;
#pragma unroll
            for (int k = 0; k < LNV_CH; k++) {
                const int c = 256 * k + 4 * lane;
                if (c < D) {
                    f4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = keep ? bf2f(f2bf((v[u][k][e] - mu) * rs * gm[k][e] + bt[k][e])) : 0.f;   // what the projection reads
                    st4(yr + c, o);
                    acc[k] += o;
                }
            }
            if (lane == 0) { const long row = (long)b * rows + (j - pad); mean[row] = mu; rstd[row] = rs; }
        }
    };
    f4 va[LMU][LNV_CH], vb[LMU][LNV_CH];
    float ka[LMU], kb[LMU];
    load_chunk(va, ka, jbeg);
    for (int j0 = jbeg; j0 < jend; j0 += 2 * LMU) {
        if (j0 + LMU < jend) load_chunk(vb, kb, j0 + LMU);
        do_chunk(va, ka, j0);
        if (j0 + LMU >= jend) break;
        if (j0 + 2 * LMU < jend) load_chunk(va, ka, j0 + 2 * LMU);
        do_chunk(vb, kb, j0 + LMU);
    }
    if (SPLIT > 1) {
        if (sub) {
#pragma unroll
            for (int k = 0; k < LNV_CH; k++) part[wave][k][lane] = acc[k];
        }
        __syncthreads();
        if (sub) return;
        for (int w = 1; w < SPLIT; w++) {
#pragma unroll
            for (int k = 0; k < LNV_CH; k++) acc[k] += part[wave + w][k][lane];
        }
    }
    const float inv = (lm_scale ? lm_scale[(long)b * m + g] : 1.f) / (float)l;
#pragma unroll
    for (int k = 0; k < LNV_CH; k++) {
        const int c = 256 * k + 4 * lane;
        if (c < D) {
            if (xpm) st4(xpm + ((long)b * m + g) * D + c, acc[k] * inv);
            if (xpm16) st4(xpm16 + ((long)b * m + g) * D + c, acc[k] * inv);      // what a bf16 MFMA operand load of xpm rounds to
        }
    }
}

// LayerNorm forward that ALSO leaves an e4m3 copy of its output (BASELINE config 5: the fp8 forward of to_qkv reads it; the
// bf16 copy is still needed by the weight gradient): delayed scaling exactly as mh_quant_fp8_delayed (ring of three per-site
// maxima rotated by the device-side step counter), so the projection's input needs no quantisation pass of its own.
// q has y's row addressing in bytes (q_bs bytes per batch).
template <int LNV_CH>
__global__ __launch_bounds__(256) void layernorm_fwd_q8_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                               float* __restrict__ mean, float* __restrict__ rstd,
                                                               long rows, int rpb, int D, long x_bs, long y_bs, float eps,
                                                               unsigned char* __restrict__ q, unsigned* __restrict__ ring,
                                                               const float* __restrict__ tick, float margin, float* __restrict__ scale) {
    constexpr float F8M = 448.f;
    __shared__ float red[4];
    const int lane = threadIdx.x & 63;
    const int t = (int)tick[0];
    const int cur = t % 3, prev = (t + 2) % 3, nxt = (t + 1) % 3;
    const float amax = __uint_as_float(ring[prev]) * margin;
    const float mul = amax > 0.f ? F8M / amax : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scale[0] = amax > 0.f ? amax / F8M : 1.f;
        ring[nxt] = 0u;
    }
    float m = 0.f;
    // a wave walks rows (grid-stride) and keeps the running maximum: ONE atomic per workgroup at the end, not one per row
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
        const long b = row / rpb, i = row % rpb;
        const float* xr = x + b * x_bs + i * D;
        bf16_t* yr = y + b * y_bs + i * D;
        unsigned char* qr = q + b * y_bs + i * D;
        f4 v[LNV_CH];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LNV_CH; k++) {
            const int c = 256 * k + 4 * lane;
            if (c < D) { v[k] = ld4(xr + c); s += v[k][0] + v[k][1] + v[k][2] + v[k][3]; }
            else v[k] = (f4){0.f, 0.f, 0.f, 0.f};
        }
        const float mu = wave_sum(s) / D;
        float qq = 0.f;
#pragma unroll
        for (int k = 0; k < LNV_CH; k++) {
            const int c = 256 * k + 4 * lane;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 4; e++) { const float d = v[k][e] - mu; qq += d * d; }
            }
        }
        const float rs = rsqrtf(wave_sum(qq) / D + eps);
#pragma unroll
        for (int k = 0; k < LNV_CH; k++) {
            const int c = 256 * k + 4 * lane;
            if (c < D) {
                const f4 g = ld4(gamma + c), bt = ld4(beta + c);
                f4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = (v[k][e] - mu) * rs * g[e] + bt[e];
                st4(yr + c, o);
                // quantise what the bf16 copy holds (the same operand values as the separate quantisation pass would read)
#pragma unroll
                for (int e = 0; e < 4; e++) { o[e] = bf2f(f2bf(o[e])); m = fmaxf(m, fabsf(o[e])); }
                const f4 w4 = o * mul;
                unsigned w = 0;
                w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(w4[0], -F8M), F8M), fminf(fmaxf(w4[1], -F8M), F8M), w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(w4[2], -F8M), F8M), fminf(fmaxf(w4[3], -F8M), F8M), w, true);
                *reinterpret_cast<unsigned*>(qr + c) = w;
            }
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
    m = block_max256(m, red);
    if (threadIdx.x == 0) atomicMax(ring + cur, __float_as_uint(m));
}

template <typename TX, typename TDY, int LNV_CH>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, TX* __restrict__ dx,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                long rows, int rpb, int D, long x_bs, long y_bs, int acc_dx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 pg[LNV_CH], pb[LNV_CH], gm[LNV_CH];
#pragma unroll
    for (int k = 0; k < LNV_CH; k++) {
        pg[k] = (f4){0.f, 0.f, 0.f, 0.f};
        pb[k] = pg[k];
        const int c = 256 * k + 4 * lane;
        gm[k] = (c < D) ? ld4(gamma + c) : pg[k];
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const long b = row / rpb, i = row % rpb;
        const TX* xr = x + b * x_bs + i * D;
        const TDY* dyr = dy + b * y_bs + i * D;
        TX* dxr = dx + b * x_bs + i * D;
        const float mu = mean[row], rs = rstd[row];
        f4 xh[LNV_CH], gd[LNV_CH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LNV_CH; k++) {
            const int c = 256 * k + 4 * lane;
            if (c < D) {
                const f4 d = ld4(dyr + c), xv = ld4(xr + c);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    xh[k][e] = (xv[e] - mu) * rs;
                    gd[k][e] = d[e] * gm[k][e];
                    pg[k][e] += d[e] * xh[k][e];
                    pb[k][e] += d[e];
                    s1 += gd[k][e];
                    s2 += gd[k][e] * xh[k][e];
                }
            }
        }
        s1 = wave_sum(s1) / D;
        s2 = wave_sum(s2) / D;
#pragma unroll
        for (int k = 0; k < LNV_CH; k++) {
            const int c = 256 * k + 4 * lane;
            if (c < D) {
                f4 r;
#pragma unroll
                for (int e = 0; e < 4; e++) r[e] = rs * (gd[k][e] - s1 - xh[k][e] * s2);
                if (acc_dx) r += ld4(dxr + c);
                st4(dxr + c, r);
            }
        }
    }
    __shared__ f4 red[2][4][64];
#pragma unroll
    for (int k = 0; k < LNV_CH; k++) {
        if (256 * k >= D) break;
        __syncthreads();
        red[0][wave][lane] = pg[k];
        red[1][wave][lane] = pb[k];
        __syncthreads();
        if (wave == 0) {
            const int c = 256 * k + 4 * lane;
            if (c < D) {
                const f4 g = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
                const f4 bb = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
#pragma unroll
                for (int e = 0; e < 4; e++) { atomicAdd(dgamma + c + e, g[e]); atomicAdd(dbeta + c + e, bb[e]); }
            }
        }
    }
}

extern "C" int mh_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                int batches, int rpb, int D, int64_t x_bs, int64_t y_bs, float eps, int dt_x, int dt_y,
                                mh_stream s) {
    MH_REQUIRE(D >= 1 && D <= 64 * LN_MAXPL, "mh_layernorm_fwd: D=%d unsupported (max %d)", D, 64 * LN_MAXPL);
    const long rows = (long)batches * rpb;
    if (rows == 0) return MH_OK;
    dim3 grid(mh_cdiv(rows, 4));
    const bool vecok = D % 4 == 0 && x_bs % 4 == 0 && y_bs % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                       ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0;
#define LN_FV1(TX, TY, NC) hipLaunchKernelGGL((layernorm_fwd_vec_kernel<TX, TY, NC>), grid, dim3(256), 0, (hipStream_t)s, (const TX*)x, gamma, beta, (TY*)y, mean, rstd, rows, rpb, D, (long)x_bs, (long)y_bs, eps)
#define LN_FV(TX, TY) do { if (D <= 512) LN_FV1(TX, TY, 2); else if (D <= 1024) LN_FV1(TX, TY, 4); else LN_FV1(TX, TY, 8); } while (0)
    if (vecok) {
        if (dt_x == MH_F32 && dt_y == MH_F32) LN_FV(float, float);
        else if (dt_x == MH_F32 && dt_y == MH_BF16) LN_FV(float, bf16_t);
        else if (dt_x == MH_BF16 && dt_y == MH_BF16) LN_FV(bf16_t, bf16_t);
        else LN_FV(bf16_t, float);
        MH_LAUNCH_CHECK("mh_layernorm_fwd");
        return MH_OK;
    }
#undef LN_FV
#undef LN_FV1
#define LN_F(TX, TY) hipLaunchKernelGGL((layernorm_fwd_kernel<TX, TY>), grid, dim3(256), 0, (hipStream_t)s, (const TX*)x, gamma, beta, (TY*)y, mean, rstd, rows, rpb, D, (long)x_bs, (long)y_bs, eps)
    if (dt_x == MH_F32 && dt_y == MH_F32) LN_F(float, float);
    else if (dt_x == MH_F32 && dt_y == MH_BF16) LN_F(float, bf16_t);
    else if (dt_x == MH_BF16 && dt_y == MH_BF16) LN_F(bf16_t, bf16_t);
    else LN_F(bf16_t, float);
#undef LN_F
    MH_LAUNCH_CHECK("mh_layernorm_fwd");
    return MH_OK;
}

extern "C" int mh_layernorm_fwd_lm(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, void* xpm,
                                   void* xpm_bf16, int batches, int rows, int D, int64_t x_bs, int pad, int l, float eps, const float* row_mask,
                                   const float* lm_scale, mh_stream s) {
    MH_REQUIRE(l >= 1 && pad >= 0 && rows >= 1 && (pad + rows) % l == 0, "mh_layernorm_fwd_lm: pad + rows = %d must be a multiple of l = %d", pad + rows, l);
    MH_REQUIRE(D % 4 == 0 && D <= 2048 && x_bs % 4 == 0 && (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 &&
                   ((uintptr_t)y & 7) == 0 && ((uintptr_t)xpm & 15) == 0 && ((uintptr_t)xpm_bf16 & 7) == 0,
               "mh_layernorm_fwd_lm: D %% 4 == 0, D <= 2048 and aligned buffers (D=%d)", D);
    if (batches == 0) return MH_OK;
    const int m = (pad + rows) / l, groups = batches * m;
    // waves per group: enough waves to fill the chip (>= 16 per CU) while a wave keeps >= 4 rows
    // (config 4, B = 8: 2048 groups of 33 rows -> 2 waves each, 49 -> 38 us alone; c2's 4096 groups of 17 rows stay one wave each: 38.5 us vs 41)
    int split = 1;
    while (split < 4 && (long)groups * split < 4096 && l >= 8 * split && groups % (4 / (2 * split)) == 0) split *= 2;
#ifdef MH_EXP
    if (const char* e = getenv("MH_LN_LM_SPLIT")) split = atoi(e);      // tools/exp/time_ln_lm.py, tools/exp/ab_ln_lm_split.sh
#endif
    MH_REQUIRE((split == 1 || split == 2 || split == 4) && groups % (4 / split) == 0, "mh_layernorm_fwd_lm: %d waves per group with %d groups", split, groups);
    dim3 grid(mh_cdiv(groups, 4 / split));
#define LNL_(NC, SP) hipLaunchKernelGGL((layernorm_fwd_lm_kernel<NC, SP>), grid, dim3(256), 0, (hipStream_t)s, x, gamma, beta, (bf16_t*)y, mean, rstd, (float*)xpm, (bf16_t*)xpm_bf16, groups, m, rows, D, (long)x_bs, pad, l, eps, row_mask, lm_scale)
#define LNL(NC) do { if (split == 1) LNL_(NC, 1); else if (split == 2) LNL_(NC, 2); else LNL_(NC, 4); } while (0)
    if (D <= 512) LNL(2); else if (D <= 1024) LNL(4); else LNL(8);
#undef LNL
#undef LNL_
    MH_LAUNCH_CHECK("mh_layernorm_fwd_lm");
    return MH_OK;
}

// f32 output + a bf16 copy with the same row addressing, one pass: the encoder's final norm feeds an f32 consumer (retention
// target, cls row: models/mirror.py:699, :684) and a projection that takes bf16 operands (retention_embed, :690) — without the
// copy the projection's input costs a separate 214 MB cast pass.
extern "C" int mh_layernorm_fwd_dual(const float* x, const float* gamma, const float* beta, float* y, void* y_bf16, float* mean, float* rstd,
                                     int batches, int rpb, int D, int64_t x_bs, int64_t y_bs, float eps, mh_stream s) {
    const long rows = (long)batches * rpb;
    if (rows == 0) return MH_OK;
    MH_REQUIRE(y_bf16 && D % 4 == 0 && D <= 2048 && x_bs % 4 == 0 && y_bs % 4 == 0 &&
                   (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && ((uintptr_t)y_bf16 & 7) == 0,
               "mh_layernorm_fwd_dual: D %% 4 == 0, D <= 2048 and aligned buffers (D=%d)", D);
    dim3 grid(mh_cdiv(rows, 4));
#define LND(NC) hipLaunchKernelGGL((layernorm_fwd_vec_kernel<float, float, NC>), grid, dim3(256), 0, (hipStream_t)s, x, gamma, beta, y, mean, rstd, rows, rpb, D, (long)x_bs, (long)y_bs, eps, (bf16_t*)y_bf16)
    if (D <= 512) LND(2); else if (D <= 1024) LND(4); else LND(8);
#undef LND
    MH_LAUNCH_CHECK("mh_layernorm_fwd_dual");
    return MH_OK;
}

extern "C" int mh_layernorm_fwd_q8(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                   int batches, int rows_per_batch, int D, int64_t x_bs, int64_t y_bs, float eps, void* q8, unsigned* ring,
                                   const float* tick, float margin, float* scale, mh_stream s) {
    const long rows = (long)batches * rows_per_batch;
    if (rows == 0) return MH_OK;
    MH_REQUIRE(D % 4 == 0 && D <= 2048 && x_bs % 4 == 0 && y_bs % 4 == 0 && (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 &&
                   ((uintptr_t)y & 7) == 0 && ((uintptr_t)q8 & 3) == 0,
               "mh_layernorm_fwd_q8: D %% 4 == 0, D <= 2048 and aligned buffers (D=%d)", D);
    MH_REQUIRE(ring && tick && scale && margin >= 1.f, "mh_layernorm_fwd_q8: ring, tick, scale and margin >= 1 are required");
    dim3 grid((unsigned)min((long)mh_cdiv(rows, 4), 4096L));
#define LNQ(NC) hipLaunchKernelGGL((layernorm_fwd_q8_kernel<NC>), grid, dim3(256), 0, (hipStream_t)s, x, gamma, beta, (bf16_t*)y, mean, rstd, rows, rows_per_batch, D, (long)x_bs, (long)y_bs, eps, (unsigned char*)q8, ring, tick, margin, scale)
    if (D <= 512) LNQ(2); else if (D <= 1024) LNQ(4); else LNQ(8);
#undef LNQ
    MH_LAUNCH_CHECK("mh_layernorm_fwd_q8");
    return MH_OK;
}

// LayerNorm backward, workspace form: dx as above, but (a) two rows per wave are in flight per iteration (the row loop
// is a serial load -> reduce -> store chain: one row per wave left HBM at 1.8 TB/s), (b) 32-bit row arithmetic, and
// (c) the per-block dgamma / dbeta partials go to ws[block][2][D] with plain stores — 2048 blocks adding into the same
// 2 D addresses cost ~100 us of serialised atomics — and a second small kernel folds them.
template <typename TX, typename TDY, int LNV_CH, bool RELU = false, bool FAN = false, bool DROP = false>
// D <= 512 (LNV_CH == 2): held to 4 waves per SIMD — at 130 registers instead of 128 the ReLU-fused instance lost a wave and 30 us
__global__ __launch_bounds__(256, LNV_CH == 2 ? 4 : 1) void layernorm_bwd_ws_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                               const float* __restrict__ gamma, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, TX* __restrict__ dx,
                                                               float* __restrict__ ws, int rows, int rpb, int D, long x_bs,
                                                               long y_bs, int acc_dx, int rows_per_block,
                                                               const TDY* __restrict__ gadd = nullptr, int ga_pad = 0, int ga_l = 1,
                                                               int ga_m = 0, float ga_scale = 0.f, bf16_t* __restrict__ relu_out = nullptr,
                                                               int relu_first = 0, int relu_rows = 0, int relu_cs = 0,
                                                               const bf16_t* __restrict__ fan = nullptr, float fan_alpha = 0.f,
                                                               const float* __restrict__ fan_cls = nullptr,
                                                               bf16_t* __restrict__ drop_out = nullptr, float drop_p = 0.f, uint64_t drop_seed = 0,
                                                               uint64_t drop_offset = 0, const uint64_t* __restrict__ drop_base = nullptr,
                                                               int drop_rpb = 0, const float* __restrict__ rmask = nullptr,
                                                               const float* __restrict__ lm_scale = nullptr) {
    // rmask [batches, ga_pad + rpb] (with gadd; may be null): the forward's row mask — a masked row's output was zero whatever x was, so its
    // total dy (its own and its share of the landmark gradient) is zero
    // DROP (round 5): x is the output of `resid + Dropout_p(Linear(.))` ([3P] to_out = Sequential(Linear, Dropout) + TransLayer's residual
    // add, models/mirror.py:312), so this launch's dx IS that Dropout's upstream gradient: drop_out [rows, D] bf16 receives
    // mask * dx / (1 - p) on the lite Philox stream (element (row, c) = 16-bit field of block (offset + row D + c) >> 3: the masks of the
    // forward epilogue), the operand of to_out's two gradient products, and a third partial row its column sums (to_out's bias
    // gradient) — mh_dropout_lite_colsum's pass over dx (read 4 B, write 2 B per element) is not launched.  drop_rpb >= rpb: rows per batch of
    // the Dropout's tensor (the norm may read only the first rpb of them: square-padded sequences; the caller zero-fills the rest)
    // FAN (round 5; the WSI encoder's final norm, whose output feeds the decoder, the retention target and the cls heads,
    // models/mirror.py:684-700): dy of row i >= 1 of batch b is dy + fan_alpha * fan[b, i - 1] (fan bf16 [batches, rpb - 1, D]: the
    // masked MSE's -dpred), of row 0 dy + fan_cls[b] (f32 [batches, D], may be null) — the sum mh_fanout_bwd would have written as a
    // [B, T, D] f32 tensor for this launch to read back
    // relu_cs (RELU only): a third partial row ws[block][2][D] = column sums of what went to relu_out (as rounded to bf16): the bias
    // gradient of the Linear in front of the ReLU, folded by the same fold launch — ws is then [blocks][3][D]
    // relu_out (bf16 [batches, relu_rows, D], round 5): x is the OUTPUT of a ReLU on rows [relu_first, relu_first + relu_rows) of every batch
    // (_fc1 of models/mirror.py:346, :652-654 writes the sequence LayerNorm 1 reads): those rows' total gradient leaves as
    // bf16 (x > 0 ? dx : 0) — what the ReLU's own backward pass would make of it for the weight-gradient product — instead of as f32 dx
    // (the pass that read x and dx again, 335 MB, is gone; rows outside the range, the cls row, keep their f32 dx).
    // gadd [batches, ga_m, D] in dy's dtype: dy of row i of batch b is dy + ga_scale * gadd[b, (i + ga_pad) / ga_l] (the gradient of the
    // landmark means mh_layernorm_fwd_lm produced beside the rows)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // RELU: a third running sum (pr) — its 8 registers would cost the instance its 4th wave per SIMD (measured: 129 us instead of 99), and
    // LDS atomics for it cost more still (256 us): gamma moves to LDS instead (read back per row, 2 x ds_read_b128) and pr takes its registers
    constexpr bool THIRD = RELU || DROP;        // a third running column sum: gamma is parked in LDS to make room for it
    __shared__ f4 sgm[THIRD ? LNV_CH : 1][64];
    f4 pg[LNV_CH], pb[LNV_CH], gm[THIRD ? 1 : LNV_CH], pr[THIRD ? LNV_CH : 1];
    uint32_t dthr = 0;
    float dscale = 1.f;
    if constexpr (DROP) {
        if (drop_base) drop_offset += *drop_base & ~7ull;
        dthr = drop16_thr(drop_p);
        dscale = drop16_scale(dthr);
    }
#pragma unroll
    for (int k = 0; k < LNV_CH; k++) {
        pg[k] = (f4){0.f, 0.f, 0.f, 0.f};
        pb[k] = pg[k];
        if constexpr (THIRD) pr[k] = pg[k];
        const int c = 256 * k + 4 * lane;
        if constexpr (THIRD) {
            if (wave == 0) sgm[k][lane] = (c < D) ? ld4(gamma + c) : pg[k];
        } else gm[k] = (c < D) ? ld4(gamma + c) : pg[k];
    }
    if constexpr (THIRD) __syncthreads();
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int row = r0 + wave; row < r1; row += 8) {
        // DROP: the keep bits of this iteration's quads depend on (row, column) alone: drawn in FRONT of the loads (under them the Philox
        // state competed with 48 load-destination registers).  A Philox block serves 8 elements = the quads of lanes 2 j and 2 j + 1 of
        // one (row, chunk): the even lane draws the blocks of row `row`, the odd lane those of row `row + 4`, and they swap (one DPP move)
        // — every block drawn once, as in mh_dropout_lite_colsum (drawn by both lanes the integer multiplies made this launch as long as
        // the two it replaces).  keepbits: one nibble per (u, k) quad of this lane.
        uint32_t keepbits = 0;
        if constexpr (DROP) {
            static_assert(LNV_CH <= 4, "one byte per chunk in a 32-bit word");
            const int odd = lane & 1;
            const int rr = row + 4 * odd;
            uint32_t mine = 0;
            if (rr < r1) {
                const long orow = drop_rpb == rpb ? (long)rr : (long)(rr / rpb) * drop_rpb + (rr % rpb);       // its row in the Dropout's tensor
#pragma unroll
                for (int k = 0; k < LNV_CH; k++) {
                    const int c8 = 256 * k + 8 * (lane >> 1);
                    if (c8 < D) mine |= drop16_keep8((drop_offset + (uint64_t)(orow * D + c8)) >> 3, drop_seed, dthr) << (8 * k);
                }
            }
            const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t src = (odd == u) ? mine : other;
#pragma unroll
                for (int k = 0; k < LNV_CH; k++) keepbits |= (((src >> (8 * k)) >> (4 * odd)) & 15u) << (4 * (u * LNV_CH + k));
            }
            asm volatile("" : "+v"(keepbits));
        }
        f4 dv[2][LNV_CH], xv[2][LNV_CH], ov[2][LNV_CH];
        float mu[2], rs[2];
        long xo[2], ro[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {               // issue both rows' loads before anything is reduced
            const int rr = row + 4 * u;
            ok[u] = rr < r1;
            const int rc = ok[u] ? rr : row;
            const int b = rc / rpb, i = rc - b * rpb;
            xo[u] = b * x_bs + (long)i * D;
            ro[u] = (RELU && i >= relu_first && i < relu_first + relu_rows) ? ((long)b * relu_rows + (i - relu_first)) * D
                    : (DROP ? ((long)b * drop_rpb + i) * D : -1);     // DROP: the row's first element in the Dropout's [batches, drop_rpb, D] tensor
            const TDY* dyr = dy + b * y_bs + (long)i * D;
            mu[u] = mean[rc];
            rs[u] = rstd[rc];
#pragma unroll
            for (int k = 0; k < LNV_CH; k++) {
                const int c = 256 * k + 4 * lane;
                if (c < D) {
                    dv[u][k] = ld4(dyr + c);
                    xv[u][k] = ld4(x + xo[u] + c);
                    if (acc_dx) ov[u][k] = ld4(dx + xo[u] + c);
                    if (gadd) dv[u][k] += ld4(gadd + ((long)b * ga_m + (i + ga_pad) / ga_l) * D + c) *
                                          (lm_scale ? ga_scale * lm_scale[(long)b * ga_m + (i + ga_pad) / ga_l] : ga_scale);     // (the forward's landmark scale)
                    if (rmask && rmask[(long)b * (ga_pad + rpb) + ga_pad + i] == 0.f) dv[u][k] = (f4){0.f, 0.f, 0.f, 0.f};
                    if constexpr (FAN) {
                        if (i >= 1) dv[u][k] += ld4(fan + ((long)b * (rpb - 1) + (i - 1)) * D + c) * fan_alpha;
                        else if (fan_cls) dv[u][k] += ld4(fan_cls + (long)b * D + c);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (!ok[u]) continue;
            float s1 = 0.f, s2 = 0.f;
            unsigned pos = 0;               // bit 4 k + e: x > 0 (the ReLU was active)
#pragma unroll
            for (int k = 0; k < LNV_CH; k++) {
                const int c = 256 * k + 4 * lane;
                if (c < D) {
                    f4 gq;
                    if constexpr (THIRD) gq = sgm[k][lane]; else gq = gm[k];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if constexpr (RELU) pos |= (xv[u][k][e] > 0.f ? 1u : 0u) << (4 * k + e);
                        const float xh = (xv[u][k][e] - mu[u]) * rs[u];
                        const float d = dv[u][k][e];
                        const float gd = d * gq[e];
                        pg[k][e] += d * xh;
                        pb[k][e] += d;
                        s1 += gd;
                        s2 += gd * xh;
                        xv[u][k][e] = xh;
                        dv[u][k][e] = gd;
                    }
                }
            }
            s1 = wave_sum(s1) / D;
            s2 = wave_sum(s2) / D;
#pragma unroll
            for (int k = 0; k < LNV_CH; k++) {
                const int c = 256 * k + 4 * lane;
                if (c < D) {
                    f4 r;
#pragma unroll
                    for (int e = 0; e < 4; e++) r[e] = rs[u] * (dv[u][k][e] - s1 - xv[u][k][e] * s2);
                    if (acc_dx) r += ov[u][k];
                    if (RELU && !DROP && ro[u] >= 0) {
                        typedef unsigned ln_u32x2 __attribute__((ext_vector_type(2)));
                        const ln_u32x2 w = {pack_bf2((pos >> (4 * k)) & 1u ? r[0] : 0.f, (pos >> (4 * k + 1)) & 1u ? r[1] : 0.f),
                                         pack_bf2((pos >> (4 * k + 2)) & 1u ? r[2] : 0.f, (pos >> (4 * k + 3)) & 1u ? r[3] : 0.f)};
                        *reinterpret_cast<ln_u32x2*>(relu_out + ro[u] + c) = w;
                        pr[k] += (f4){__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u),
                                      __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
                    } else {
                        st4(dx + xo[u] + c, r);
                        if constexpr (DROP) {
                            typedef unsigned ln_u32x2d __attribute__((ext_vector_type(2)));
                            const long e0 = ro[u] + c;
                            const uint32_t keep = keepbits >> (4 * (u * LNV_CH + k));
                            const ln_u32x2d w = {pack_bf2((keep & 1u) ? __fmul_rn(r[0], dscale) : 0.f, (keep & 2u) ? __fmul_rn(r[1], dscale) : 0.f),
                                                 pack_bf2((keep & 4u) ? __fmul_rn(r[2], dscale) : 0.f, (keep & 8u) ? __fmul_rn(r[3], dscale) : 0.f)};
                            *reinterpret_cast<ln_u32x2d*>(drop_out + e0) = w;
                            pr[k] += (f4){__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u),
                                          __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
                        }
                    }
                }
            }
        }
    }
    __shared__ f4 red[THIRD ? 3 : 2][4][64];
    const int segs = ((RELU && relu_cs) || DROP) ? 3 : 2;
    float* wsb = ws + (long)blockIdx.x * segs * D;
#pragma unroll
    for (int k = 0; k < LNV_CH; k++) {
        if (256 * k >= D) break;
        __syncthreads();
        red[0][wave][lane] = pg[k];
        red[1][wave][lane] = pb[k];
        if constexpr (THIRD) red[2][wave][lane] = pr[k];
        __syncthreads();
        if (wave == 0) {
            const int c = 256 * k + 4 * lane;
            if (c < D) {
                *reinterpret_cast<f4*>(wsb + c) = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
                *reinterpret_cast<f4*>(wsb + D + c) = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
                if constexpr (THIRD)
                    if (DROP || relu_cs) *reinterpret_cast<f4*>(wsb + 2 * D + c) = red[2][0][lane] + red[2][1][lane] + red[2][2][lane] + red[2][3][lane];
            }
        }
    }
}

// dgamma[c] += sum_b ws[b][0][c], dbeta[c] += sum_b ws[b][1][c]; grid (ceil(2D / 256), splits)
__global__ __launch_bounds__(256) void layernorm_bwd_fold_kernel(const float* __restrict__ ws, int nb, int D, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * D) return;
    const int per = (nb + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(nb, b0 + per);
    float s = 0.f;
    for (int b = b0; b < b1; b++) s += ws[(long)b * 2 * D + t];
    atomicAdd(t < D ? dgamma + t : dbeta + (t - D), s);
}

// the same with one QUAD of columns per lane, the partial rows dealt to the four waves of a workgroup (eight loads in flight each) and
// summed through LDS: the scalar form walks ~29 rows per thread one dependent 4-byte load at a time, and every same-address atomic
// costs ~0.13 us (a 114-way row split was slower than the fold it replaced) — 64 row chunks, 16 atomics per address
// segs = 3: a third partial row per block (the ReLU-fused instance's bias column sums) -> dthird
__global__ __launch_bounds__(256) void layernorm_bwd_fold4_kernel(const float* __restrict__ ws, int nb, int D, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, int segs = 2, float* __restrict__ dthird = nullptr) {
    typedef float lf4 __attribute__((ext_vector_type(4)));
    __shared__ lf4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = (blockIdx.x * 64 + lane) * 4;
    const bool live = t < segs * D;
    const int chunks = gridDim.y * 4, per = (nb + chunks - 1) / chunks;
    const int b0 = (blockIdx.y * 4 + wave) * per, b1 = min(nb, b0 + per);
    lf4 s = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            lf4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const lf4*>(ws + (long)(b + u) * segs * D + t);
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; b < b1; b++) s += *reinterpret_cast<const lf4*>(ws + (long)b * segs * D + t);
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && live) {
        s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        float* dst = t < D ? dgamma + t : (t < 2 * D ? dbeta + (t - D) : dthird + (t - 2 * D));      // D % 4 == 0: a quad never straddles two segments
#pragma unroll
        for (int e = 0; e < 4; e++) atomicAdd(dst + e, s[e]);
    }
}

static int ln_bwd_blocks() { return 512; }

extern "C" int64_t mh_layernorm_bwd_workspace_bytes(int64_t rows, int D) {
    if (rows < 64 || D <= 0) return 0;
    const int64_t nb = rows / 16 < 1024 ? (rows / 16 < 1 ? 1 : rows / 16) : 1024;
    return 3 * (int64_t)D * nb * 4;      // three partial rows per block: dgamma, dbeta and (relu_db) a bias gradient — at two the ReLU-fused
                                         // instance ran 631 blocks instead of 911 (62 % of the chip's slots: 119 us instead of ~100)
}

static int ln_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                       void* dx, float* dgamma, float* dbeta, int batches, int rpb, int D, int64_t x_bs,
                       int64_t y_bs, int dt_x, int dt_dy, int dt_dx, int acc_dx, float* workspace, int64_t ws_floats,
                       const void* gadd, int ga_pad, int ga_l, mh_stream s, void* relu_out = nullptr, int relu_first = 0, int relu_rows = 0,
                       float* relu_db = nullptr, const void* fan = nullptr, float fan_alpha = 0.f, const float* fan_cls = nullptr,
                       void* drop_out = nullptr, float drop_p = 0.f, uint64_t drop_seed = 0, uint64_t drop_offset = 0,
                       const uint64_t* drop_base = nullptr, float* drop_db = nullptr, int drop_rpb = 0, const float* rmask = nullptr, const float* lm_scale = nullptr);

extern "C" int mh_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                void* dx, float* dgamma, float* dbeta, int batches, int rpb, int D, int64_t x_bs,
                                int64_t y_bs, int dt_x, int dt_dy, int dt_dx, int acc_dx, float* workspace, int64_t ws_floats,
                                mh_stream s) {
    return ln_bwd_impl(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, batches, rpb, D, x_bs, y_bs, dt_x, dt_dy, dt_dx, acc_dx, workspace,
                       ws_floats, nullptr, 0, 1, s);
}

// the backward of mh_layernorm_fwd_lm: dy of row i of batch b is dy[b, i] + gadd[b, (i + pad) / l] / l, gadd [batches, (pad + rpb) / l, D]
// in dy's dtype = the gradient of the landmark means (needs the workspace form: D % 4 == 0, aligned buffers, a workspace)
extern "C" int mh_layernorm_bwd_lm(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                   void* dx, float* dgamma, float* dbeta, int batches, int rpb, int D, int64_t x_bs,
                                   int64_t y_bs, int dt_x, int dt_dy, int dt_dx, int acc_dx, float* workspace, int64_t ws_floats,
                                   const void* gadd, int pad, int l, void* relu_out, int relu_first, int relu_rows, float* relu_db,
                                   const float* row_mask, const float* lm_scale, mh_stream s) {
    MH_REQUIRE(gadd && l >= 1 && pad >= 0 && (pad + rpb) % l == 0 && ((uintptr_t)gadd & 15) == 0, "mh_layernorm_bwd_lm: gadd, l >= 1, (pad + rows) %% l == 0");
    MH_REQUIRE(!relu_out || (dt_x == MH_F32 && relu_first >= 0 && relu_rows >= 0 && relu_first + relu_rows <= rpb && ((uintptr_t)relu_out & 7) == 0 && D % 4 == 0),
               "mh_layernorm_bwd_lm: relu_out needs f32 x and a row range inside the batch");
    MH_REQUIRE(!relu_db || (relu_out && ((uintptr_t)workspace & 15) == 0 && ws_floats >= 3L * D), "mh_layernorm_bwd_lm: relu_db rides on relu_out (workspace of >= 3 D floats)");
    return ln_bwd_impl(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, batches, rpb, D, x_bs, y_bs, dt_x, dt_dy, dt_dx, acc_dx, workspace,
                       ws_floats, gadd, pad, l, s, relu_out, relu_first, relu_rows, relu_db, nullptr, 0.f, nullptr, nullptr, 0.f, 0, 0, nullptr, nullptr, 0,
                       row_mask, lm_scale);
}

// mh_layernorm_bwd whose dy rows also receive the other two gradients of a fanned-out LayerNorm output (see FAN above): needs the
// workspace form with f32 x and f32 dy
extern "C" int mh_layernorm_bwd_fan(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                    void* dx, float* dgamma, float* dbeta, int batches, int rpb, int D, int64_t x_bs,
                                    int64_t y_bs, int dt_dy, int accumulate_dx, float* workspace, int64_t ws_floats,
                                    const void* fan_bf16, float fan_alpha, const float* fan_cls, mh_stream s) {
    MH_REQUIRE(fan_bf16 && rpb >= 2 && ((uintptr_t)fan_bf16 & 7) == 0 && (!fan_cls || ((uintptr_t)fan_cls & 15) == 0) && D % 4 == 0,
               "mh_layernorm_bwd_fan: fan [batches, rows - 1, D] bf16 on quads, rows >= 2");
    return ln_bwd_impl(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, batches, rpb, D, x_bs, y_bs, MH_F32, dt_dy, MH_F32, accumulate_dx, workspace,
                       ws_floats, nullptr, 0, 1, s, nullptr, 0, 0, nullptr, fan_bf16, fan_alpha, fan_cls);
}

// mh_layernorm_bwd (optionally with mh_layernorm_bwd_fan's two extra gradients) for a LayerNorm whose input x is the output of
// `resid + Dropout_p(Linear(core))`: drop_out [batches * rows, D] bf16 = dropout backward of this launch's dx on the lite stream,
// drop_db [D] += its column sums.  f32 x / dx, dy f32 or bf16 (f32 with fan), the workspace form
extern "C" int mh_layernorm_bwd_drop(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                     void* dx, float* dgamma, float* dbeta, int batches, int rpb, int D, int64_t x_bs,
                                     int64_t y_bs, int dt_dy, int accumulate_dx, float* workspace, int64_t ws_floats,
                                     const void* fan_bf16, float fan_alpha, const float* fan_cls,
                                     void* drop_out, float drop_p, uint64_t drop_seed, uint64_t drop_offset, const uint64_t* drop_base,
                                     float* drop_db, int drop_rows_per_batch, mh_stream s) {
    MH_REQUIRE(drop_rows_per_batch >= rpb, "mh_layernorm_bwd_drop: the Dropout's tensor has %d rows per batch, the norm reads %d", drop_rows_per_batch, rpb);
    MH_REQUIRE(drop_out && drop_db && drop_p >= 0.f && drop_p < 1.f && (drop_offset & 7) == 0 && ((uintptr_t)drop_out & 7) == 0 && D % 8 == 0 &&
                   ((uintptr_t)workspace & 15) == 0 && ws_floats >= 3L * D,
               "mh_layernorm_bwd_drop: drop_out / drop_db, p in [0, 1), offset %% 8 == 0, D %% 8 == 0, a workspace of >= 3 D floats");
    MH_REQUIRE(!fan_bf16 || (rpb >= 2 && ((uintptr_t)fan_bf16 & 7) == 0 && (!fan_cls || ((uintptr_t)fan_cls & 15) == 0)),
               "mh_layernorm_bwd_drop: fan [batches, rows - 1, D] bf16 on quads");
    return ln_bwd_impl(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, batches, rpb, D, x_bs, y_bs, MH_F32, dt_dy, MH_F32, accumulate_dx, workspace,
                       ws_floats, nullptr, 0, 1, s, nullptr, 0, 0, nullptr, fan_bf16, fan_alpha, fan_cls, drop_out, drop_p, drop_seed, drop_offset,
                       drop_base, drop_db, drop_rows_per_batch);
}

static int ln_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                       void* dx, float* dgamma, float* dbeta, int batches, int rpb, int D, int64_t x_bs,
                       int64_t y_bs, int dt_x, int dt_dy, int dt_dx, int acc_dx, float* workspace, int64_t ws_floats,
                       const void* gadd, int ga_pad, int ga_l, mh_stream s, void* relu_out, int relu_first, int relu_rows, float* relu_db,
                       const void* fan, float fan_alpha, const float* fan_cls, void* drop_out, float drop_p, uint64_t drop_seed,
                       uint64_t drop_offset, const uint64_t* drop_base, float* drop_db, int drop_rpb, const float* rmask, const float* lm_scale) {
    MH_REQUIRE(D >= 1 && D <= 64 * LN_MAXPL, "mh_layernorm_bwd: D=%d unsupported", D);
    MH_REQUIRE(dt_dx == dt_x, "mh_layernorm_bwd: dx dtype must equal x dtype");
    const long rows = (long)batches * rpb;
    if (rows == 0) return MH_OK;
    MH_REQUIRE(rows < (1L << 31), "mh_layernorm_bwd: too many rows");
    dim3 grid((unsigned)min((long)mh_cdiv(rows, 4), (long)ln_bwd_blocks()));
    const bool vecok = D % 4 == 0 && x_bs % 4 == 0 && y_bs % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 &&
                       ((uintptr_t)dx & 15) == 0 && ((uintptr_t)gamma & 15) == 0;
#define LN_BV1(TX, TDY, NC) hipLaunchKernelGGL((layernorm_bwd_vec_kernel<TX, TDY, NC>), grid, dim3(256), 0, (hipStream_t)s, (const TDY*)dy, (const TX*)x, gamma, mean, rstd, (TX*)dx, dgamma, dbeta, rows, rpb, D, (long)x_bs, (long)y_bs, acc_dx)
#define LN_BV(TX, TDY) do { if (D <= 512) LN_BV1(TX, TDY, 2); else if (D <= 1024) LN_BV1(TX, TDY, 4); else LN_BV1(TX, TDY, 8); } while (0)
    // workspace form: rows_per_block rows per block (a multiple of 8: two rows per wave per iteration), nb <= ws capacity
    if (vecok && workspace && ws_floats >= 2L * D && rows >= 64) {
        const int segs = (relu_db || drop_out) ? 3 : 2;       // partial rows per block: dgamma, dbeta (, a bias gradient's column sums)
        if (drop_out) relu_db = drop_db;                      // the fold's third destination
        const int relu_cs = relu_db ? 1 : 0;
        const long cap = ws_floats / ((long)segs * D);
        long nb = min(cap, min((long)mh_cdiv(rows, 16), 1024L));
        const int rows_per_block = (int)(mh_cdiv(mh_cdiv(rows, nb), 8) * 8);
        nb = mh_cdiv(rows, rows_per_block);
        dim3 g2((unsigned)nb);
#define LN_BW1(TX, TDY, NC, RL) hipLaunchKernelGGL((layernorm_bwd_ws_kernel<TX, TDY, NC, RL>), g2, dim3(256), 0, (hipStream_t)s, (const TDY*)dy, (const TX*)x, gamma, mean, rstd, (TX*)dx, workspace, (int)rows, rpb, D, (long)x_bs, (long)y_bs, acc_dx, rows_per_block, (const TDY*)gadd, ga_pad, ga_l, (ga_pad + rpb) / ga_l, 1.f / (float)ga_l, (bf16_t*)relu_out, relu_first, relu_rows, relu_cs, (const bf16_t*)nullptr, 0.f, (const float*)nullptr, (bf16_t*)nullptr, 0.f, (uint64_t)0, (uint64_t)0, (const uint64_t*)nullptr, 0, rmask, lm_scale)
#define LN_BW(TX, TDY) do { if (D <= 512) LN_BW1(TX, TDY, 2, false); else if (D <= 1024) LN_BW1(TX, TDY, 4, false); else LN_BW1(TX, TDY, 8, false); } while (0)
#define LN_BWF_(TDY, NC) hipLaunchKernelGGL((layernorm_bwd_ws_kernel<float, TDY, NC, false, true>), g2, dim3(256), 0, (hipStream_t)s, (const TDY*)dy, (const float*)x, gamma, mean, rstd, (float*)dx, workspace, (int)rows, rpb, D, (long)x_bs, (long)y_bs, acc_dx, rows_per_block, (const TDY*)nullptr, 0, 1, 0, 0.f, (bf16_t*)nullptr, 0, 0, 0, (const bf16_t*)fan, fan_alpha, fan_cls)
#define LN_BWF(NC) do { if (dt_dy == MH_F32) LN_BWF_(float, NC); else LN_BWF_(bf16_t, NC); } while (0)
#define LN_BWD_(TDY, NC, FN) hipLaunchKernelGGL((layernorm_bwd_ws_kernel<float, TDY, NC, false, FN, true>), g2, dim3(256), 0, (hipStream_t)s, (const TDY*)dy, (const float*)x, gamma, mean, rstd, (float*)dx, workspace, (int)rows, rpb, D, (long)x_bs, (long)y_bs, acc_dx, rows_per_block, (const TDY*)nullptr, 0, 1, 0, 0.f, (bf16_t*)nullptr, 0, 0, 0, (const bf16_t*)fan, fan_alpha, fan_cls, (bf16_t*)drop_out, drop_p, drop_seed, drop_offset, drop_base, drop_rpb)
        if (drop_out) {         // to_out's Dropout backward + bias gradient inside this launch: instances of their own (f32 x)
            MH_REQUIRE(dt_x == MH_F32 && D <= 1024, "mh_layernorm_bwd_drop: f32 x, D <= 1024");
            if (fan && dt_dy == MH_F32) { if (D <= 512) LN_BWD_(float, 2, true); else LN_BWD_(float, 4, true); }
            else if (fan) { if (D <= 512) LN_BWD_(bf16_t, 2, true); else LN_BWD_(bf16_t, 4, true); }
            else if (dt_dy == MH_F32) { if (D <= 512) LN_BWD_(float, 2, false); else LN_BWD_(float, 4, false); }
            else { if (D <= 512) LN_BWD_(bf16_t, 2, false); else LN_BWD_(bf16_t, 4, false); }
        } else
        if (fan) {              // the fanned-out form (f32 x, f32 dy) is an instance of its own too
            if (D <= 512) LN_BWF(2); else if (D <= 1024) LN_BWF(4); else LN_BWF(8);
        } else
        if (relu_out) {         // the ReLU-fused form (f32 x, bf16 dy: layer 1 of the bf16 policy) is an instance of its own
            MH_REQUIRE(dt_x == MH_F32 && dt_dy == MH_BF16, "mh_layernorm_bwd_lm: relu_out needs f32 x and bf16 dy");
            if (D <= 512) LN_BW1(float, bf16_t, 2, true); else if (D <= 1024) LN_BW1(float, bf16_t, 4, true); else LN_BW1(float, bf16_t, 8, true);
        } else if (dt_x == MH_F32 && dt_dy == MH_F32) LN_BW(float, float);
        else if (dt_x == MH_F32 && dt_dy == MH_BF16) LN_BW(float, bf16_t);
        else if (dt_x == MH_BF16 && dt_dy == MH_BF16) LN_BW(bf16_t, bf16_t);
        else LN_BW(bf16_t, float);
#undef LN_BWD_
#undef LN_BWF
#undef LN_BWF_
#undef LN_BW
#undef LN_BW1
        if (D % 4 == 0 && ((uintptr_t)workspace & 15) == 0)
            hipLaunchKernelGGL(layernorm_bwd_fold4_kernel, dim3(mh_cdiv(segs * D / 4, 64), (unsigned)min(mh_cdiv(nb, 32), 16)), dim3(256), 0,
                               (hipStream_t)s, (const float*)workspace, (int)nb, D, dgamma, dbeta, segs, relu_db);
        else
            hipLaunchKernelGGL(layernorm_bwd_fold_kernel, dim3(mh_cdiv(2 * D, 256), (unsigned)min(nb, 32L)), dim3(256), 0, (hipStream_t)s,
                               (const float*)workspace, (int)nb, D, dgamma, dbeta);
        MH_LAUNCH_CHECK("mh_layernorm_bwd");
        return MH_OK;
    }
    MH_REQUIRE(!drop_out, "mh_layernorm_bwd_drop: needs the workspace form (D %% 4 == 0, 16-byte aligned buffers, a workspace of >= 3 D floats, >= 64 rows)");
    MH_REQUIRE(!fan, "mh_layernorm_bwd_fan: needs the workspace form (D %% 4 == 0, 16-byte aligned buffers, a workspace of >= 2 D floats, >= 64 rows)");
    MH_REQUIRE(!gadd && !relu_out, "mh_layernorm_bwd_lm: needs the workspace form (D %% 4 == 0, 16-byte aligned buffers, a workspace of >= 2 D floats, >= 64 rows)");
    if (vecok) {
        if (dt_x == MH_F32 && dt_dy == MH_F32) LN_BV(float, float);
        else if (dt_x == MH_F32 && dt_dy == MH_BF16) LN_BV(float, bf16_t);
        else if (dt_x == MH_BF16 && dt_dy == MH_BF16) LN_BV(bf16_t, bf16_t);
        else LN_BV(bf16_t, float);
        MH_LAUNCH_CHECK("mh_layernorm_bwd");
        return MH_OK;
    }
#undef LN_BV
#undef LN_BV1
#define LN_B(TX, TDY) hipLaunchKernelGGL((layernorm_bwd_kernel<TX, TDY, TX>), grid, dim3(256), 0, (hipStream_t)s, (const TDY*)dy, (const TX*)x, gamma, mean, rstd, (TX*)dx, dgamma, dbeta, rows, rpb, D, (long)x_bs, (long)y_bs, acc_dx)
    if (dt_x == MH_F32 && dt_dy == MH_F32) LN_B(float, float);
    else if (dt_x == MH_F32 && dt_dy == MH_BF16) LN_B(float, bf16_t);
    else if (dt_x == MH_BF16 && dt_dy == MH_BF16) LN_B(bf16_t, bf16_t);
    else LN_B(bf16_t, float);
#undef LN_B
    MH_LAUNCH_CHECK("mh_layernorm_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------------------ softmax
// WAVE=true: one wave per row (4 rows per block); WAVE=false: one block per row.
template <typename TX, typename TY, bool WAVE>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const TX* x, TY* y, long rows,
                                                          int cols, long ldx, long ldy) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63;
    const long row = WAVE ? (long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long)blockIdx.x;
    if (WAVE && row >= rows) return;
    const int t0 = WAVE ? lane : threadIdx.x, step = WAVE ? 64 : 256;
    const TX* xr = x + row * ldx;
    TY* yr = y + row * ldy;
    float m = -INFINITY;
    for (int c = t0; c < cols; c += step) m = fmaxf(m, ldf(xr + c));
    m = WAVE ? wave_max(m) : block_max256(m, red);
    float sum = 0.f;
    for (int c = t0; c < cols; c += step) sum += __expf(ldf(xr + c) - m);
    sum = WAVE ? wave_sum(sum) : block_sum256(sum, red);
    const float inv = 1.f / sum;
    for (int c = t0; c < cols; c += step) stf(yr + c, __expf(ldf(xr + c) - m) * inv);
}

template <typename TY, typename TDY, typename TDX, bool WAVE>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const TY* y, const TDY* dy,
                                                          TDX* dx, long rows, int cols, long ldy, long lddy,
                                                          long lddx) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63;
    const long row = WAVE ? (long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long)blockIdx.x;
    if (WAVE && row >= rows) return;
    const int t0 = WAVE ? lane : threadIdx.x, step = WAVE ? 64 : 256;
    const TY* yr = y + row * ldy;
    const TDY* dyr = dy + row * lddy;
    TDX* dxr = dx + row * lddx;
    float dot = 0.f;
    for (int c = t0; c < cols; c += step) dot += ldf(yr + c) * ldf(dyr + c);
    dot = WAVE ? wave_sum(dot) : block_sum256(dot, red);
    for (int c = t0; c < cols; c += step) stf(dxr + c, ldf(yr + c) * (ldf(dyr + c) - dot));
}

// Register-cached, 4-wide variants (cols % 4 == 0): every element is read from HBM exactly once.
// WAVE: one wave per row, cols <= 1024 (lane keeps <= 4 f4); block: one 256-thread block per row, cols <= 12288.
template <typename TX, typename TY, bool WAVE, int NV>
__global__ __launch_bounds__(256) void softmax_fwd_vec_kernel(const TX* x, TY* y, long rows, int cols, long ldx, long ldy) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63;
    const long row = WAVE ? (long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long)blockIdx.x;
    if (WAVE && row >= rows) return;
    const int t0 = (WAVE ? lane : threadIdx.x) * 4, step = (WAVE ? 64 : 256) * 4;
    const TX* xr = x + row * ldx;
    TY* yr = y + row * ldy;
    f4 v[NV];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int c = t0 + k * step;
        if (c < cols) { v[k] = ld4(xr + c); m = fmaxf(fmaxf(fmaxf(m, v[k][0]), fmaxf(v[k][1], v[k][2])), v[k][3]); }
    }
    m = WAVE ? wave_max(m) : block_max256(m, red);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int c = t0 + k * step;
        if (c < cols) {
#pragma unroll
            for (int e = 0; e < 4; e++) { v[k][e] = __expf(v[k][e] - m); sum += v[k][e]; }
        }
    }
    sum = WAVE ? wave_sum(sum) : block_sum256(sum, red);
    const float inv = 1.f / sum;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int c = t0 + k * step;
        if (c < cols) st4(yr + c, v[k] * inv);
    }
}

template <typename TY, typename TD, bool WAVE, int NV>
__global__ __launch_bounds__(256) void softmax_bwd_vec_kernel(const TY* y, const TD* dy, TD* dx, long rows, int cols, long ldy,
                                                              long lddy, long lddx) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63;
    const long row = WAVE ? (long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long)blockIdx.x;
    if (WAVE && row >= rows) return;
    const int t0 = (WAVE ? lane : threadIdx.x) * 4, step = (WAVE ? 64 : 256) * 4;
    const TY* yr = y + row * ldy;
    const TD* dyr = dy + row * lddy;
    TD* dxr = dx + row * lddx;
    f4 yv[NV], dv[NV];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int c = t0 + k * step;
        if (c < cols) {
            yv[k] = ld4(yr + c);
            dv[k] = ld4(dyr + c);
            dot += yv[k][0] * dv[k][0] + yv[k][1] * dv[k][1] + yv[k][2] * dv[k][2] + yv[k][3] * dv[k][3];
        }
    }
    dot = WAVE ? wave_sum(dot) : block_sum256(dot, red);
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int c = t0 + k * step;
        if (c < cols) st4(dxr + c, yv[k] * (dv[k] - dot));
    }
}

// ---- key-padding-mask variants ([3P] NystromAttention.forward with `mask`: sim.masked_fill_(~(rowmask & colmask), -finfo.max)
// before the softmax).  x is [batches, h, R, C]; rowmask [batches, R] and colmask [batches, C] hold 0 / 1 floats.  An entry whose
// row or column is masked out is replaced by -FLT_MAX, so a fully masked row comes out uniform (1 / C), exactly like the
// package.  One 256-thread block per row (c4 parity path: not tuned).
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void softmax_masked_fwd_kernel(const TX* x, TY* y, const float* __restrict__ rowmask,
                                                                 const float* __restrict__ colmask, long rows, int cols, long rows_per_batch,
                                                                 int R) {
    __shared__ float red[4];
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const long b = row / rows_per_batch;
        const bool rv = rowmask[b * R + (row % R)] != 0.f;
        const float* cm = colmask + b * cols;
        const TX* xr = x + row * cols;
        TY* yr = y + row * cols;
        float mx = -3.402823466e38f;
        for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, (rv && cm[c] != 0.f) ? ldf(xr + c) : -3.402823466e38f);
        mx = block_max256(mx, red);
        float sum = 0.f;
        for (int c = threadIdx.x; c < cols; c += 256) sum += __expf(((rv && cm[c] != 0.f) ? ldf(xr + c) : -3.402823466e38f) - mx);
        sum = block_sum256(sum, red);
        const float inv = 1.f / sum;
        __syncthreads();       // in place: every read of the row is done
        for (int c = threadIdx.x; c < cols; c += 256) stf(yr + c, __expf(((rv && cm[c] != 0.f) ? ldf(xr + c) : -3.402823466e38f) - mx) * inv);
    }
}
// dx = filled ? 0 : y (dy - sum_c y dy)   (softmax backward followed by masked_fill's backward)
template <typename TY, typename TD>
__global__ __launch_bounds__(256) void softmax_masked_bwd_kernel(const TY* y, const TD* dy, TD* dx, const float* __restrict__ rowmask,
                                                                 const float* __restrict__ colmask, long rows, int cols, long rows_per_batch,
                                                                 int R) {
    __shared__ float red[4];
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const long b = row / rows_per_batch;
        const bool rv = rowmask[b * R + (row % R)] != 0.f;
        const float* cm = colmask + b * cols;
        const TY* yr = y + row * cols;
        const TD* gr = dy + row * cols;
        TD* dr = dx + row * cols;
        float dot = 0.f;
        for (int c = threadIdx.x; c < cols; c += 256) dot += ldf(yr + c) * ldf(gr + c);
        dot = block_sum256(dot, red);
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += 256) stf(dr + c, (rv && cm[c] != 0.f) ? ldf(yr + c) * (ldf(gr + c) - dot) : 0.f);
    }
}

extern "C" int mh_softmax_masked_fwd(const void* x, void* y, const float* rowmask, const float* colmask, int64_t batches, int h, int R,
                                     int cols, int dt_x, int dt_y, mh_stream s) {
    MH_REQUIRE(cols >= 1 && R >= 1 && h >= 1, "mh_softmax_masked_fwd: bad shape");
    const long rows = (long)batches * h * R;
    if (rows == 0) return MH_OK;
    dim3 grid((unsigned)min(rows, 65535L));
#define SMF_(TX, TY) hipLaunchKernelGGL((softmax_masked_fwd_kernel<TX, TY>), grid, dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, rowmask, colmask, rows, cols, (long)h * R, R)
    if (dt_x == MH_F32 && dt_y == MH_F32) { SMF_(float, float); }
    else if (dt_x == MH_F32) { SMF_(float, bf16_t); }
    else if (dt_y == MH_F32) { SMF_(bf16_t, float); }
    else { SMF_(bf16_t, bf16_t); }
#undef SMF_
    MH_LAUNCH_CHECK("mh_softmax_masked_fwd");
    return MH_OK;
}

extern "C" int mh_softmax_masked_bwd(const void* y, const void* dy, void* dx, const float* rowmask, const float* colmask, int64_t batches,
                                     int h, int R, int cols, int dt_y, int dt_d, mh_stream s) {
    MH_REQUIRE(cols >= 1 && R >= 1 && h >= 1, "mh_softmax_masked_bwd: bad shape");
    const long rows = (long)batches * h * R;
    if (rows == 0) return MH_OK;
    dim3 grid((unsigned)min(rows, 65535L));
#define SMB_(TY, TD) hipLaunchKernelGGL((softmax_masked_bwd_kernel<TY, TD>), grid, dim3(256), 0, (hipStream_t)s, (const TY*)y, (const TD*)dy, (TD*)dx, rowmask, colmask, rows, cols, (long)h * R, R)
    if (dt_y == MH_F32 && dt_d == MH_F32) { SMB_(float, float); }
    else if (dt_y == MH_F32) { SMB_(float, bf16_t); }
    else if (dt_d == MH_F32) { SMB_(bf16_t, float); }
    else { SMB_(bf16_t, bf16_t); }
#undef SMB_
    MH_LAUNCH_CHECK("mh_softmax_masked_bwd");
    return MH_OK;
}

extern "C" int mh_softmax_fwd(const void* x, void* y, int64_t rows, int cols, int64_t ldx, int64_t ldy, int dt_x,
                              int dt_y, mh_stream s) {
    MH_REQUIRE(cols >= 1, "mh_softmax_fwd: cols=%d", cols);
    if (rows == 0) return MH_OK;
    const bool wave = cols <= 1024;
    dim3 grid(wave ? mh_cdiv(rows, 4) : (unsigned)rows);
    if (cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && cols <= 12288 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0) {
#define SM_FV(TX, TY)                                                                                                     \
    if (wave) hipLaunchKernelGGL((softmax_fwd_vec_kernel<TX, TY, true, 4>), grid, dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, (long)rows, cols, (long)ldx, (long)ldy); \
    else hipLaunchKernelGGL((softmax_fwd_vec_kernel<TX, TY, false, 12>), grid, dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, (long)rows, cols, (long)ldx, (long)ldy)
        if (dt_x == MH_F32 && dt_y == MH_F32) { SM_FV(float, float); }
        else if (dt_x == MH_F32 && dt_y == MH_BF16) { SM_FV(float, bf16_t); }
        else if (dt_x == MH_BF16 && dt_y == MH_BF16) { SM_FV(bf16_t, bf16_t); }
        else { SM_FV(bf16_t, float); }
#undef SM_FV
        MH_LAUNCH_CHECK("mh_softmax_fwd");
        return MH_OK;
    }
#define SM_F(TX, TY)                                                                                                   \
    if (wave) hipLaunchKernelGGL((softmax_fwd_kernel<TX, TY, true>), grid, dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, (long)rows, cols, (long)ldx, (long)ldy); \
    else hipLaunchKernelGGL((softmax_fwd_kernel<TX, TY, false>), grid, dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, (long)rows, cols, (long)ldx, (long)ldy)
    if (dt_x == MH_F32 && dt_y == MH_F32) { SM_F(float, float); }
    else if (dt_x == MH_F32 && dt_y == MH_BF16) { SM_F(float, bf16_t); }
    else if (dt_x == MH_BF16 && dt_y == MH_BF16) { SM_F(bf16_t, bf16_t); }
    else { SM_F(bf16_t, float); }
#undef SM_F
    MH_LAUNCH_CHECK("mh_softmax_fwd");
    return MH_OK;
}

extern "C" int mh_softmax_bwd(const void* y, const void* dy, void* dx, int64_t rows, int cols, int64_t ldy, int64_t lddy,
                              int64_t lddx, int dt_y, int dt_dy, int dt_dx, mh_stream s) {
    MH_REQUIRE(cols >= 1, "mh_softmax_bwd: cols=%d", cols);
    MH_REQUIRE(dt_dy == dt_dx, "mh_softmax_bwd: dy and dx dtypes must match");
    if (rows == 0) return MH_OK;
    const bool wave = cols <= 1024;
    dim3 grid(wave ? mh_cdiv(rows, 4) : (unsigned)rows);
    if (cols % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && cols <= 12288 && ((uintptr_t)y & 15) == 0 &&
        ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0) {
#define SM_BV(TY, TD)                                                                                                     \
    if (wave) hipLaunchKernelGGL((softmax_bwd_vec_kernel<TY, TD, true, 4>), grid, dim3(256), 0, (hipStream_t)s, (const TY*)y, (const TD*)dy, (TD*)dx, (long)rows, cols, (long)ldy, (long)lddy, (long)lddx); \
    else hipLaunchKernelGGL((softmax_bwd_vec_kernel<TY, TD, false, 12>), grid, dim3(256), 0, (hipStream_t)s, (const TY*)y, (const TD*)dy, (TD*)dx, (long)rows, cols, (long)ldy, (long)lddy, (long)lddx)
        if (dt_y == MH_F32 && dt_dy == MH_F32) { SM_BV(float, float); }
        else if (dt_y == MH_F32 && dt_dy == MH_BF16) { SM_BV(float, bf16_t); }
        else if (dt_y == MH_BF16 && dt_dy == MH_BF16) { SM_BV(bf16_t, bf16_t); }
        else { SM_BV(bf16_t, float); }
#undef SM_BV
        MH_LAUNCH_CHECK("mh_softmax_bwd");
        return MH_OK;
    }
#define SM_B(TY, TD)                                                                                                   \
    if (wave) hipLaunchKernelGGL((softmax_bwd_kernel<TY, TD, TD, true>), grid, dim3(256), 0, (hipStream_t)s, (const TY*)y, (const TD*)dy, (TD*)dx, (long)rows, cols, (long)ldy, (long)lddy, (long)lddx); \
    else hipLaunchKernelGGL((softmax_bwd_kernel<TY, TD, TD, false>), grid, dim3(256), 0, (hipStream_t)s, (const TY*)y, (const TD*)dy, (TD*)dx, (long)rows, cols, (long)ldy, (long)lddy, (long)lddx)
    if (dt_y == MH_F32 && dt_dy == MH_F32) { SM_B(float, float); }
    else if (dt_y == MH_F32 && dt_dy == MH_BF16) { SM_B(float, bf16_t); }
    else if (dt_y == MH_BF16 && dt_dy == MH_BF16) { SM_B(bf16_t, bf16_t); }
    else { SM_B(bf16_t, float); }
#undef SM_B
    MH_LAUNCH_CHECK("mh_softmax_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------------------ L2 normalise
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const TX* __restrict__ x, TY* __restrict__ y, float* __restrict__ nrm,
                                                         int rows, int D, long x_rs, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TX* xr = x + (long)row * x_rs;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = ldf(xr + c); s += v * v; }
    const float n = fmaxf(sqrtf(wave_sum(s)), eps);
    for (int c = lane; c < D; c += 64) stf(y + (long)row * D + c, ldf(xr + c) / n);
    if (lane == 0) nrm[row] = n;
}

// dx = (dy - y*(y.dy)) / n
template <typename TY, typename TDY, typename TDX>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const TY* __restrict__ y, const float* __restrict__ nrm,
                                                         const TDY* __restrict__ dy, TDX* __restrict__ dx, int rows, int D,
                                                         long dx_rs, int acc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float dot = 0.f;
    for (int c = lane; c < D; c += 64) dot += ldf(y + (long)row * D + c) * ldf(dy + (long)row * D + c);
    dot = wave_sum(dot);
    const float inv = 1.f / nrm[row];
    for (int c = lane; c < D; c += 64) {
        float r = (ldf(dy + (long)row * D + c) - ldf(y + (long)row * D + c) * dot) * inv;
        TDX* p = dx + (long)row * dx_rs + c;
        if (acc) r += ldf(p);
        stf(p, r);
    }
}

extern "C" int mh_l2norm_fwd(const void* x, void* y, float* nrm, int rows, int D, int64_t x_rs, float eps, int dt_x,
                             int dt_y, mh_stream s) {
    if (rows == 0) return MH_OK;
    dim3 grid(mh_cdiv(rows, 4));
#define L2_F(TX, TY) hipLaunchKernelGGL((l2norm_fwd_kernel<TX, TY>), grid, dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, nrm, rows, D, (long)x_rs, eps)
    if (dt_x == MH_F32 && dt_y == MH_F32) L2_F(float, float);
    else if (dt_x == MH_F32 && dt_y == MH_BF16) L2_F(float, bf16_t);
    else if (dt_x == MH_BF16 && dt_y == MH_BF16) L2_F(bf16_t, bf16_t);
    else L2_F(bf16_t, float);
#undef L2_F
    MH_LAUNCH_CHECK("mh_l2norm_fwd");
    return MH_OK;
}

extern "C" int mh_l2norm_bwd(const void* y, const float* nrm, const void* dy, void* dx, int rows, int D, int64_t dx_rs,
                             float eps, int dt_y, int dt_dy, int dt_dx, int acc, mh_stream s) {
    (void)eps;
    MH_REQUIRE(dt_y == dt_dy, "mh_l2norm_bwd: y and dy dtypes must match");
    if (rows == 0) return MH_OK;
    dim3 grid(mh_cdiv(rows, 4));
#define L2_B(TY, TDX) hipLaunchKernelGGL((l2norm_bwd_kernel<TY, TY, TDX>), grid, dim3(256), 0, (hipStream_t)s, (const TY*)y, nrm, (const TY*)dy, (TDX*)dx, rows, D, (long)dx_rs, acc)
    if (dt_y == MH_F32 && dt_dx == MH_F32) L2_B(float, float);
    else if (dt_y == MH_F32 && dt_dx == MH_BF16) L2_B(float, bf16_t);
    else if (dt_y == MH_BF16 && dt_dx == MH_BF16) L2_B(bf16_t, bf16_t);
    else L2_B(bf16_t, float);
#undef L2_B
    MH_LAUNCH_CHECK("mh_l2norm_bwd");
    return MH_OK;
}
