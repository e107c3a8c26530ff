// Loss kernels: CLIP / InfoNCE cross-entropy over logit rows, masked MSE (retention), style KL, symmetric
// KL between prototype soft assignments (losses/mirror_loss.py:74-135, losses/info_nce.py:144-164), plus
// the step glue of train_mirror.py (prototype row-normalise :1133-1136, logit_scale clamp :1254-1255, Adam :1230).
// Scalars accumulate with f32 atomics into caller-zeroed device words: no host synchronisation anywhere.
#include "common.h"

// ------------------------------------------------------------------ cross-entropy rows
// logits L = (scale*scale_mul) * G; row r has label (label_off + r). One wave per row.
__global__ __launch_bounds__(256) void ce_rows_fwd_kernel(const float* __restrict__ G, long ldg, const float* __restrict__ scale,
                                                          float scale_mul, int R, int C, int label_off, float coef,
                                                          float* __restrict__ loss_rows, float* __restrict__ lse,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float s = (scale ? scale[0] : 1.f) * scale_mul;
    const float* g = G + (long)r * ldg;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, s * g[c]);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += __expf(s * g[c] - m);
    sum = wave_sum(sum);
    if (lane == 0) {
        const float l = m + __logf(sum);
        const float loss = l - s * g[label_off + r];
        lse[r] = l;
        if (loss_rows) loss_rows[r] = coef * loss;
        if (out) atomicAdd(out, coef * loss);
    }
}

// dG[r,c] = w_r * s * (softmax(L)[r,c] - [c == label]);  dscale += scale_mul * sum_rc w_r * (softmax - onehot) * G
// w_r = gcoef * (g_per_row ? g[r] : g[0])
__global__ __launch_bounds__(256) void ce_rows_bwd_kernel(const float* __restrict__ G, long ldg, const float* __restrict__ scale,
                                                          float scale_mul, const float* __restrict__ lse,
                                                          const float* __restrict__ g, int g_per_row, float gcoef,
                                                          float* __restrict__ dG, float* __restrict__ dscale, int R, int C,
                                                          int label_off) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float s = (scale ? scale[0] : 1.f) * scale_mul;
    const float w = gcoef * (g_per_row ? g[r] : g[0]);
    const float* gr = G + (long)r * ldg;
    const float l = lse[r];
    float ds = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float p = __expf(s * gr[c] - l) - (c == label_off + r ? 1.f : 0.f);
        dG[(long)r * C + c] = w * s * p;
        ds += w * p * gr[c];
    }
    ds = wave_sum(ds);
    if (lane == 0 && dscale) atomicAdd(dscale, ds * scale_mul);
}

extern "C" int mh_ce_rows_fwd(const float* G, int64_t ldg, const float* scale, float scale_mul, int R, int C, int label_off,
                              float coef, float* loss_rows, float* lse, float* out, mh_stream s) {
    MH_REQUIRE(label_off >= 0 && label_off + R <= C, "mh_ce_rows_fwd: labels [%d,%d) outside %d columns", label_off, label_off + R, C);
    if (R == 0) return MH_OK;
    hipLaunchKernelGGL(ce_rows_fwd_kernel, dim3(mh_cdiv(R, 4)), dim3(256), 0, (hipStream_t)s, G, (long)ldg, scale, scale_mul, R, C, label_off, coef, loss_rows, lse, out);
    MH_LAUNCH_CHECK("mh_ce_rows_fwd");
    return MH_OK;
}

extern "C" int mh_ce_rows_bwd(const float* G, int64_t ldg, const float* scale, float scale_mul, const float* lse, const float* g,
                              int g_per_row, float gcoef, float* dG, float* dscale, int R, int C, int label_off, mh_stream s) {
    if (R == 0) return MH_OK;
    hipLaunchKernelGGL(ce_rows_bwd_kernel, dim3(mh_cdiv(R, 4)), dim3(256), 0, (hipStream_t)s, G, (long)ldg, scale, scale_mul, lse, g, g_per_row, gcoef, dG, dscale, R, C, label_off);
    MH_LAUNCH_CHECK("mh_ce_rows_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------ masked MSE (retention losses)
// acc[0] += sum_r mask[r] * (1/D) sum_d (p-t)^2 ; acc[1] += sum_r mask[r]; one wave per row, grid-stride.
// pred is [rows, D] contiguous; row r of the target sits at tgt + (r / rpb) * tgt_bs + (r % rpb) * D (a row window of a
// larger [B, T, D] buffer: the WSI target is encoder_output[:, 1:], models/mirror.py:700).
template <typename TP, typename TT, bool VEC>
__global__ __launch_bounds__(256) void mse_masked_fwd_kernel(const TP* __restrict__ pred, const TT* __restrict__ tgt,
                                                             const float* __restrict__ mask, float* __restrict__ acc, long rows, int D,
                                                             long rpb, long tgt_bs) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nw = (long)gridDim.x * 4;
    float num = 0.f, den = 0.f;
    // four rows in flight per wave (rows r, r + nw, r + 2 nw, r + 3 nw): the mask -> data dependency of one row at a time
    // left the kernel latency bound (1.3 TB/s); few blocks, so the two same-address atomics per block stay cheap
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += 4 * nw) {
        float mk[4], s[4] = {0.f, 0.f, 0.f, 0.f};
        const TP* pr[4];
        const TT* tr[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long ru = r + u * nw;
            mk[u] = ru < rows ? mask[ru] : 0.f;
            pr[u] = pred + ru * D;
            tr[u] = tgt + (ru / rpb) * tgt_bs + (ru % rpb) * D;
            den += mk[u];   // every lane carries it; only lane 0's copy is used below
        }
        if (VEC) {
            for (int c = 4 * lane; c < D; c += 256) {
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (mk[u] != 0.f) {
                        const f4_t d = ld4(pr[u] + c) - ld4(tr[u] + c);
                        s[u] += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
                    }
            }
        } else {
            for (int c = lane; c < D; c += 64) {
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (mk[u] != 0.f) {
                        const float d = ldf(pr[u] + c) - ldf(tr[u] + c);
                        s[u] += d * d;
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) num += mk[u] * s[u] / D;
    }
    num = block_sum256(num, red);
    den = block_sum256(lane == 0 ? den : 0.f, red);
    if (threadIdx.x == 0) { atomicAdd(acc, num); atomicAdd(acc + 1, den); }
}

// D == 1 (RNA retention loss: the channel axis is the masked axis): a flat reduction, one element per thread iteration
template <typename TP, typename TT>
__global__ __launch_bounds__(256) void mse_masked_fwd_flat_kernel(const TP* __restrict__ pred, const TT* __restrict__ tgt,
                                                                  const float* __restrict__ mask, float* __restrict__ acc, long n) {
    __shared__ float red[4];
    float num = 0.f, den = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float mk = mask[i], d = ldf(pred + i) - ldf(tgt + i);
        num += mk * d * d;
        den += mk;
    }
    num = block_sum256(num, red);
    den = block_sum256(den, red);
    if (threadIdx.x == 0) { atomicAdd(acc, num); atomicAdd(acc + 1, den); }
}

// dpred = k mask (p - t) in TD; dtgt = -dpred in TT when the caller wants it materialised (nullptr: not written)
template <typename TP, typename TT, typename TD>
__global__ __launch_bounds__(256) void mse_masked_bwd_kernel(const TP* __restrict__ pred, const TT* __restrict__ tgt,
                                                             const float* __restrict__ mask, const float* __restrict__ acc,
                                                             const float* __restrict__ g, TD* __restrict__ dpred, TT* __restrict__ dtgt,
                                                             long rows, int D, long rpb, long tgt_bs, float gmul) {
    const float k = g[0] * gmul * 2.f / ((float)D * acc[1]);
    const long total = rows * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / D;
        const float mk = mask[r];
        const float d = mk != 0.f ? k * mk * (ldf(pred + i) - ldf(tgt + (r / rpb) * tgt_bs + (r % rpb) * D + (i - r * D))) : 0.f;
        stf(dpred + i, d);
        if (dtgt) stf(dtgt + i, -d);
    }
}

// D % 4 == 0: one wave per row, quads; rows the mask drops are written as zeros without reading pred / tgt
template <typename TP, typename TT, typename TD>
__global__ __launch_bounds__(256) void mse_masked_bwd_vec_kernel(const TP* __restrict__ pred, const TT* __restrict__ tgt,
                                                                 const float* __restrict__ mask, const float* __restrict__ acc,
                                                                 const float* __restrict__ g, TD* __restrict__ dpred, TT* __restrict__ dtgt,
                                                                 long rows, int D, long rpb, long tgt_bs, float gmul) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float k = g[0] * gmul * 2.f / ((float)D * acc[1]);
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        const float mk = mask[r];
        const float km = k * mk;
        const TP* pr = pred + r * D;
        const TT* tr = tgt + (r / rpb) * tgt_bs + (r % rpb) * D;
        for (int c = 4 * lane; c < D; c += 256) {
            f4_t d = {0.f, 0.f, 0.f, 0.f};
            if (mk != 0.f) d = (ld4(pr + c) - ld4(tr + c)) * km;
            st4(dpred + r * D + c, d);
            if (dtgt) st4(dtgt + r * D + c, -d);
        }
    }
}

// the same pass with the column sums of dpred beside it (round 5): cs[block][D] gets the sums over the rows this block walked (plain
// stores, every block writes its whole row — nothing to zero), which mh_colsum over the [blocks, D] table folds into the bias gradient of
// the Linear that produced pred: the colsum pass over all of dpred (rows x D bf16 read again, 27 us at 65536 x 1024) is not launched.
// The sums are of the values as stored (rounded to TD), like mh_colsum over dpred would see them.  NCH = D / 256.
template <typename TP, typename TT, typename TD, int NCH>
__global__ __launch_bounds__(256) void mse_masked_bwd_cs_kernel(const TP* __restrict__ pred, const TT* __restrict__ tgt,
                                                                const float* __restrict__ mask, const float* __restrict__ acc,
                                                                const float* __restrict__ g, TD* __restrict__ dpred, TT* __restrict__ dtgt,
                                                                long rows, long rpb, long tgt_bs, float gmul, float* __restrict__ cs) {
    constexpr int D = 256 * NCH;
    constexpr int RIF = NCH <= 2 ? 4 : 2;       // rows in flight per wave (48 B per lane and row at NCH = 2)
    __shared__ f4_t red[3][NCH][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float k = g[0] * gmul * 2.f / ((float)D * acc[1]);
    f4_t sum[NCH];
#pragma unroll
    for (int j = 0; j < NCH; j++) sum[j] = (f4_t){0.f, 0.f, 0.f, 0.f};
    for (long r0 = (long)blockIdx.x * (4 * RIF) + wave; r0 < rows; r0 += (long)gridDim.x * (4 * RIF)) {
        float mk[RIF];
        f4_t p[RIF][NCH], t[RIF][NCH];
        bool ok[RIF];
#pragma unroll
        for (int u = 0; u < RIF; u++) {           // RIF rows per wave in flight
            const long r = r0 + 4 * u;
            ok[u] = r < rows;
            mk[u] = ok[u] ? mask[r] : 0.f;
            if (mk[u] != 0.f) {
                const TP* pr = pred + r * D;
                const TT* tr = tgt + (r / rpb) * tgt_bs + (r % rpb) * D;
#pragma unroll
                for (int j = 0; j < NCH; j++) {
                    p[u][j] = ld4(pr + 256 * j + 4 * lane);
                    t[u][j] = ld4(tr + 256 * j + 4 * lane);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RIF; u++) {
            if (!ok[u]) continue;
            const long r = r0 + 4 * u;
            const float km = k * mk[u];
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                f4_t d = {0.f, 0.f, 0.f, 0.f};
                if (mk[u] != 0.f) {
                    d = (p[u][j] - t[u][j]) * km;
                    if constexpr (sizeof(TD) == 2) {
#pragma unroll
                        for (int e = 0; e < 4; e++) d[e] = bf2f(f2bf(d[e]));
                    }
                    sum[j] += d;
                }
                st4(dpred + r * D + 256 * j + 4 * lane, d);
                if (dtgt) st4(dtgt + r * D + 256 * j + 4 * lane, -d);
            }
        }
    }
    if (wave) {
#pragma unroll
        for (int j = 0; j < NCH; j++) red[wave - 1][j][lane] = sum[j];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < NCH; j++)
            *reinterpret_cast<f4_t*>(cs + (long)blockIdx.x * D + 256 * j + 4 * lane) = sum[j] + red[0][j][lane] + red[1][j][lane] + red[2][j][lane];
    }
}

extern "C" int mh_mse_masked_fwd(const void* pred, const void* tgt, const float* mask, float* acc, int64_t rows, int D,
                                 int64_t rows_per_batch, int64_t tgt_bs, int dt_p, int dt_t, mh_stream s) {
    if (rows == 0) return MH_OK;
    MH_REQUIRE(rows_per_batch > 0, "mh_mse_masked_fwd: rows_per_batch must be positive");
    if (D == 1 && tgt_bs == rows_per_batch) {       // contiguous [rows] vectors
        dim3 gf((unsigned)min((long)mh_cdiv(rows, 256), 64L));
#define MSEFL(TP, TT) hipLaunchKernelGGL((mse_masked_fwd_flat_kernel<TP, TT>), gf, dim3(256), 0, (hipStream_t)s, (const TP*)pred, (const TT*)tgt, mask, acc, (long)rows)
        if (dt_p == MH_F32 && dt_t == MH_F32) { MSEFL(float, float); }
        else if (dt_p == MH_BF16 && dt_t == MH_BF16) { MSEFL(bf16_t, bf16_t); }
        else if (dt_p == MH_F32) { MSEFL(float, bf16_t); }
        else { MSEFL(bf16_t, float); }
#undef MSEFL
        MH_LAUNCH_CHECK("mh_mse_masked_fwd");
        return MH_OK;
    }
    dim3 grid((unsigned)min((long)mh_cdiv(rows, 16), 1024L));
    const bool vec = D % 4 == 0 && tgt_bs % 4 == 0 && mh_quad_ok(pred, mh_dt_size(dt_p)) && mh_quad_ok(tgt, mh_dt_size(dt_t));
#define MSEF(TP, TT)                                                                                                              \
    if (vec) hipLaunchKernelGGL((mse_masked_fwd_kernel<TP, TT, true>), grid, dim3(256), 0, (hipStream_t)s, (const TP*)pred, (const TT*)tgt, mask, acc, (long)rows, D, (long)rows_per_batch, (long)tgt_bs); \
    else hipLaunchKernelGGL((mse_masked_fwd_kernel<TP, TT, false>), grid, dim3(256), 0, (hipStream_t)s, (const TP*)pred, (const TT*)tgt, mask, acc, (long)rows, D, (long)rows_per_batch, (long)tgt_bs)
    if (dt_p == MH_F32 && dt_t == MH_F32) { MSEF(float, float); }
    else if (dt_p == MH_BF16 && dt_t == MH_BF16) { MSEF(bf16_t, bf16_t); }
    else if (dt_p == MH_F32) { MSEF(float, bf16_t); }
    else { MSEF(bf16_t, float); }
#undef MSEF
    MH_LAUNCH_CHECK("mh_mse_masked_fwd");
    return MH_OK;
}

extern "C" int mh_mse_masked_bwd(const void* pred, const void* tgt, const float* mask, const float* acc, const float* g,
                                 float gmul, void* dpred, void* dtgt, int64_t rows, int D, int64_t rows_per_batch, int64_t tgt_bs,
                                 int dt_p, int dt_t, int dt_dp, float* colsum_ws, int cs_blocks, mh_stream s) {
    if (rows == 0) return MH_OK;
    MH_REQUIRE(rows_per_batch > 0, "mh_mse_masked_bwd: rows_per_batch must be positive");
    const bool vec = D % 4 == 0 && tgt_bs % 4 == 0 && mh_quad_ok(pred, mh_dt_size(dt_p)) && mh_quad_ok(tgt, mh_dt_size(dt_t)) &&
                     mh_quad_ok(dpred, mh_dt_size(dt_dp)) && (!dtgt || mh_quad_ok(dtgt, mh_dt_size(dt_t)));
    if (colsum_ws) {
        MH_REQUIRE(vec && D % 256 == 0 && D <= 1024 && cs_blocks >= 1 && cs_blocks <= 4096 && dt_p == MH_BF16 && dt_t == MH_F32 && dt_dp == MH_BF16 &&
                       (((uintptr_t)colsum_ws) & 15) == 0,
                   "mh_mse_masked_bwd: colsum_ws needs bf16 pred / f32 target / bf16 dpred on quads and D in {256, 512, 768, 1024} (got D=%d, dtypes %d %d %d)", D, dt_p, dt_t, dt_dp);
#define MSECS(NCH) hipLaunchKernelGGL((mse_masked_bwd_cs_kernel<bf16_t, float, bf16_t, NCH>), dim3(cs_blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)pred, (const float*)tgt, mask, acc, g, (bf16_t*)dpred, (float*)dtgt, (long)rows, (long)rows_per_batch, (long)tgt_bs, gmul, colsum_ws)
        if (D == 256) MSECS(1); else if (D == 512) MSECS(2); else if (D == 768) MSECS(3); else MSECS(4);
#undef MSECS
        MH_LAUNCH_CHECK("mh_mse_masked_bwd");
        return MH_OK;
    }
    dim3 grid((unsigned)min((long)mh_cdiv(rows * D, 256), 16384L)), gv((unsigned)min((long)mh_cdiv(rows, 4), 16384L));
#define MSEB3(TP, TT, TD)                                                                                                         \
    if (vec) hipLaunchKernelGGL((mse_masked_bwd_vec_kernel<TP, TT, TD>), gv, dim3(256), 0, (hipStream_t)s, (const TP*)pred, (const TT*)tgt, mask, acc, g, (TD*)dpred, (TT*)dtgt, (long)rows, D, (long)rows_per_batch, (long)tgt_bs, gmul); \
    else hipLaunchKernelGGL((mse_masked_bwd_kernel<TP, TT, TD>), grid, dim3(256), 0, (hipStream_t)s, (const TP*)pred, (const TT*)tgt, mask, acc, g, (TD*)dpred, (TT*)dtgt, (long)rows, D, (long)rows_per_batch, (long)tgt_bs, gmul)
#define MSEB2(TP, TT)                                 \
    if (dt_dp == MH_F32) { MSEB3(TP, TT, float); }    \
    else { MSEB3(TP, TT, bf16_t); }
    if (dt_p == MH_F32 && dt_t == MH_F32) { MSEB2(float, float) }
    else if (dt_p == MH_BF16 && dt_t == MH_BF16) { MSEB2(bf16_t, bf16_t) }
    else if (dt_p == MH_F32) { MSEB2(float, bf16_t) }
    else { MSEB2(bf16_t, float) }
#undef MSEB2
#undef MSEB3
    MH_LAUNCH_CHECK("mh_mse_masked_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------ weighted sum of the loss terms (losses/mirror_loss.py:121-127)
// out[0] = sum_i w[i] * t_i[0] over up to 6 separate 0-d tensors; the backward is dt[i] = w[i] * g[0].  One launch each
// instead of ~10 scalar torch kernels forward and ~10 backward.
struct WSum6 { const float* t[6]; float w[6]; int n; };
__global__ void wsum6_kernel(WSum6 a, float* out) {
    float s = 0.f;
    for (int i = 0; i < a.n; i++) s += a.w[i] * a.t[i][0];
    out[0] = s;
}
__global__ void wscale6_kernel(const float* g, WSum6 a, float* out) {
    if ((int)threadIdx.x < a.n) out[threadIdx.x] = a.w[threadIdx.x] * g[0];
}
extern "C" int mh_weighted_sum(const float* t0, const float* t1, const float* t2, const float* t3, const float* t4, const float* t5,
                               float w0, float w1, float w2, float w3, float w4, float w5, int n, float* out, mh_stream s) {
    MH_REQUIRE(n >= 1 && n <= 6, "mh_weighted_sum: n=%d (1..6 terms)", n);
    WSum6 a = {{t0, t1, t2, t3, t4, t5}, {w0, w1, w2, w3, w4, w5}, n};
    hipLaunchKernelGGL(wsum6_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, a, out);
    MH_LAUNCH_CHECK("mh_weighted_sum");
    return MH_OK;
}
extern "C" int mh_weighted_sum_bwd(const float* g, float w0, float w1, float w2, float w3, float w4, float w5, int n, float* dterms,
                                   mh_stream s) {
    MH_REQUIRE(n >= 1 && n <= 6, "mh_weighted_sum_bwd: n=%d (1..6 terms)", n);
    WSum6 a = {{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, {w0, w1, w2, w3, w4, w5}, n};
    hipLaunchKernelGGL(wscale6_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, g, a, dterms);
    MH_LAUNCH_CHECK("mh_weighted_sum_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------ style KL to N(0, I)
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ ls, float* __restrict__ out,
                                                     long n, float coef) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        s += __expf(ls[i]) + mu[i] * mu[i] - 1.f - ls[i];
    s = block_sum256(s, red);
    if (threadIdx.x == 0) atomicAdd(out, coef * s);
}
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ ls, const float* __restrict__ g,
                                                     float* __restrict__ dmu, float* __restrict__ dls, long n, float coef) {
    const float k = g[0] * coef;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        dmu[i] = k * 2.f * mu[i];
        dls[i] = k * (__expf(ls[i]) - 1.f);
    }
}
extern "C" int mh_kl_fwd(const float* mu, const float* ls, float* out, int64_t n, float coef, mh_stream s) {
    if (n == 0) return MH_OK;
    hipLaunchKernelGGL(kl_fwd_kernel, dim3((unsigned)min((long)mh_cdiv(n, 256), 1024L)), dim3(256), 0, (hipStream_t)s, mu, ls, out, (long)n, coef);
    MH_LAUNCH_CHECK("mh_kl_fwd");
    return MH_OK;
}
extern "C" int mh_kl_bwd(const float* mu, const float* ls, const float* g, float* dmu, float* dls, int64_t n, float coef, mh_stream s) {
    if (n == 0) return MH_OK;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3((unsigned)min((long)mh_cdiv(n, 256), 1024L)), dim3(256), 0, (hipStream_t)s, mu, ls, g, dmu, dls, (long)n, coef);
    MH_LAUNCH_CHECK("mh_kl_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------ symmetric KL of two softmaxes (cluster loss)
// per row: term = sum_k (p_r - p_w)(log p_r - log p_w); one 256-thread block per row
struct RowStats { float mw, sw, mr, sr; };
__device__ __forceinline__ RowStats row_stats(const float* w, const float* r, int P, float* red) {
    float mw = -INFINITY, mr = -INFINITY;
    for (int k = threadIdx.x; k < P; k += 256) { mw = fmaxf(mw, w[k]); mr = fmaxf(mr, r[k]); }
    mw = block_max256(mw, red);
    mr = block_max256(mr, red);
    float sw = 0.f, sr = 0.f;
    for (int k = threadIdx.x; k < P; k += 256) { sw += __expf(w[k] - mw); sr += __expf(r[k] - mr); }
    sw = block_sum256(sw, red);
    sr = block_sum256(sr, red);
    return {mw, __logf(sw), mr, __logf(sr)};
}

__global__ __launch_bounds__(256) void symkl_fwd_kernel(const float* __restrict__ w, const float* __restrict__ r, float* __restrict__ out,
                                                        int P, float coef) {
    __shared__ float red[4];
    const float* wr = w + (long)blockIdx.x * P;
    const float* rr = r + (long)blockIdx.x * P;
    const RowStats st = row_stats(wr, rr, P, red);
    float s = 0.f;
    for (int k = threadIdx.x; k < P; k += 256) {
        const float lw = wr[k] - st.mw - st.sw, lr = rr[k] - st.mr - st.sr;
        s += (__expf(lr) - __expf(lw)) * (lr - lw);
    }
    s = block_sum256(s, red);
    if (threadIdx.x == 0) atomicAdd(out, coef * s);
}

// d term / d r_k = p_r,k (d_k - E_r[d]) + q_k ;  d term / d w_k = -p_w,k (d_k - E_w[d]) - q_k ;  d = lr - lw, q = p_r - p_w
__global__ __launch_bounds__(256) void symkl_bwd_kernel(const float* __restrict__ w, const float* __restrict__ r, const float* __restrict__ g,
                                                        float* __restrict__ dw, float* __restrict__ dr, int P, float coef) {
    __shared__ float red[4];
    const long base = (long)blockIdx.x * P;
    const float* wr = w + base;
    const float* rr = r + base;
    const RowStats st = row_stats(wr, rr, P, red);
    float er = 0.f, ew = 0.f;
    for (int k = threadIdx.x; k < P; k += 256) {
        const float lw = wr[k] - st.mw - st.sw, lr = rr[k] - st.mr - st.sr;
        er += __expf(lr) * (lr - lw);
        ew += __expf(lw) * (lr - lw);
    }
    er = block_sum256(er, red);
    ew = block_sum256(ew, red);
    const float c = g[0] * coef;
    for (int k = threadIdx.x; k < P; k += 256) {
        const float lw = wr[k] - st.mw - st.sw, lr = rr[k] - st.mr - st.sr;
        const float pw = __expf(lw), pr = __expf(lr), d = lr - lw, q = pr - pw;
        dr[base + k] = c * (pr * (d - er) + q);
        dw[base + k] = c * (-pw * (d - ew) - q);
    }
}

extern "C" int mh_symkl_fwd(const float* w, const float* r, float* out, int B, int P, float coef, mh_stream s) {
    if (B == 0) return MH_OK;
    hipLaunchKernelGGL(symkl_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)s, w, r, out, P, coef);
    MH_LAUNCH_CHECK("mh_symkl_fwd");
    return MH_OK;
}
extern "C" int mh_symkl_bwd(const float* w, const float* r, const float* g, float* dw, float* dr, int B, int P, float coef, mh_stream s) {
    if (B == 0) return MH_OK;
    hipLaunchKernelGGL(symkl_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)s, w, r, g, dw, dr, P, coef);
    MH_LAUNCH_CHECK("mh_symkl_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------ step glue
__global__ __launch_bounds__(256) void rownorm_kernel(float* w, bf16_t* shadow, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* wr = w + (long)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += wr[c] * wr[c];
    const float n = fmaxf(sqrtf(wave_sum(s)), eps);
    for (int c = lane; c < D; c += 64) {
        const float v = wr[c] / n;
        wr[c] = v;
        if (shadow) shadow[(long)row * D + c] = f2bf(v);      // the bf16 copy the GEMMs read: no cast launch behind this one
    }
}
extern "C" int mh_rownorm_(float* w, void* shadow_bf16, int rows, int D, float eps, mh_stream s) {
    if (rows == 0) return MH_OK;
    hipLaunchKernelGGL(rownorm_kernel, dim3(mh_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)s, w, (bf16_t*)shadow_bf16, rows, D, eps);
    MH_LAUNCH_CHECK("mh_rownorm_");
    return MH_OK;
}

__global__ void clamp_kernel(float* x, long n, float lo, float hi) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = fminf(fmaxf(x[i], lo), hi);
}
extern "C" int mh_clamp_(float* x, int64_t n, float lo, float hi, mh_stream s) {
    if (n == 0) return MH_OK;
    hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)min((long)mh_cdiv(n, 256), 1024L)), dim3(256), 0, (hipStream_t)s, x, (long)n, lo, hi);
    MH_LAUNCH_CHECK("mh_clamp_");
    return MH_OK;
}

// torch.optim.Adam semantics (no weight decay, no amsgrad): 4 floats per thread, 16-B accesses (HBM-bound: 28 B/param)
__device__ __forceinline__ void adam_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ shadow, long n, float lr, float b1,
                                                   float b2, float eps, float bc1, float bc2, float gscale,
                                                   const float* __restrict__ state, long clamp_i, float clamp_lo, float clamp_hi,
                                                   long hole_lo4, long hole_hi4) {
    // quads [hole_lo4, hole_hi4) are left alone: a range another launch of the same step has already updated (the RNA encoder's
    // parameters, whose gradients are complete 2 ms before the step's last one: TrainEngine's early update)
    if (state) {   // device-resident step state {t, 1 - b1^t, 1 - b2^t, lr, clip}: nothing step-dependent is a launch argument
        bc1 = state[1];
        bc2 = state[2];
        lr = state[3];
        gscale *= state[4];     // gradient-clipping factor written by mh_grad_clip (1 when clipping is off)
    }
    const float step = lr / bc1;
    const float isq = rsqrtf(bc2);
    const long n4 = n / 4;
    const long hole = hole_hi4 - hole_lo4, live4 = n4 - hole;
    for (long q0 = (long)blockIdx.x * 256 + threadIdx.x; q0 < live4; q0 += (long)gridDim.x * 256) {
        const long q = q0 < hole_lo4 ? q0 : q0 + hole;       // the live quads are numbered densely: no idle threads over the hole
        float4 pp = reinterpret_cast<float4*>(p)[q];
        const float4 gg = reinterpret_cast<const float4*>(g)[q];
        float4 mm = reinterpret_cast<float4*>(m)[q];
        float4 vv = reinterpret_cast<float4*>(v)[q];
        float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float gr = ga[e] * gscale;
            ma[e] = b1 * ma[e] + (1.f - b1) * gr;
            va[e] = b2 * va[e] + (1.f - b2) * gr * gr;
            pa[e] -= step * ma[e] / (sqrtf(va[e]) * isq + eps);
        }
        // one element (logit_scale, train_mirror.py:1255) is clamped right behind its update: master and shadow get the clamped value
        if ((clamp_i >> 2) == q && clamp_i >= 0) pa[clamp_i & 3] = fminf(fmaxf(pa[clamp_i & 3], clamp_lo), clamp_hi);
        reinterpret_cast<float4*>(p)[q] = pp;
        reinterpret_cast<float4*>(m)[q] = mm;
        reinterpret_cast<float4*>(v)[q] = vv;
        if (shadow) {
            uint2 sh;
            sh.x = pack_bf2(pa[0], pa[1]);
            sh.y = pack_bf2(pa[2], pa[3]);
            reinterpret_cast<uint2*>(shadow)[q] = sh;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = n4 * 4 + threadIdx.x;
        const float gr = g[i] * gscale;
        m[i] = b1 * m[i] + (1.f - b1) * gr;
        v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
        float pn = p[i] - step * m[i] / (sqrtf(v[i]) * isq + eps);
        if (i == clamp_i) pn = fminf(fmaxf(pn, clamp_lo), clamp_hi);
        p[i] = pn;
        if (shadow) shadow[i] = f2bf(pn);
    }
}

#define ADAM_ARGS_ float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, bf16_t *__restrict__ shadow, \
                   long n, float lr, float b1, float b2, float eps, float bc1, float bc2, float gscale, const float *__restrict__ state, \
                   long clamp_i, float clamp_lo, float clamp_hi, long hole_lo4, long hole_hi4
// the launch that ENDS a step (the whole arena, or everything around the hole): profiling tools cut a trace into steps at this name
__global__ __launch_bounds__(256) void adam_kernel(ADAM_ARGS_) {
    adam_body(p, g, m, v, shadow, n, lr, b1, b2, eps, bc1, bc2, gscale, state, clamp_i, clamp_lo, clamp_hi, hole_lo4, hole_hi4);
}
// the early launch of a two-launch step (a sub-range, beside the backward): same arithmetic under another name
__global__ __launch_bounds__(256) void adam_range_kernel(ADAM_ARGS_) {
    adam_body(p, g, m, v, shadow, n, lr, b1, b2, eps, bc1, bc2, gscale, state, clamp_i, clamp_lo, clamp_hi, hole_lo4, hole_hi4);
}
#undef ADAM_ARGS_

// state = {t, 1 - b1^t, 1 - b2^t, lr, clip, |g|}: t += 1 and the bias corrections are refreshed on the device, so a
// captured HIP graph of the whole step replays with the right Adam step every time
__global__ void adam_tick_kernel(float* state, float b1, float b2, long long* counter, long long counter_add) {
    if (counter) *counter += counter_add;      // the dropout streams' device-side base (functional.dropout_step_end) rides along
    if (!state) return;
    const float t = state[0] + 1.f;
    state[0] = t;
    state[1] = 1.f - powf(b1, t);
    state[2] = 1.f - powf(b2, t);
}

// ---- gradient clipping by global L2 norm (timm's clip_grad "norm" mode, train_mirror.py:1206-1230): the factor stays on
// the device (state[4]) and mh_adam multiplies it into its gradient scale — no host round trip, graph-capturable
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ acc) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = n / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[q];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
    s = block_sum256(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}
__global__ void clip_factor_kernel(const float* acc, float gscale, float max_norm, float* state) {
    const float norm = sqrtf(acc[0]) * gscale;
    state[5] = norm;
    state[4] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
}

extern "C" int mh_grad_clip(const float* g, int64_t n, float grad_scale, float max_norm, float* scratch1, float* dev_state,
                            mh_stream s) {
    MH_REQUIRE(((uintptr_t)g & 15) == 0 && dev_state && scratch1, "mh_grad_clip: bad arguments");
    if (hipMemsetAsync(scratch1, 0, sizeof(float), (hipStream_t)s) != hipSuccess) { mh_set_error("mh_grad_clip: memset failed"); return MH_EHIP; }
    if (n > 0) hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)min((long)mh_cdiv(mh_cdiv(n, 4), 256), 2048L)), dim3(256), 0, (hipStream_t)s, g, (long)n, scratch1);
    hipLaunchKernelGGL(clip_factor_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, (const float*)scratch1, grad_scale, max_norm, dev_state);
    MH_LAUNCH_CHECK("mh_grad_clip");
    return MH_OK;
}

extern "C" int mh_adam(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, float lr, float b1, float b2,
                       float eps, float bc1, float bc2, float gscale, float* dev_state, int64_t clamp_index, float clamp_lo,
                       float clamp_hi, int64_t* counter, int64_t counter_add, int tick, int64_t hole_lo, int64_t hole_hi, mh_stream s) {
    if (n == 0) return MH_OK;
    MH_REQUIRE(((uintptr_t)p & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)m & 15) == 0 && ((uintptr_t)v & 15) == 0 &&
                   ((uintptr_t)shadow & 7) == 0, "mh_adam: buffers must be 16-byte aligned");
    MH_REQUIRE(clamp_index < n, "mh_adam: clamp_index %ld outside the %ld parameters", (long)clamp_index, (long)n);
    MH_REQUIRE(hole_lo >= 0 && hole_lo <= hole_hi && hole_hi <= n && hole_lo % 4 == 0 && (hole_hi % 4 == 0 || hole_hi == hole_lo) &&
                   (clamp_index < hole_lo || clamp_index >= hole_hi || hole_lo == hole_hi),
               "mh_adam: hole [%ld, %ld) must be quad-aligned, inside the %ld parameters and not hold the clamped one", (long)hole_lo, (long)hole_hi, (long)n);
    if ((dev_state && tick) || counter)
        hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, tick ? dev_state : nullptr, b1, b2, (long long*)counter, (long long)counter_add);
    const long live = n - (hole_hi - hole_lo);
    if (live == 0) return MH_OK;
#define ADAM_LAUNCH_(KERN) hipLaunchKernelGGL(KERN, dim3((unsigned)min((long)mh_cdiv(mh_cdiv(live, 4), 256), 8192L)), dim3(256), 0, (hipStream_t)s, p, g, m, v, (bf16_t*)shadow, (long)n, lr, b1, b2, eps, bc1, bc2, gscale, (const float*)dev_state, \
                       clamp_index < 0 ? -1L : (long)clamp_index, clamp_lo, clamp_hi, (long)(hole_lo / 4), (long)(hole_hi / 4))
    if (tick == 2) ADAM_LAUNCH_(adam_range_kernel); else ADAM_LAUNCH_(adam_kernel);
#undef ADAM_LAUNCH_
    MH_LAUNCH_CHECK("mh_adam");
    return MH_OK;
}
