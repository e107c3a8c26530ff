// MIRRORLoss's small terms in ONE launch each way (losses/mirror_loss.py:74-135): the rank-local CLIP alignment term
// (logits, both cross-entropies and their gradient products, :37-52), the RNA retention MSE (:98-103, D = 1), the two style
// KLs (:105-112), the prototype cluster KL (:114-119) and the weighted total (:121-127).  As separate entry points these are
// ~28 launches of 3-10 us kernels in a row on the critical path between the forward and the backward of the step, ~9 us of
// dependent-launch latency apiece inside the replayed HIP graph; only the WSI retention MSE (400 MB of HBM traffic) stays
// a kernel of its own (mh_mse_masked_fwd / _bwd), launched before the forward and after the backward kernel here.
//
// Block roles (256 threads): [0, Bc) one prototype-score row each; then the alignment block, the two style blocks and NMSE
// blocks of the flat RNA MSE.  The forward's last block to finish (device-scope counter in the zeroed scratch) folds the
// terms into out[8] = {total, alignment, wsi retention, rna retention, style (sum), cluster, style_w, style_r}.
#include "common.h"

namespace {

constexpr int NMSE = 32;       // blocks of the flat RNA MSE (4 elements per thread at B x G = 16 x 2048)
constexpr int SK_E = 16;       // prototype scores per thread held in registers (P <= 4096; longer rows re-read global memory)
constexpr int BMAX = 32;       // alignment block: B x B logits, B <= 32 (the batch of one rank)

// scratch words (zeroed by the caller): partial sums and the completion counter
enum { A_ALIGN = 0, A_RNUM = 1, A_RDEN = 2, A_STW = 3, A_STR = 4, A_CLU = 5, A_COUNT = 7 };

__device__ __forceinline__ float dev_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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

// The same statistics with the row held in registers (P <= 256 * SK_E): every load of the row is issued up front, the passes
// below run on registers (the global-memory form pays one L2 round trip per pass and element: 4-5 passes x 12 elements).
__device__ __forceinline__ void load_row(const float* __restrict__ w, const float* __restrict__ r, int P, float (&wv)[SK_E], float (&rv)[SK_E]) {
#pragma unroll
    for (int e = 0; e < SK_E; e++) {
        const int k = threadIdx.x + e * 256;
        wv[e] = k < P ? w[k] : -INFINITY;
        rv[e] = k < P ? r[k] : -INFINITY;
    }
}
__device__ __forceinline__ RowStats row_stats_regs(const float (&wv)[SK_E], const float (&rv)[SK_E], float* red) {
    float mw = -INFINITY, mr = -INFINITY;
#pragma unroll
    for (int e = 0; e < SK_E; e++) { mw = fmaxf(mw, wv[e]); mr = fmaxf(mr, rv[e]); }
    mw = block_max256(mw, red);
    mr = block_max256(mr, red);
    float sw = 0.f, sr = 0.f;
#pragma unroll
    for (int e = 0; e < SK_E; e++) { sw += __expf(wv[e] - mw); sr += __expf(rv[e] - mr); }      // exp(-inf) = 0 past the row
    sw = block_sum256(sw, red);
    sr = block_sum256(sr, red);
    return {mw, __logf(sw), mr, __logf(sr)};
}

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void loss_terms_fwd_kernel(mh_loss_terms d) {
    extern __shared__ float lds[];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int role = blockIdx.x;
    if (role < d.Bc) {                                  // ---- cluster: one score row
        const float* wr = d.w_score + (long)role * d.P;
        const float* rr = d.r_score + (long)role * d.P;
        float s = 0.f;
        if (d.P <= 256 * SK_E) {
            float wv[SK_E], rv[SK_E];
            load_row(wr, rr, d.P, wv, rv);
            const RowStats st = row_stats_regs(wv, rv, red);
#pragma unroll
            for (int e = 0; e < SK_E; e++)
                if (tid + e * 256 < d.P) {
                    const float lw = wv[e] - st.mw - st.sw, lr = rv[e] - st.mr - st.sr;
                    s += (__expf(lr) - __expf(lw)) * (lr - lw);
                }
        } else {
            const RowStats st = row_stats(wr, rr, d.P, red);
            for (int k = tid; k < d.P; k += 256) {
                const float lw = wr[k] - st.mw - st.sw, lr = rr[k] - st.mr - st.sr;
                s += (__expf(lr) - __expf(lw)) * (lr - lw);
            }
        }
        s = block_sum256(s, red);
        if (tid == 0) atomicAdd(d.scratch + A_CLU, (0.5f / d.Bc) * s);
    } else if ((role -= d.Bc) == 0) {                   // ---- alignment
        if (d.has_align) {
            const int B = d.B, D = d.D, ldw = D + 1, ldg = B + 1;
            float* Ws = lds;
            float* Rs = Ws + B * ldw;
            float* Gs = Rs + B * ldw;
            if ((D & 3) == 0) {          // quads, four of each matrix in flight per thread
                const int nq = B * D / 4;
#pragma unroll 4
                for (int q = tid; q < nq; q += 256) {
                    const float4 a = reinterpret_cast<const float4*>(d.wsi_emb)[q], b = reinterpret_cast<const float4*>(d.rna_emb)[q];
                    const int r = (4 * q) / D, c = 4 * q - r * D;
                    float* wp = Ws + r * ldw + c;
                    float* rp = Rs + r * ldw + c;
                    wp[0] = a.x; wp[1] = a.y; wp[2] = a.z; wp[3] = a.w;
                    rp[0] = b.x; rp[1] = b.y; rp[2] = b.z; rp[3] = b.w;
                }
            } else {
                for (int i = tid; i < B * D; i += 256) {
                    const int r = i / D, c = i - r * D;
                    Ws[r * ldw + c] = d.wsi_emb[i];
                    Rs[r * ldw + c] = d.rna_emb[i];
                }
            }
            __syncthreads();
            for (int e = tid; e < B * B; e += 256) {
                const int r = e / B, c = e - r * B;
                const float* wp = Ws + r * ldw;
                const float* rp = Rs + c * ldw;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                int k = 0;
                for (; k + 32 <= D; k += 32) {       // 64 LDS reads in flight per trip: the loop is LDS-latency bound otherwise
                    float wq[32], rq[32];
#pragma unroll
                    for (int u = 0; u < 32; u++) { wq[u] = wp[k + u]; rq[u] = rp[k + u]; }
#pragma unroll
                    for (int u = 0; u < 32; u += 4) { a0 += wq[u] * rq[u]; a1 += wq[u + 1] * rq[u + 1]; a2 += wq[u + 2] * rq[u + 2]; a3 += wq[u + 3] * rq[u + 3]; }
                }
                for (; k + 4 <= D; k += 4) { a0 += wp[k] * rp[k]; a1 += wp[k + 1] * rp[k + 1]; a2 += wp[k + 2] * rp[k + 2]; a3 += wp[k + 3] * rp[k + 3]; }
                for (; k < D; k++) a0 += wp[k] * rp[k];
                const float g = (a0 + a1) + (a2 + a3);
                Gs[r * ldg + c] = g;
                d.save[e] = g;
            }
            __syncthreads();
            const float s = d.logit_scale[0];
            float loss = 0.f;        // lane 0 of each wave carries its rows' sums
            for (int r = wave; r < B; r += 4) {
                // image -> rna direction: row r of s G; rna -> image: column r
                const float vr = lane < B ? s * Gs[r * ldg + lane] : -INFINITY;
                const float vc = lane < B ? s * Gs[lane * ldg + r] : -INFINITY;
                const float mr = wave_max(vr), mc = wave_max(vc);
                const float sr = wave_sum(lane < B ? __expf(vr - mr) : 0.f), sc = wave_sum(lane < B ? __expf(vc - mc) : 0.f);
                const float lr = mr + __logf(sr), lc = mc + __logf(sc);
                if (lane == 0) {
                    d.save[B * B + r] = lr;
                    d.save[B * B + B + r] = lc;
                    loss += (lr + lc) - 2.f * s * Gs[r * ldg + r];
                }
            }
            loss = block_sum256(lane == 0 ? loss : 0.f, red);
            if (tid == 0) atomicAdd(d.scratch + A_ALIGN, (0.5f / B) * loss);
        } else if (tid == 0 && d.align_ext) {
            atomicAdd(d.scratch + A_ALIGN, d.align_ext[0]);
        }
    } else if (role <= 2) {                             // ---- style KL (wsi, rna)
        const float* mu = role == 1 ? d.w_mu : d.r_mu;
        const float* ls = role == 1 ? d.w_logstd : d.r_logstd;
        const long n = role == 1 ? d.n_wstyle : d.n_rstyle;
        const int rows = role == 1 ? d.rows_wstyle : d.rows_rstyle;
        float s = 0.f;
#pragma unroll 8
        for (long i = tid; i < n; i += 256) s += __expf(ls[i]) + mu[i] * mu[i] - 1.f - ls[i];
        s = block_sum256(s, red);
        if (tid == 0) atomicAdd(d.scratch + (role == 1 ? A_STW : A_STR), (0.5f / rows) * s);
    } else {                                            // ---- RNA retention, flat masked MSE
        const int b = role - 3;
        float num = 0.f, den = 0.f;
#pragma unroll 4
        for (long i = (long)b * 256 + tid; i < d.n_rna; i += (long)NMSE * 256) {
            const float mk = d.rna_mask[i], df = d.rna_pred[i] - d.rna_tgt[i];
            num += mk * df * df;
            den += mk;
        }
        num = block_sum256(num, red);
        den = block_sum256(den, red);
        if (tid == 0) { atomicAdd(d.scratch + A_RNUM, num); atomicAdd(d.scratch + A_RDEN, den); }
    }
    // ---- the last block to get here folds the terms (every block's sums are device-visible before its counter tick)
    if (tid == 0) {
        __threadfence();
        const unsigned done = atomicAdd(reinterpret_cast<unsigned*>(d.scratch + A_COUNT), 1u);
        if (done == gridDim.x - 1) {
            __threadfence();
            const float align = dev_load(d.scratch + A_ALIGN);
            const float wsi = d.wsi_acc ? d.wsi_acc[0] / d.wsi_acc[1] : 0.f;     // written by the launch before this one
            const float rna = dev_load(d.scratch + A_RNUM) / dev_load(d.scratch + A_RDEN);
            const float sw = dev_load(d.scratch + A_STW), sr = dev_load(d.scratch + A_STR), clu = dev_load(d.scratch + A_CLU);
            d.out[0] = d.weight[0] * align + d.weight[1] * wsi + d.weight[2] * rna + d.weight[3] * sw + d.weight[4] * sr + d.weight[5] * clu;
            d.out[1] = align;
            d.out[2] = wsi;
            d.out[3] = rna;
            d.out[4] = sw + sr;
            d.out[5] = clu;
            d.out[6] = sw;
            d.out[7] = sr;
        }
    }
}

// ------------------------------------------------------------------ backward
// upstream of term i: u_i = weight[i] * g_total[0] (+ g_terms[i] when the caller also backpropagates through a single term)
__device__ __forceinline__ float upstream(const mh_loss_terms& d, int i) {
    return d.weight[i] * (d.g_total ? d.g_total[0] : 0.f) + (d.g_terms ? d.g_terms[i] : 0.f);
}

constexpr int NALIGN = 2;      // backward blocks of the alignment term

// d wsi_emb = P rna_emb, d rna_emb = P^T wsi_emb with P = dL / dG [B, B] rebuilt from the saved logits: a thread per feature
// column, P zero-padded to BM x BM in LDS and broadcast from there (no per-j guards: rows past B multiply zeros).
template <int BM>
__device__ __forceinline__ void align_bwd(const mh_loss_terms& d, float* Ps, float* red, float u, int ablk) {
    const int tid = threadIdx.x, B = d.B, D = d.D;
    const float s = d.logit_scale[0], w = u * (0.5f / B);
    float ds = 0.f;
    for (int e = tid; e < BM * BM; e += 256) {
        const int r = e / BM, c = e - r * BM;
        float pv = 0.f;
        if (r < B && c < B) {
            const float g = d.save[r * B + c], eye = r == c ? 2.f : 0.f;
            const float p = __expf(s * g - d.save[B * B + r]) + __expf(s * g - d.save[B * B + B + c]) - eye;
            pv = w * s * p;
            ds += w * p * g;
        }
        Ps[e] = pv;
    }
    ds = block_sum256(ds, red);
    if (tid == 0 && ablk == 0 && d.d_logit_scale) d.d_logit_scale[0] = ds;
    __syncthreads();
    for (int c = ablk * 256 + tid; c < D; c += NALIGN * 256) {
        float wv[BM], rv[BM];
#pragma unroll
        for (int j = 0; j < BM; j++) {
            const int jj = j < B ? j : B - 1;
            const float a = d.wsi_emb[(long)jj * D + c], b = d.rna_emb[(long)jj * D + c];
            wv[j] = j < B ? a : 0.f;
            rv[j] = j < B ? b : 0.f;
        }
        for (int r = 0; r < B; r++) {
            float aw0 = 0.f, aw1 = 0.f, ar0 = 0.f, ar1 = 0.f;
#pragma unroll
            for (int j = 0; j < BM; j += 2) {
                aw0 += Ps[r * BM + j] * rv[j];
                aw1 += Ps[r * BM + j + 1] * rv[j + 1];
                ar0 += Ps[j * BM + r] * wv[j];
                ar1 += Ps[(j + 1) * BM + r] * wv[j + 1];
            }
            d.d_wsi_emb[(long)r * D + c] = aw0 + aw1;
            d.d_rna_emb[(long)r * D + c] = ar0 + ar1;
        }
    }
}

__global__ __launch_bounds__(256) void loss_terms_bwd_kernel(mh_loss_terms d) {
    extern __shared__ float lds[];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    int role = blockIdx.x;
    if (role < d.Bc) {                                  // ---- cluster row
        const long base = (long)role * d.P;
        const float* wr = d.w_score + base;
        const float* rr = d.r_score + base;
        const float c = upstream(d, 5) * (0.5f / d.Bc);
        float er = 0.f, ew = 0.f;
        if (d.P <= 256 * SK_E) {
            float wv[SK_E], rv[SK_E];
            load_row(wr, rr, d.P, wv, rv);
            const RowStats st = row_stats_regs(wv, rv, red);
#pragma unroll
            for (int e = 0; e < SK_E; e++)
                if (tid + e * 256 < d.P) {
                    const float lw = wv[e] - st.mw - st.sw, lr = rv[e] - st.mr - st.sr;
                    wv[e] = lw;
                    rv[e] = lr;
                    er += __expf(lr) * (lr - lw);
                    ew += __expf(lw) * (lr - lw);
                }
            er = block_sum256(er, red);
            ew = block_sum256(ew, red);
#pragma unroll
            for (int e = 0; e < SK_E; e++) {
                const int k = tid + e * 256;
                if (k < d.P) {
                    const float pw = __expf(wv[e]), pr = __expf(rv[e]), df = rv[e] - wv[e], q = pr - pw;
                    d.d_r_score[base + k] = c * (pr * (df - er) + q);
                    d.d_w_score[base + k] = c * (-pw * (df - ew) - q);
                }
            }
        } else {
            const RowStats st = row_stats(wr, rr, d.P, red);
            for (int k = tid; k < d.P; k += 256) {
                const float lw = wr[k] - st.mw - st.sw, lr = rr[k] - st.mr - st.sr;
                er += __expf(lr) * (lr - lw);
                ew += __expf(lw) * (lr - lw);
            }
            er = block_sum256(er, red);
            ew = block_sum256(ew, red);
            for (int k = tid; k < d.P; k += 256) {
                const float lw = wr[k] - st.mw - st.sw, lr = rr[k] - st.mr - st.sr;
                const float pw = __expf(lw), pr = __expf(lr), df = lr - lw, q = pr - pw;
                d.d_r_score[base + k] = c * (pr * (df - er) + q);
                d.d_w_score[base + k] = c * (-pw * (df - ew) - q);
            }
        }
    } else if ((role -= d.Bc) < NALIGN) {               // ---- alignment (NALIGN blocks share the feature columns)
        const int ablk = role;
        const float u = upstream(d, 0);
        if (!d.has_align) {
            if (tid == 0 && ablk == 0 && d.d_align_ext) d.d_align_ext[0] = u;
            return;
        }
        if (d.B <= 16) align_bwd<16>(d, lds, red, u, ablk);
        else align_bwd<BMAX>(d, lds, red, u, ablk);
    } else if ((role -= NALIGN - 1) <= 2) {             // ---- style KL
        const float* mu = role == 1 ? d.w_mu : d.r_mu;
        const float* ls = role == 1 ? d.w_logstd : d.r_logstd;
        float* dmu = role == 1 ? d.d_w_mu : d.d_r_mu;
        float* dls = role == 1 ? d.d_w_logstd : d.d_r_logstd;
        const long n = role == 1 ? d.n_wstyle : d.n_rstyle;
        const float k = upstream(d, role == 1 ? 3 : 4) * (0.5f / (role == 1 ? d.rows_wstyle : d.rows_rstyle));
#pragma unroll 8
        for (long i = tid; i < n; i += 256) {
            dmu[i] = k * 2.f * mu[i];
            dls[i] = k * (__expf(ls[i]) - 1.f);
        }
    } else {                                            // ---- RNA retention
        const int b = role - 3;
        const float k = upstream(d, 2) * 2.f / d.scratch[A_RDEN];
#pragma unroll 4
        for (long i = (long)b * 256 + tid; i < d.n_rna; i += (long)NMSE * 256) {
            const float mk = d.rna_mask[i];
            const float g = mk != 0.f ? k * mk * (d.rna_pred[i] - d.rna_tgt[i]) : 0.f;
            d.d_rna_pred[i] = g;
            if (d.d_rna_tgt) d.d_rna_tgt[i] = -g;
        }
    }
}

size_t align_lds_bytes(const mh_loss_terms* d) {
    return d->has_align ? sizeof(float) * ((size_t)2 * d->B * (d->D + 1) + (size_t)d->B * (d->B + 1)) : sizeof(float) * 4;
}

int check(const mh_loss_terms* d, const char* who) {
    MH_REQUIRE(d && d->scratch && d->out && d->w_score && d->r_score && d->w_mu && d->w_logstd && d->r_mu && d->r_logstd && d->rna_pred &&
                   d->rna_tgt && d->rna_mask, "%s: null pointer", who);
    MH_REQUIRE(d->Bc > 0 && d->P > 0 && d->n_rna > 0 && d->n_wstyle > 0 && d->n_rstyle > 0 && d->rows_wstyle > 0 && d->rows_rstyle > 0,
               "%s: empty term", who);
    if (d->has_align) {
        MH_REQUIRE(d->wsi_emb && d->rna_emb && d->logit_scale && d->save, "%s: alignment operands missing", who);
        MH_REQUIRE(d->B >= 1 && d->B <= BMAX && d->D >= 1 && align_lds_bytes(d) <= 150 * 1024,
                   "%s: alignment block takes B <= %d and 2 B (D + 1) floats of LDS (B=%d D=%d)", who, BMAX, d->B, d->D);
    }
    return MH_OK;
}

}  // namespace

extern "C" int mh_loss_terms_fwd(const mh_loss_terms* d, mh_stream s) {
    const int rc = check(d, "mh_loss_terms_fwd");
    if (rc != MH_OK) return rc;
    const size_t lds = align_lds_bytes(d);
    static bool once = [] { return hipFuncSetAttribute((const void*)loss_terms_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess; }();
    MH_REQUIRE(once, "mh_loss_terms_fwd: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(loss_terms_fwd_kernel, dim3(d->Bc + 3 + NMSE), dim3(256), lds, (hipStream_t)s, *d);
    MH_LAUNCH_CHECK("mh_loss_terms_fwd");
    return MH_OK;
}

extern "C" int mh_loss_terms_bwd(const mh_loss_terms* d, mh_stream s) {
    const int rc = check(d, "mh_loss_terms_bwd");
    if (rc != MH_OK) return rc;
    MH_REQUIRE(d->d_w_score && d->d_r_score && d->d_w_mu && d->d_w_logstd && d->d_r_mu && d->d_r_logstd && d->d_rna_pred,
               "mh_loss_terms_bwd: null gradient pointer");
    MH_REQUIRE(!d->has_align || (d->d_wsi_emb && d->d_rna_emb), "mh_loss_terms_bwd: null embedding gradient pointer");
    const size_t lds = sizeof(float) * BMAX * BMAX;
    hipLaunchKernelGGL(loss_terms_bwd_kernel, dim3(d->Bc + NALIGN + 2 + NMSE), dim3(256), lds, (hipStream_t)s, *d);
    MH_LAUNCH_CHECK("mh_loss_terms_bwd");
    return MH_OK;
}
