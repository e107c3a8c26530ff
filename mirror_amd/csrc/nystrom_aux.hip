// Nystrom-attention side kernels ([3P] nystrom_attention, called at models/mirror.py:312) and the
// TransMIL sequence glue (models/mirror.py:657-665): landmark means, 33-tap residual conv of V,
// pseudo-inverse initial scaling (tensor-wide max) and its adjoint, d*I - P, cls/square-pad rows.
#include "common.h"
#include <cstdlib>

// resconv_mfma.hip
bool resconv_try_mfma(const void* v, long ldv, long v_bs, const float* w, void* out, long ldo, long o_bs, int B, int n_p, int heads,
                      int dh, int taps, int transpose, int accumulate, int dt_v, int dt_o, hipStream_t s);
bool resconv_wgrad_try_mfma(const void* v, long ldv, long v_bs, const void* dout, long ldo, long o_bs, float* dw, int B, int n_p,
                            int heads, int dh, int taps, int dt_v, int dt_o, hipStream_t s);
long resconv_bwd_part_floats(int B, int n_p, int heads, int dh, int taps);
bool resconv_bwd_try_mfma(const void* dout, long ldo, long o_bs, const void* v, long ldv, long v_bs, const float* w, void* dv, long lddv,
                          long dv_bs, float* dw, float* part, long part_floats, int B, int n_p, int heads, int dh, int taps, int dt,
                          hipStream_t s);
static bool resconv_mfma_on() { return true; }

// ------------------------------------------------------------------ landmarks
// lm[b, j, c] = (1/l) sum_t qkv[b, j*l + t, c], c < 2D (q and k column blocks)
template <typename T>
__global__ __launch_bounds__(256) void landmark_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ lm, int B, int n_p,
                                                           int D, int l) {
    const int m = n_p / l;
    const long total = (long)B * m * 2 * D;
    const float inv = 1.f / l;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx % (2 * D);
        const long bj = idx / (2 * D);
        const int j = bj % m;
        const long b = bj / m;
        const T* src = qkv + (b * n_p + (long)j * l) * 3 * D + c;
        float s = 0.f;
        for (int t = 0; t < l; t++) s += ldf(src + (long)t * 3 * D);
        stf(lm + idx, s * inv);
    }
}

// quad form (D % 4 == 0, quad-aligned): a thread owns 4 consecutive columns of one landmark; same summation order
template <typename T>
__global__ __launch_bounds__(256) void landmark_fwd_vec_kernel(const T* __restrict__ qkv, T* __restrict__ lm, int B, int n_p,
                                                               int D, int l) {
    const int m = n_p / l, q2 = 2 * D / 4;
    const long total = (long)B * m * q2;
    const float inv = 1.f / l;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % q2) * 4;
        const long bj = idx / q2;
        const int j = bj % m;
        const long b = bj / m;
        const T* src = qkv + (b * n_p + (long)j * l) * 3 * D + c;
        f4_t s = {0.f, 0.f, 0.f, 0.f};
        int t = 0;
        for (; t + 4 <= l; t += 4) {      // four rows in flight
            const f4_t a0 = ld4(src + (long)t * 3 * D), a1 = ld4(src + (long)(t + 1) * 3 * D), a2 = ld4(src + (long)(t + 2) * 3 * D),
                       a3 = ld4(src + (long)(t + 3) * 3 * D);
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; t < l; t++) s += ld4(src + (long)t * 3 * D);
        st4(lm + bj * 2 * D + c, s * inv);
    }
}

// dqkv[b, r, c] += dlm[b, r / l, c] / l, c < 2D
template <typename T>
__global__ __launch_bounds__(256) void landmark_bwd_kernel(const T* __restrict__ dlm, T* __restrict__ dqkv, int B, int n_p,
                                                           int D, int l) {
    const int m = n_p / l;
    const long total = (long)B * n_p * 2 * D;
    const float inv = 1.f / l;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx % (2 * D);
        const long br = idx / (2 * D);
        const int r = br % n_p;
        const long b = br / n_p;
        T* dst = dqkv + (b * n_p + r) * 3 * D + c;
        stf(dst, ldf(dst) + ldf(dlm + (b * m + r / l) * 2 * D + c) * inv);
    }
}

// 8 columns per thread (16-B bf16 / 2 x 16-B f32 accesses); needs D % 4 == 0 and 16-B aligned rows
template <typename T>
__global__ __launch_bounds__(256) void landmark_bwd_vec_kernel(const T* __restrict__ dlm, T* __restrict__ dqkv, int B, int n_p,
                                                               int D, int l);

extern "C" int mh_landmark_fwd(const void* qkv, void* lm, int B, int n_p, int D, int l, int dt, mh_stream s) {
    MH_REQUIRE(l >= 1 && n_p % l == 0, "mh_landmark_fwd: n_p=%d not a multiple of l=%d", n_p, l);
    const long total = (long)B * (n_p / l) * 2 * D;
    if (total == 0) return MH_OK;
    if (D % 4 == 0 && mh_quad_ok(qkv, mh_dt_size(dt)) && mh_quad_ok(lm, mh_dt_size(dt))) {
        dim3 gv((unsigned)min((long)mh_cdiv(total / 4, 256), 16384L));
        MH_DISPATCH_DT(dt, T, hipLaunchKernelGGL((landmark_fwd_vec_kernel<T>), gv, dim3(256), 0, (hipStream_t)s, (const T*)qkv, (T*)lm, B, n_p, D, l));
        MH_LAUNCH_CHECK("mh_landmark_fwd");
        return MH_OK;
    }
    dim3 grid((unsigned)min((long)mh_cdiv(total, 256), 8192L));
    MH_DISPATCH_DT(dt, T, hipLaunchKernelGGL((landmark_fwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)s, (const T*)qkv, (T*)lm, B, n_p, D, l));
    MH_LAUNCH_CHECK("mh_landmark_fwd");
    return MH_OK;
}

extern "C" int mh_landmark_bwd(const void* dlm, void* dqkv, int B, int n_p, int D, int l, int dt, mh_stream s) {
    MH_REQUIRE(l >= 1 && n_p % l == 0, "mh_landmark_bwd: n_p=%d not a multiple of l=%d", n_p, l);
    const long total = (long)B * n_p * 2 * D;
    if (total == 0) return MH_OK;
    if (D % 8 == 0 && ((uintptr_t)dlm & 15) == 0 && ((uintptr_t)dqkv & 15) == 0) {
        dim3 vgrid((unsigned)min((long)mh_cdiv(total / 8, 256), 8192L));
        MH_DISPATCH_DT(dt, T, hipLaunchKernelGGL((landmark_bwd_vec_kernel<T>), vgrid, dim3(256), 0, (hipStream_t)s, (const T*)dlm, (T*)dqkv, B, n_p, D, l));
        MH_LAUNCH_CHECK("mh_landmark_bwd");
        return MH_OK;
    }
    dim3 grid((unsigned)min((long)mh_cdiv(total, 256), 8192L));
    MH_DISPATCH_DT(dt, T, hipLaunchKernelGGL((landmark_bwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)s, (const T*)dlm, (T*)dqkv, B, n_p, D, l));
    MH_LAUNCH_CHECK("mh_landmark_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------ residual depthwise conv over the sequence
// out[b,t,c] (+)= sum_j w[h(c)][jj] * v[b, t + j - taps/2, c],  jj = j (forward) or taps-1-j (adjoint).
// Each thread owns one column c and RT consecutive rows: a sliding window keeps the re-read factor at
// (RT+taps-1)/RT instead of taps.
#define RC_RT 16
#define RC_MAXTAPS 64
template <typename TV, typename TO>
__global__ __launch_bounds__(256) void resconv_kernel(const TV* __restrict__ v, long ldv, long v_bs, const float* __restrict__ w,
                                                      TO* out, long ldo, long o_bs, int n_p, int C, int dh, int taps,
                                                      int transpose, int accumulate) {
    __shared__ float ws[8 * RC_MAXTAPS];  // weights of every head touched by this block's 256 columns (<= 8)
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int t0 = blockIdx.y * RC_RT;
    const int b = blockIdx.z;
    const int h0 = (blockIdx.x * 256) / dh;
    const int nh = min((blockIdx.x * 256 + 255) / dh, (C - 1) / dh) - h0 + 1;
    for (int i = threadIdx.x; i < nh * taps; i += 256) {
        const int hh = i / taps, j = i % taps;
        ws[hh * RC_MAXTAPS + j] = w[(h0 + hh) * taps + (transpose ? taps - 1 - j : j)];
    }
    __syncthreads();
    if (c >= C) return;
    const float* wl = ws + (c / dh - h0) * RC_MAXTAPS;
    const int half = taps / 2;
    float acc[RC_RT];
#pragma unroll
    for (int i = 0; i < RC_RT; i++) acc[i] = 0.f;
    const TV* vb = v + (long)b * v_bs + c;
    for (int u = 0; u < RC_RT + taps - 1; u++) {
        const int t = t0 - half + u;  // input row
        const float x = (t >= 0 && t < n_p) ? ldf(vb + (long)t * ldv) : 0.f;
#pragma unroll
        for (int i = 0; i < RC_RT; i++) {
            const int j = u - i;  // tap index for output row t0+i
            if (j >= 0 && j < taps) acc[i] += wl[j] * x;
        }
    }
    TO* ob = out + (long)b * o_bs + c;
#pragma unroll
    for (int i = 0; i < RC_RT; i++) {
        const int t = t0 + i;
        if (t < n_p) {
            float r = acc[i];
            if (accumulate) r += ldf(ob + (long)t * ldo);
            stf(ob + (long)t * ldo, r);
        }
    }
}

// Vectorised variant (dh % 8 == 0): a thread owns 8 adjacent columns (one 16-B bf16 load per input row) and
// RV_RT consecutive output rows; block = 64 column groups x 4 row blocks.
#define RV_RT 8
typedef float rf4 __attribute__((ext_vector_type(4)));
typedef unsigned ru4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&o)[8]) {
    const rf4 a = *reinterpret_cast<const rf4*>(p), b = *reinterpret_cast<const rf4*>(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&o)[8]) {
    const ru4 r = *reinterpret_cast<const ru4*>(p);
#pragma unroll
    for (int e = 0; e < 4; e++) { o[2 * e] = __uint_as_float(r[e] << 16); o[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<rf4*>(p) = (rf4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<rf4*>(p + 4) = (rf4){v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    ru4 r;
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
    *reinterpret_cast<ru4*>(p) = r;
}

template <typename T>
__global__ __launch_bounds__(256) void landmark_bwd_vec_kernel(const T* __restrict__ dlm, T* __restrict__ dqkv, int B, int n_p,
                                                               int D, int l) {
    const int m = n_p / l;
    const int cpr = 2 * D / 8;                       // 8-column chunks per row (q and k blocks)
    const long total = (long)B * n_p * cpr;
    const float inv = 1.f / l;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (idx % cpr) * 8;
        const long br = idx / cpr;
        const int r = br % n_p;
        const long b = br / n_p;
        T* dst = dqkv + (b * n_p + r) * 3 * D + c;
        float a[8], g[8];
        ld8(dst, a);
        ld8(dlm + (b * m + r / l) * 2 * D + c, g);
#pragma unroll
        for (int e = 0; e < 8; e++) a[e] += g[e] * inv;
        st8(dst, a);
    }
}

template <typename TV, typename TO>
__global__ __launch_bounds__(256) void resconv_vec_kernel(const TV* __restrict__ v, long ldv, long v_bs, const float* __restrict__ w,
                                                          TO* out, long ldo, long o_bs, int n_p, int C, int dh, int taps,
                                                          int transpose, int accumulate) {
    __shared__ float ws[8 * RC_MAXTAPS];
    const int cg = threadIdx.x & 63, rb = threadIdx.x >> 6;
    const int c = blockIdx.x * 512 + cg * 8;
    const int t0 = (blockIdx.y * 4 + rb) * RV_RT;
    const int b = blockIdx.z;
    const int h0 = (blockIdx.x * 512) / dh;
    const int nh = min((blockIdx.x * 512 + 511) / dh, (C - 1) / dh) - h0 + 1;
    for (int i = threadIdx.x; i < nh * taps; i += 256) {
        const int hh = i / taps, j = i % taps;
        ws[hh * RC_MAXTAPS + j] = w[(h0 + hh) * taps + (transpose ? taps - 1 - j : j)];
    }
    __syncthreads();
    if (c >= C || t0 >= n_p) return;
    const float* wl = ws + (c / dh - h0) * RC_MAXTAPS;
    const int half = taps / 2;
    float acc[RV_RT][8];
#pragma unroll
    for (int i = 0; i < RV_RT; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) acc[i][e] = 0.f;
    const TV* vb = v + (long)b * v_bs + c;
    for (int u = 0; u < RV_RT + taps - 1; u++) {
        const int t = t0 - half + u;
        if (t < 0 || t >= n_p) continue;
        float x[8];
        ld8(vb + (long)t * ldv, x);
#pragma unroll
        for (int i = 0; i < RV_RT; i++) {
            const int j = u - i;
            if (j >= 0 && j < taps) {
                const float wj = wl[j];
#pragma unroll
                for (int e = 0; e < 8; e++) acc[i][e] += wj * x[e];
            }
        }
    }
    TO* ob = out + (long)b * o_bs + c;
#pragma unroll
    for (int i = 0; i < RV_RT; i++) {
        const int t = t0 + i;
        if (t < n_p) {
            if (accumulate) {
                float old[8];
                ld8(ob + (long)t * ldo, old);
#pragma unroll
                for (int e = 0; e < 8; e++) acc[i][e] += old[e];
            }
            st8(ob + (long)t * ldo, acc[i]);
        }
    }
}

extern "C" int mh_resconv_fwd(const void* v, int64_t ldv, int64_t v_bs, const float* w, void* out, int64_t ldo,
                              int64_t o_bs, int B, int n_p, int heads, int dh, int taps, int transpose, int accumulate,
                              int dt_v, int dt_o, mh_stream s) {
    MH_REQUIRE(taps >= 1 && taps <= RC_MAXTAPS && (taps & 1), "mh_resconv_fwd: taps=%d unsupported", taps);
    const int C = heads * dh;
    MH_REQUIRE(heads <= 8 || 255 / dh + 2 <= 8, "mh_resconv_fwd: more than 8 heads per 256 columns (heads=%d dh=%d)", heads, dh);
    if (B == 0 || n_p == 0) return MH_OK;
    if (resconv_mfma_on() && resconv_try_mfma(v, ldv, v_bs, w, out, ldo, o_bs, B, n_p, heads, dh, taps, transpose, accumulate, dt_v,
                                              dt_o, (hipStream_t)s)) {   // bf16, dh = 64, 33 taps: banded Toeplitz product on MFMA
        MH_LAUNCH_CHECK("mh_resconv_fwd");
        return MH_OK;
    }
    if (dh % 8 == 0 && ldv % 8 == 0 && v_bs % 8 == 0 && ldo % 8 == 0 && o_bs % 8 == 0 && ((uintptr_t)v & 15) == 0 &&
        ((uintptr_t)out & 15) == 0 && (heads <= 8 || 511 / dh + 2 <= 8)) {
        dim3 vgrid(mh_cdiv(C, 512), mh_cdiv(n_p, 4 * RV_RT), B);
#define RCV(TV, TO) hipLaunchKernelGGL((resconv_vec_kernel<TV, TO>), vgrid, dim3(256), 0, (hipStream_t)s, (const TV*)v, (long)ldv, (long)v_bs, w, (TO*)out, (long)ldo, (long)o_bs, n_p, C, dh, taps, transpose, accumulate)
        if (dt_v == MH_F32 && dt_o == MH_F32) RCV(float, float);
        else if (dt_v == MH_BF16 && dt_o == MH_BF16) RCV(bf16_t, bf16_t);
        else if (dt_v == MH_BF16 && dt_o == MH_F32) RCV(bf16_t, float);
        else RCV(float, bf16_t);
#undef RCV
        MH_LAUNCH_CHECK("mh_resconv_fwd");
        return MH_OK;
    }
    dim3 grid(mh_cdiv(C, 256), mh_cdiv(n_p, RC_RT), B);
#define RC(TV, TO) hipLaunchKernelGGL((resconv_kernel<TV, TO>), grid, dim3(256), 0, (hipStream_t)s, (const TV*)v, (long)ldv, (long)v_bs, w, (TO*)out, (long)ldo, (long)o_bs, n_p, C, dh, taps, transpose, accumulate)
    if (dt_v == MH_F32 && dt_o == MH_F32) RC(float, float);
    else if (dt_v == MH_BF16 && dt_o == MH_BF16) RC(bf16_t, bf16_t);
    else if (dt_v == MH_BF16 && dt_o == MH_F32) RC(bf16_t, float);
    else RC(float, bf16_t);
#undef RC
    MH_LAUNCH_CHECK("mh_resconv_fwd");
    return MH_OK;
}

// dw[h][j] += sum_{b,t,d} dout[b,t,h,d] * v[b,t+j-half,h,d].  Block = (head h, 64 rows, b): the dout tile [64][64]
// and the v tile [64+taps-1][64] sit in LDS (f32, pitch 68: the two rows a 16-lane ds_read_b128 group touches land
// on disjoint banks); thread (tap j = tid>>3, q = tid&7) dots 8-column slices over the 64 rows, the 8 q-lanes
// are combined with shuffles, one f32 atomic per (head, tap) per block.
#define RW_ROWS 64
#define RW_PITCH 68
template <typename TV, typename TO>
__global__ __launch_bounds__(256) void resconv_wgrad_kernel(const TV* __restrict__ v, long ldv, long v_bs,
                                                            const TO* __restrict__ dout, long ldo, long o_bs,
                                                            float* __restrict__ dw, int n_p, int dh, int taps, int vec8) {
    __shared__ __attribute__((aligned(16))) float vs[(RW_ROWS + RC_MAXTAPS) * RW_PITCH];
    __shared__ __attribute__((aligned(16))) float ds[RW_ROWS * RW_PITCH];
    const int h = blockIdx.x, t0 = blockIdx.y * RW_ROWS, b = blockIdx.z;
    const int half = taps / 2;
    const int tid = threadIdx.x;
    const int q = tid & 7, j0 = tid >> 3;
    float acc[2] = {0.f, 0.f};   // taps j0 and j0 + 32
    for (int d0 = 0; d0 < dh; d0 += 64) {
        const int dwd = min(64, dh - d0);
        __syncthreads();
        if (vec8) {
            for (int i = tid; i < (RW_ROWS + taps - 1) * 8; i += 256) {
                const int rr = i >> 3, dd = (i & 7) * 8;
                const int t = t0 - half + rr;
                float x8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (dd < dwd && t >= 0 && t < n_p) ld8(v + (long)b * v_bs + (long)t * ldv + h * dh + d0 + dd, x8);
#pragma unroll
                for (int e = 0; e < 8; e++) vs[rr * RW_PITCH + dd + e] = x8[e];
            }
            for (int i = tid; i < RW_ROWS * 8; i += 256) {
                const int rr = i >> 3, dd = (i & 7) * 8;
                const int t = t0 + rr;
                float x8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (dd < dwd && t < n_p) ld8(dout + (long)b * o_bs + (long)t * ldo + h * dh + d0 + dd, x8);
#pragma unroll
                for (int e = 0; e < 8; e++) ds[rr * RW_PITCH + dd + e] = x8[e];
            }
        } else {
            for (int i = tid; i < (RW_ROWS + taps - 1) * 64; i += 256) {
                const int rr = i >> 6, dd = i & 63;
                const int t = t0 - half + rr;
                vs[rr * RW_PITCH + dd] = (dd < dwd && t >= 0 && t < n_p) ? ldf(v + (long)b * v_bs + (long)t * ldv + h * dh + d0 + dd) : 0.f;
            }
            for (int i = tid; i < RW_ROWS * 64; i += 256) {
                const int rr = i >> 6, dd = i & 63;
                const int t = t0 + rr;
                ds[rr * RW_PITCH + dd] = (dd < dwd && t < n_p) ? ldf(dout + (long)b * o_bs + (long)t * ldo + h * dh + d0 + dd) : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int j = j0 + 32 * pass;
            if (j < taps) {
                float s = 0.f;
                for (int rr = 0; rr < RW_ROWS; rr++) {
                    const rf4 g0 = *reinterpret_cast<const rf4*>(ds + rr * RW_PITCH + q * 8);
                    const rf4 g1 = *reinterpret_cast<const rf4*>(ds + rr * RW_PITCH + q * 8 + 4);
                    const rf4 x0 = *reinterpret_cast<const rf4*>(vs + (rr + j) * RW_PITCH + q * 8);
                    const rf4 x1 = *reinterpret_cast<const rf4*>(vs + (rr + j) * RW_PITCH + q * 8 + 4);
                    s += g0[0] * x0[0] + g0[1] * x0[1] + g0[2] * x0[2] + g0[3] * x0[3] + g1[0] * x1[0] + g1[1] * x1[1] +
                         g1[2] * x1[2] + g1[3] * x1[3];
                }
                acc[pass] += s;
            }
        }
    }
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        float s = acc[pass];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        const int j = j0 + 32 * pass;
        if (q == 0 && j < taps) atomicAdd(dw + h * taps + j, s);
    }
}

extern "C" int mh_resconv_wgrad(const void* v, int64_t ldv, int64_t v_bs, const void* dout, int64_t ldo, int64_t o_bs,
                                float* dw, int B, int n_p, int heads, int dh, int taps, int dt_v, int dt_o, mh_stream s) {
    MH_REQUIRE(taps >= 1 && taps <= 63 && (taps & 1), "mh_resconv_wgrad: taps=%d unsupported", taps);
    if (B == 0 || n_p == 0) return MH_OK;
    if (resconv_mfma_on() && resconv_wgrad_try_mfma(v, ldv, v_bs, dout, ldo, o_bs, dw, B, n_p, heads, dh, taps, dt_v, dt_o, (hipStream_t)s)) {
        MH_LAUNCH_CHECK("mh_resconv_wgrad");
        return MH_OK;
    }
    dim3 grid(heads, mh_cdiv(n_p, RW_ROWS), B);
#define RW(TV, TO) hipLaunchKernelGGL((resconv_wgrad_kernel<TV, TO>), grid, dim3(256), 0, (hipStream_t)s, (const TV*)v, (long)ldv, (long)v_bs, (const TO*)dout, (long)ldo, (long)o_bs, dw, n_p, dh, taps, vec8)
    const int vec8 = dh % 8 == 0 && ldv % 8 == 0 && v_bs % 8 == 0 && ldo % 8 == 0 && o_bs % 8 == 0 && ((uintptr_t)v & 15) == 0 && ((uintptr_t)dout & 15) == 0;
    if (dt_v == MH_F32 && dt_o == MH_F32) RW(float, float);
    else if (dt_v == MH_BF16 && dt_o == MH_BF16) RW(bf16_t, bf16_t);
    else if (dt_v == MH_BF16 && dt_o == MH_F32) RW(bf16_t, float);
    else RW(float, bf16_t);
#undef RW
    MH_LAUNCH_CHECK("mh_resconv_wgrad");
    return MH_OK;
}

extern "C" int64_t mh_resconv_bwd_workspace_bytes(int B, int n_p, int heads, int dh, int taps) {
    return 4 * (int64_t)resconv_bwd_part_floats(B, n_p, heads, dh, taps);
}

extern "C" int mh_resconv_bwd(const void* dout, int64_t ldo, int64_t o_bs, const void* v, int64_t ldv, int64_t v_bs, const float* w, void* dv,
                              int64_t lddv, int64_t dv_bs, float* dw, float* workspace, int64_t ws_floats, int B, int n_p, int heads, int dh,
                              int taps, int dt_v, int dt_o, mh_stream s) {
    MH_REQUIRE(taps >= 1 && taps <= 63 && (taps & 1), "mh_resconv_bwd: taps=%d unsupported", taps);
    MH_REQUIRE(dout && v && w && dv && dw, "mh_resconv_bwd: null pointer");
    if (B == 0 || n_p == 0) return MH_OK;
    if (resconv_mfma_on() && dt_v == dt_o &&
        resconv_bwd_try_mfma(dout, ldo, o_bs, v, ldv, v_bs, w, dv, lddv, dv_bs, dw, workspace, ws_floats, B, n_p, heads, dh, taps, dt_v, (hipStream_t)s)) {
        MH_LAUNCH_CHECK("mh_resconv_bwd");
        return MH_OK;
    }
    // other dtypes / geometries, or no workspace: the two passes
    if (int e = mh_resconv_wgrad(v, ldv, v_bs, dout, ldo, o_bs, dw, B, n_p, heads, dh, taps, dt_v, dt_o, s)) return e;
    return mh_resconv_fwd(dout, ldo, o_bs, w, dv, lddv, dv_bs, B, n_p, heads, dh, taps, 1, 1, dt_o, dt_v, s);
}

// ------------------------------------------------------------------ pinv initial scaling
// stats64[0] = max over (bh,i) of sum_j |x[bh,i,j]| packed as (float bits << 32 | flat row index bh*m+i)
// stats64[1] = max over (bh,j) of sum_i |x[bh,i,j]| packed likewise (flat index bh*m+j). Values are >= 0 so
// the integer order of the packed words equals the float order. Caller zeroes stats64 before the call.
__global__ __launch_bounds__(256) void pinv_absmax_kernel(const float* __restrict__ x, unsigned long long* stats, int m) {
    const int bh = blockIdx.x;
    const float* xb = x + (long)bh * m * m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long best_r = 0, best_c = 0;
    __shared__ unsigned long long sr[256], sc[256];
    __shared__ __attribute__((aligned(16))) float colp[4 * 512];   // [4 waves][m] partial column sums
    if ((m & 3) == 0 && m <= 512) {
        // one pass, 16-byte loads, 8 rows in flight per wave: lane owns columns [256 c + 4 lane, +4), c = 0, 1, of the rows its
        // wave walks; row sums close with a wave reduction, column sums stay per lane and are folded over the 4 waves in LDS
        // (one scalar-load pass per sum took 106 us for 33 MB at m = 256 and 450 us at m = 384, the template's landmarks)
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 cs[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const bool two = m > 256;
        const int c0 = 4 * lane, c1 = 256 + 4 * lane;
        for (int i0 = wave; i0 < m; i0 += 32) {
            f4 v[8][2];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = min(i0 + 4 * u, m - 1);
                v[u][0] = *reinterpret_cast<const f4*>(xb + (long)i * m + min(c0, m - 4));
                v[u][1] = two ? *reinterpret_cast<const f4*>(xb + (long)i * m + min(c1, m - 4)) : f4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool row_ok = i0 + 4 * u < m;
                f4 a = {fabsf(v[u][0][0]), fabsf(v[u][0][1]), fabsf(v[u][0][2]), fabsf(v[u][0][3])};
                f4 b = {fabsf(v[u][1][0]), fabsf(v[u][1][1]), fabsf(v[u][1][2]), fabsf(v[u][1][3])};
                if (!row_ok || c0 >= m) a = f4{0.f, 0.f, 0.f, 0.f};
                if (!row_ok || c1 >= m) b = f4{0.f, 0.f, 0.f, 0.f};
                cs[0] += a;
                cs[1] += b;
                const float sv = wave_sum(a[0] + a[1] + a[2] + a[3] + b[0] + b[1] + b[2] + b[3]);
                if (row_ok) {
                    const unsigned long long p = ((unsigned long long)__float_as_uint(sv) << 32) | (unsigned)(bh * m + i0 + 4 * u);
                    best_r = p > best_r ? p : best_r;
                }
            }
        }
        if (c0 < m) *reinterpret_cast<f4*>(colp + wave * 512 + c0) = cs[0];
        if (c1 < m) *reinterpret_cast<f4*>(colp + wave * 512 + c1) = cs[1];
        __syncthreads();
        for (int j = threadIdx.x; j < m; j += 256) {
            const float sv = colp[j] + colp[512 + j] + colp[1024 + j] + colp[1536 + j];
            const unsigned long long p = ((unsigned long long)__float_as_uint(sv) << 32) | (unsigned)(bh * m + j);
            best_c = p > best_c ? p : best_c;
        }
    } else {
        // row sums: one wave per row
        for (int i = wave; i < m; i += 4) {
            float sv = 0.f;
            for (int j = lane; j < m; j += 64) sv += fabsf(xb[(long)i * m + j]);
            sv = wave_sum(sv);
            const unsigned long long p = ((unsigned long long)__float_as_uint(sv) << 32) | (unsigned)(bh * m + i);
            best_r = p > best_r ? p : best_r;
        }
        // column sums: one thread per column (coalesced across threads)
        for (int j = threadIdx.x; j < m; j += 256) {
            float sv = 0.f;
            for (int i = 0; i < m; i++) sv += fabsf(xb[(long)i * m + j]);
            const unsigned long long p = ((unsigned long long)__float_as_uint(sv) << 32) | (unsigned)(bh * m + j);
            best_c = p > best_c ? p : best_c;
        }
    }
    sr[threadIdx.x] = best_r;
    sc[threadIdx.x] = best_c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            if (sr[threadIdx.x + o] > sr[threadIdx.x]) sr[threadIdx.x] = sr[threadIdx.x + o];
            if (sc[threadIdx.x + o] > sc[threadIdx.x]) sc[threadIdx.x] = sc[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(stats + 0, sr[0]);
        atomicMax(stats + 1, sc[0]);
    }
}

extern "C" int mh_pinv_absmax(const float* x, uint64_t* stats64, int BH, int m, mh_stream s) {
    MH_REQUIRE((long)BH * m < (1L << 31), "mh_pinv_absmax: index overflow");
    if (BH == 0) return MH_OK;
    hipLaunchKernelGGL(pinv_absmax_kernel, dim3(BH), dim3(256), 0, (hipStream_t)s, x, (unsigned long long*)stats64, m);
    MH_LAUNCH_CHECK("mh_pinv_absmax");
    return MH_OK;
}

__device__ __forceinline__ float stat_val(const unsigned long long* st, int k) { return __uint_as_float((unsigned)(st[k] >> 32)); }

// z0[bh,i,j] = x[bh,j,i] / (c*r): 32x32 LDS transpose tiles
__global__ __launch_bounds__(256) void pinv_z0_kernel(const float* __restrict__ x, const unsigned long long* __restrict__ st,
                                                      float* __restrict__ z0, int m) {
    __shared__ float tile[32][33];
    const float inv = 1.f / (stat_val(st, 0) * stat_val(st, 1));
    const long base = (long)blockIdx.z * m * m;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int j0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = j0 + k, c = i0 + tx;  // read x[r][c]
        tile[k][tx] = (r < m && c < m) ? x[base + (long)r * m + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int i = i0 + k, j = j0 + tx;  // write z0[i][j] = x[j][i]
        if (i < m && j < m) z0[base + (long)i * m + j] = tile[tx][k] * inv;
    }
}

extern "C" int mh_pinv_z0(const float* x, const uint64_t* stats64, float* z0, int BH, int m, mh_stream s) {
    if (BH == 0) return MH_OK;
    dim3 grid(mh_cdiv(m, 32), mh_cdiv(m, 32), BH);
    hipLaunchKernelGGL(pinv_z0_kernel, grid, dim3(256), 0, (hipStream_t)s, x, (const unsigned long long*)stats64, z0, m);
    MH_LAUNCH_CHECK("mh_pinv_z0");
    return MH_OK;
}

// adjoint of z0 = x^T/(c r): dx[j][i] += dz0[i][j]/(c r);  S = sum dz0*z0 accumulated into scratch1[0]
__global__ __launch_bounds__(256) void pinv_z0_bwd_kernel(const float* __restrict__ z0, const float* __restrict__ dz0,
                                                          const unsigned long long* __restrict__ st, float* __restrict__ dx,
                                                          float* __restrict__ scratch, int m) {
    __shared__ float tile[32][33];
    __shared__ float red[4];
    const float inv = 1.f / (stat_val(st, 0) * stat_val(st, 1));
    const long base = (long)blockIdx.z * m * m;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    float dot = 0.f;
    for (int k = ty; k < 32; k += 8) {
        const int i = i0 + k, j = j0 + tx;
        float d = 0.f;
        if (i < m && j < m) { d = dz0[base + (long)i * m + j]; dot += d * z0[base + (long)i * m + j]; }
        tile[k][tx] = d;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int j = j0 + k, i = i0 + tx;  // dx[j][i] += dz0[i][j] * inv
        if (i < m && j < m) dx[base + (long)j * m + i] += tile[tx][k] * inv;
    }
    dot = block_sum256(dot, red);
    if (threadIdx.x == 0) atomicAdd(scratch, dot);
}

// m % 64 == 0: 64 x 64 tiles, 16-byte loads and stores (the 32 x 32 scalar tiles ran at 1.2 TB/s)
__global__ __launch_bounds__(256) void pinv_z0_bwd_vec_kernel(const float* __restrict__ z0, const float* __restrict__ dz0,
                                                              const unsigned long long* __restrict__ st, float* __restrict__ dx,
                                                              float* __restrict__ scratch, int m, const float* __restrict__ x) {
    typedef float zf4 __attribute__((ext_vector_type(4)));
    __shared__ float tile[64][65];
    __shared__ float red[4];
    const float inv = 1.f / (stat_val(st, 0) * stat_val(st, 1));
    const long base = (long)blockIdx.z * m * m;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int c4 = (threadIdx.x & 15) * 4, r = threadIdx.x >> 4;   // 16 float4 per row, 16 rows per pass
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = i0 + r + 16 * k;
        const zf4 d = *reinterpret_cast<const zf4*>(dz0 + base + (long)i * m + j0 + c4);
        if (z0) {
            const zf4 z = *reinterpret_cast<const zf4*>(z0 + base + (long)i * m + j0 + c4);
            dot += d[0] * z[0] + d[1] * z[1] + d[2] * z[2] + d[3] * z[3];
        }
#pragma unroll
        for (int e = 0; e < 4; e++) tile[r + 16 * k][c4 + e] = d[e];
    }
    zf4 o[4], xs[4];                               // the dx (and x) tiles are requested before the barrier, beside the input tiles
#pragma unroll
    for (int k = 0; k < 4; k++) {
        o[k] = *reinterpret_cast<const zf4*>(dx + base + (long)(j0 + r + 16 * k) * m + i0 + c4);
        if (!z0) xs[k] = *reinterpret_cast<const zf4*>(x + base + (long)(j0 + r + 16 * k) * m + i0 + c4);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = j0 + r + 16 * k;             // dx[j][i0 + c4 ..] += dz0[i0 + c4 ..][j] * inv
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float d = tile[c4 + e][r + 16 * k];
            o[k][e] += d * inv;
            if (!z0) dot += d * (xs[k][e] * inv);  // z0[i][j] = x[j][i] / (c r): no stored z0 needed (mh_nys_sim2 path)
        }
        *reinterpret_cast<zf4*>(dx + base + (long)j * m + i0 + c4) = o[k];
    }
    dot = block_sum256(dot, red);
    if (threadIdx.x == 0) atomicAdd(scratch, dot);
}

// sub-gradients through the two torch.max(): d(c r) = -S/(c r); dc = d(cr)*r -> row i*: dx += dc*sign(x);
// dr = d(cr)*c -> column j*: dx += dr*sign(x)
__global__ void pinv_max_bwd_kernel(const float* __restrict__ x, const unsigned long long* __restrict__ st,
                                    const float* __restrict__ scratch, float* __restrict__ dx, int m) {
    const float c = stat_val(st, 0), r = stat_val(st, 1);
    const float dcr = -scratch[0] / (c * r);
    const unsigned ri = (unsigned)(st[0] & 0xffffffffu), ci = (unsigned)(st[1] & 0xffffffffu);
    const long rbase = (long)(ri / m) * m * m + (long)(ri % m) * m;  // row i* of matrix bh*
    const long cbase = (long)(ci / m) * m * m + (ci % m);            // column j* of matrix bh'
    for (int k = threadIdx.x; k < m; k += blockDim.x) {
        const float xv = x[rbase + k];
        atomicAdd(dx + rbase + k, dcr * r * (xv > 0.f ? 1.f : (xv < 0.f ? -1.f : 0.f)));
        const float xc = x[cbase + (long)k * m];
        atomicAdd(dx + cbase + (long)k * m, dcr * c * (xc > 0.f ? 1.f : (xc < 0.f ? -1.f : 0.f)));
    }
}

extern "C" int mh_pinv_z0_bwd(const float* x, const float* z0, const float* dz0, const uint64_t* stats64, float* dx,
                              float* scratch1, int scratch_zeroed, int BH, int m, mh_stream s) {
    if (BH == 0) return MH_OK;
    if (!scratch_zeroed) {      // a caller that carves scratch1 from a buffer it clears once per step saves this memset node
        hipError_t e = hipMemsetAsync(scratch1, 0, sizeof(float), (hipStream_t)s);
        if (e != hipSuccess) { mh_set_error("mh_pinv_z0_bwd: memset failed"); return MH_EHIP; }
    }
    MH_REQUIRE(z0 || (m % 64 == 0 && (((uintptr_t)x | (uintptr_t)dz0 | (uintptr_t)dx) & 15) == 0), "mh_pinv_z0_bwd: z0 == NULL needs m %% 64 == 0 and aligned buffers");
    if (m % 64 == 0 && (((uintptr_t)z0 | (uintptr_t)dz0 | (uintptr_t)dx | (uintptr_t)x) & 15) == 0) {
        hipLaunchKernelGGL(pinv_z0_bwd_vec_kernel, dim3(m / 64, m / 64, BH), dim3(256), 0, (hipStream_t)s, z0, dz0,
                           (const unsigned long long*)stats64, dx, scratch1, m, x);
    } else {
        dim3 grid(mh_cdiv(m, 32), mh_cdiv(m, 32), BH);
        hipLaunchKernelGGL(pinv_z0_bwd_kernel, grid, dim3(256), 0, (hipStream_t)s, z0, dz0, (const unsigned long long*)stats64, dx, scratch1, m);
    }
    hipLaunchKernelGGL(pinv_max_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, x, (const unsigned long long*)stats64, scratch1, dx, m);
    MH_LAUNCH_CHECK("mh_pinv_z0_bwd");
    return MH_OK;
}

// ---- attn2's whole backward tail in one pass (x = softmax output of [3P] sim2, no key-padding mask, m = 256) ------------------------
// What mh_pinv_z0_bwd + the softmax backward do in three passes over four [BH, m, m] f32 tensors (each on the chain's stream, i.e.
// on the critical path of the backward window):   dP = dX + dz0^T / (c r);   dS = P o (dP - rowsum(P o dP)).
// The sub-gradients through the two torch.max() of moore_penrose_iter_pinv's start add g_r to ONE row and g_c to ONE column of dP
// (of the matrices that hold the maxima).  After a softmax backward a constant added to a row of dP cancels (rows of P sum to one:
// what is left is g_r (1 - rowsum) ~ 1e-7 g_r, the rounding of the sum), so only the column term survives, as a rank-one correction
// of one matrix:  dS[i][j] += g_c P[i][j] ([j == j*] - P[i][j*]),  applied by pinv_s2_colfix_kernel once the dot product
// S = sum dz0 o z0 (g_c = -S / (c r) * c) is complete.
constexpr int S2_ROWS = 32;                 // rows of dS per workgroup
constexpr int S2_PITCH = 260;               // floats: 16-byte aligned rows, the column writes of the transposed tile hit 64 distinct banks
// mlm (nullable, [BH / heads, 256]; BASELINE config 4): valid-landmark flags — P came out of a masked_fill + softmax, whose backward passes
// nothing to a filled entry (an invalid row or column): dS is zeroed there (a fully masked row is uniform, not zero, in P)
__global__ __launch_bounds__(256) void pinv_s2_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dz0,
                                                          const unsigned long long* __restrict__ st, float* __restrict__ dx,
                                                          float* __restrict__ scratch, const float* __restrict__ mlm, int heads) {
    typedef float zf4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float T[S2_ROWS * S2_PITCH];      // T[i][j] = dz0[j][i0 + i]
    __shared__ float red[4];
    const float inv = 1.f / (stat_val(st, 0) * stat_val(st, 1));
    const long base = (long)blockIdx.y * 256 * 256;
    const int i0 = blockIdx.x * S2_ROWS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this wave's 8 rows of P and dX are requested before the transposed tile is built
    zf4 p[8], g[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const long off = base + (long)(i0 + 8 * wave + r) * 256 + 4 * lane;
        p[r] = *reinterpret_cast<const zf4*>(P + off);
        g[r] = *reinterpret_cast<const zf4*>(dx + off);
    }
    {
        const float* src = dz0 + base + (long)tid * 256 + i0;                   // row j = tid, 32 consecutive i: 128 bytes
        zf4 d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = *reinterpret_cast<const zf4*>(src + 4 * k);
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int e = 0; e < 4; e++) T[(4 * k + e) * S2_PITCH + tid] = d[k][e];
    }
    __syncthreads();
    float dot = 0.f;
    const float* mb = mlm ? mlm + (long)(blockIdx.y / heads) * 256 : nullptr;
    zf4 cv = {1.f, 1.f, 1.f, 1.f};
    if (mb) cv = *reinterpret_cast<const zf4*>(mb + 4 * lane);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const zf4 t = *reinterpret_cast<const zf4*>(T + (8 * wave + r) * S2_PITCH + 4 * lane);
        const bool rv = !mb || mb[i0 + 8 * wave + r] != 0.f;
        zf4 dp;
        float rs = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            dp[e] = g[r][e] + t[e] * inv;
            dot += t[e] * (p[r][e] * inv);                  // z0[j][i] = x[i][j] / (c r)
            rs += p[r][e] * dp[e];
        }
        rs = wave_sum(rs);
        zf4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = (rv && cv[e] != 0.f) ? p[r][e] * (dp[e] - rs) : 0.f;
        *reinterpret_cast<zf4*>(dx + base + (long)(i0 + 8 * wave + r) * 256 + 4 * lane) = o;
    }
    dot = block_sum256(dot, red);
    if (tid == 0) atomicAdd(scratch, dot);
}

// the column maximum's sub-gradient through the softmax backward (see above): one wave per row of the matrix that holds it
__global__ __launch_bounds__(256) void pinv_s2_colfix_kernel(const float* __restrict__ P, const unsigned long long* __restrict__ st,
                                                             const float* __restrict__ scratch, float* __restrict__ dx,
                                                             const float* __restrict__ mlm, int heads) {
    typedef float zf4 __attribute__((ext_vector_type(4)));
    const float c = stat_val(st, 0), r = stat_val(st, 1);
    const float gc = -scratch[0] / (c * r) * c;
    const unsigned ci = (unsigned)(st[1] & 0xffffffffu);
    const long base = (long)(ci / 256) * 256 * 256;
    const int js = ci % 256, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const long off = base + (long)i * 256 + 4 * lane;
    const float pj = P[base + (long)i * 256 + js];
    const zf4 p = *reinterpret_cast<const zf4*>(P + off);
    zf4 d = *reinterpret_cast<const zf4*>(dx + off);
    const float* mb = mlm ? mlm + (long)((ci / 256) / heads) * 256 : nullptr;
    const bool rv = !mb || mb[i] != 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++)
        if (rv && (!mb || mb[4 * lane + e] != 0.f)) d[e] += gc * p[e] * ((4 * lane + e == js ? 1.f : 0.f) - pj);
    *reinterpret_cast<zf4*>(dx + off) = d;
}

extern "C" int mh_pinv_s2_bwd(const float* p, const float* dz0, const uint64_t* stats64, float* dx, float* scratch1,
                              int scratch_zeroed, int BH, int m, const float* mlm, int heads, mh_stream s) {
    MH_REQUIRE(!mlm || (heads >= 1 && BH % heads == 0 && ((uintptr_t)mlm & 15) == 0), "mh_pinv_s2_bwd: mlm [BH / heads, m] f32, 16-byte aligned");
    MH_REQUIRE(m == 256, "mh_pinv_s2_bwd: built for m = 256 (m=%d): compose mh_pinv_z0_bwd + mh_softmax_bwd", m);
    MH_REQUIRE(p && dz0 && stats64 && dx && scratch1 && (((uintptr_t)p | (uintptr_t)dz0 | (uintptr_t)dx) & 15) == 0,
               "mh_pinv_s2_bwd: null / unaligned buffer");
    if (BH == 0) return MH_OK;
    if (!scratch_zeroed) {
        hipError_t e = hipMemsetAsync(scratch1, 0, sizeof(float), (hipStream_t)s);
        if (e != hipSuccess) { mh_set_error("mh_pinv_s2_bwd: memset failed"); return MH_EHIP; }
    }
    hipLaunchKernelGGL(pinv_s2_bwd_kernel, dim3(256 / S2_ROWS, BH), dim3(256), 0, (hipStream_t)s, p, dz0, (const unsigned long long*)stats64, dx,
                       scratch1, mlm, heads > 0 ? heads : 1);
    hipLaunchKernelGGL(pinv_s2_colfix_kernel, dim3(64), dim3(256), 0, (hipStream_t)s, p, (const unsigned long long*)stats64, scratch1, dx, mlm,
                       heads > 0 ? heads : 1);
    MH_LAUNCH_CHECK("mh_pinv_s2_bwd");
    return MH_OK;
}

__global__ __launch_bounds__(256) void eye_minus_kernel(const float* __restrict__ P, float* __restrict__ T, float d, long total, int m) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int j = idx % m;
        const int i = (idx / m) % m;
        T[idx] = (i == j ? d : 0.f) - P[idx];
    }
}

extern "C" int mh_eye_minus(const float* P, float* T, float d, int BH, int m, mh_stream s) {
    const long total = (long)BH * m * m;
    if (total == 0) return MH_OK;
    dim3 grid((unsigned)min((long)mh_cdiv(total, 256), 8192L));
    hipLaunchKernelGGL(eye_minus_kernel, grid, dim3(256), 0, (hipStream_t)s, P, T, d, total, m);
    MH_LAUNCH_CHECK("mh_eye_minus");
    return MH_OK;
}

// ------------------------------------------------------------------ TransMIL sequence glue
// seq [B, n=1+N+add, D]; rows 1..N were written by the _fc1 GEMM epilogue.
template <typename T>
__global__ __launch_bounds__(256) void seq_finish_kernel(T* seq, const float* __restrict__ cls, int N, int add, int D) {
    const int n = 1 + N + add;
    const long b = blockIdx.y;
    const long total = (long)(1 + add) * D;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx % D;
        const int r = idx / D;  // 0 -> cls, 1.. -> pad row r-1
        T* sb = seq + b * n * D;
        if (r == 0) stf(sb + c, cls[c]);
        else sb[(long)(N + r) * D + c] = sb[(long)r * D + c];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void seq_finish_bwd_kernel(T* dseq, float* __restrict__ dcls, int B, int N, int add, int D) {
    const int n = 1 + N + add;
    const long total = (long)(1 + add) * D;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = idx % D;
        const int r = idx / D;
        if (r == 0) {
            float s = 0.f;
            for (int b = 0; b < B; b++) s += ldf(dseq + (long)b * n * D + c);
            dcls[c] += s;
        } else {
            for (int b = 0; b < B; b++) {
                T* sb = dseq + (long)b * n * D;
                stf(sb + (long)r * D + c, ldf(sb + (long)r * D + c) + ldf(sb + (long)(N + r) * D + c));
            }
        }
    }
}

extern "C" int mh_seq_finish(void* seq, const float* cls, int B, int N, int add, int D, int dt, mh_stream s) {
    MH_REQUIRE(add >= 0 && add <= N, "mh_seq_finish: add=%d out of range", add);
    if (B == 0) return MH_OK;
    dim3 grid((unsigned)min((long)mh_cdiv((long)(1 + add) * D, 256), 1024L), B);
    MH_DISPATCH_DT(dt, T, hipLaunchKernelGGL((seq_finish_kernel<T>), grid, dim3(256), 0, (hipStream_t)s, (T*)seq, cls, N, add, D));
    MH_LAUNCH_CHECK("mh_seq_finish");
    return MH_OK;
}

extern "C" int mh_seq_finish_bwd(void* dseq, float* dcls, int B, int N, int add, int D, int dt, mh_stream s) {
    MH_REQUIRE(add >= 0 && add <= N, "mh_seq_finish_bwd: add=%d out of range", add);
    if (B == 0) return MH_OK;
    dim3 grid((unsigned)min((long)mh_cdiv((long)(1 + add) * D, 256), 1024L));
    MH_DISPATCH_DT(dt, T, hipLaunchKernelGGL((seq_finish_bwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)s, (T*)dseq, dcls, B, N, add, D));
    MH_LAUNCH_CHECK("mh_seq_finish_bwd");
    return MH_OK;
}
