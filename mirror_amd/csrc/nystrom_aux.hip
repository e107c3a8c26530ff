// Nystrom-attention side kernels ([3P] nystrom_attention, called at models/mirror.py:312) and the
// TransMIL sequence glue (models/mirror.py:657-665): landmark means, 33-tap residual conv of V,
// pseudo-inverse initial scaling (tensor-wide max) and its adjoint, d*I - P, cls/square-pad rows.
#include "common.h"

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

extern "C" int mh_landmark_fwd(const void* qkv, void* lm, int B, int n_p, int D, int l, int dt, mh_stream s) {
    MH_REQUIRE(l >= 1 && n_p % l == 0, "mh_landmark_fwd: n_p=%d not a multiple of l=%d", n_p, l);
    const long total = (long)B * (n_p / l) * 2 * D;
    if (total == 0) return MH_OK;
    dim3 grid((unsigned)min((long)mh_cdiv(total, 256), 8192L));
    MH_DISPATCH_DT(dt, T, hipLaunchKernelGGL((landmark_fwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)s, (const T*)qkv, (T*)lm, B, n_p, D, l));
    MH_LAUNCH_CHECK("mh_landmark_fwd");
    return MH_OK;
}

extern "C" int mh_landmark_bwd(const void* dlm, void* dqkv, int B, int n_p, int D, int l, int dt, mh_stream s) {
    MH_REQUIRE(l >= 1 && n_p % l == 0, "mh_landmark_bwd: n_p=%d not a multiple of l=%d", n_p, l);
    const long total = (long)B * n_p * 2 * D;
    if (total == 0) return MH_OK;
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

extern "C" int mh_resconv_fwd(const void* v, int64_t ldv, int64_t v_bs, const float* w, void* out, int64_t ldo,
                              int64_t o_bs, int B, int n_p, int heads, int dh, int taps, int transpose, int accumulate,
                              int dt_v, int dt_o, mh_stream s) {
    MH_REQUIRE(taps >= 1 && taps <= RC_MAXTAPS && (taps & 1), "mh_resconv_fwd: taps=%d unsupported", taps);
    const int C = heads * dh;
    MH_REQUIRE(heads <= 8 || 255 / dh + 2 <= 8, "mh_resconv_fwd: more than 8 heads per 256 columns (heads=%d dh=%d)", heads, dh);
    if (B == 0 || n_p == 0) return MH_OK;
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

// dw[h][j] += sum_{b,t,d} dout[b,t,h,d] * v[b,t+j-half,h,d]. Block = (head h, chunk of rows, b); thread (tt, d-lane).
#define RW_ROWS 64
template <typename TV, typename TO>
__global__ __launch_bounds__(256) void resconv_wgrad_kernel(const TV* __restrict__ v, long ldv, long v_bs,
                                                            const TO* __restrict__ dout, long ldo, long o_bs,
                                                            float* __restrict__ dw, int n_p, int dh, int taps) {
    // LDS: v rows [RW_ROWS + taps - 1][dh_chunk=64] and dout rows [RW_ROWS][64]
    __shared__ float vs[(RW_ROWS + RC_MAXTAPS) * 64];
    __shared__ float ds[RW_ROWS * 64];
    const int h = blockIdx.x, t0 = blockIdx.y * RW_ROWS, b = blockIdx.z;
    const int half = taps / 2;
    const int tid = threadIdx.x;
    // each thread accumulates tap j = tid % 64 (if < taps) over a quarter of the rows: simple 2-D split
    const int j = tid & 63, quarter = tid >> 6;
    float acc = 0.f;
    for (int d0 = 0; d0 < dh; d0 += 64) {
        const int dw_ = min(64, dh - d0);
        __syncthreads();
        for (int i = tid; i < (RW_ROWS + taps - 1) * 64; i += 256) {
            const int rr = i / 64, dd = i % 64;
            const int t = t0 - half + rr;
            vs[i] = (dd < dw_ && t >= 0 && t < n_p) ? ldf(v + (long)b * v_bs + (long)t * ldv + h * dh + d0 + dd) : 0.f;
        }
        for (int i = tid; i < RW_ROWS * 64; i += 256) {
            const int rr = i / 64, dd = i % 64;
            const int t = t0 + rr;
            ds[i] = (dd < dw_ && t < n_p) ? ldf(dout + (long)b * o_bs + (long)t * ldo + h * dh + d0 + dd) : 0.f;
        }
        __syncthreads();
        if (j < taps) {
            for (int rr = quarter; rr < RW_ROWS; rr += 4) {
                const float* dr = ds + rr * 64;
                const float* vr = vs + (rr + j) * 64;
                float s = 0.f;
                for (int dd = 0; dd < 64; dd++) s += dr[(dd + j) & 63] * vr[(dd + j) & 63];  // skewed: no bank conflicts
                acc += s;
            }
        }
    }
    // sum the 4 quarters per tap
    __shared__ float accs[4][64];
    accs[quarter][j] = acc;
    __syncthreads();
    if (quarter == 0 && j < taps) atomicAdd(dw + h * taps + j, accs[0][j] + accs[1][j] + accs[2][j] + accs[3][j]);
}

extern "C" int mh_resconv_wgrad(const void* v, int64_t ldv, int64_t v_bs, const void* dout, int64_t ldo, int64_t o_bs,
                                float* dw, int B, int n_p, int heads, int dh, int taps, int dt_v, int dt_o, mh_stream s) {
    MH_REQUIRE(taps >= 1 && taps <= 63 && (taps & 1), "mh_resconv_wgrad: taps=%d unsupported", taps);
    if (B == 0 || n_p == 0) return MH_OK;
    dim3 grid(heads, mh_cdiv(n_p, RW_ROWS), B);
#define RW(TV, TO) hipLaunchKernelGGL((resconv_wgrad_kernel<TV, TO>), grid, dim3(256), 0, (hipStream_t)s, (const TV*)v, (long)ldv, (long)v_bs, (const TO*)dout, (long)ldo, (long)o_bs, dw, n_p, dh, taps)
    if (dt_v == MH_F32 && dt_o == MH_F32) RW(float, float);
    else if (dt_v == MH_BF16 && dt_o == MH_BF16) RW(bf16_t, bf16_t);
    else if (dt_v == MH_BF16 && dt_o == MH_F32) RW(bf16_t, float);
    else RW(float, bf16_t);
#undef RW
    MH_LAUNCH_CHECK("mh_resconv_wgrad");
    return MH_OK;
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
    // row sums: one wave per row
    for (int i = wave; i < m; i += 4) {
        float s = 0.f;
        for (int j = lane; j < m; j += 64) s += fabsf(xb[(long)i * m + j]);
        s = wave_sum(s);
        const unsigned long long p = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)(bh * m + i);
        best_r = p > best_r ? p : best_r;
    }
    // column sums: one thread per column (coalesced across threads)
    for (int j = threadIdx.x; j < m; j += 256) {
        float s = 0.f;
        for (int i = 0; i < m; i++) s += fabsf(xb[(long)i * m + j]);
        const unsigned long long p = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)(bh * m + j);
        best_c = p > best_c ? p : best_c;
    }
    __shared__ unsigned long long sr[256], sc[256];
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
                              float* scratch1, int BH, int m, mh_stream s) {
    if (BH == 0) return MH_OK;
    hipError_t e = hipMemsetAsync(scratch1, 0, sizeof(float), (hipStream_t)s);
    if (e != hipSuccess) { mh_set_error("mh_pinv_z0_bwd: memset failed"); return MH_EHIP; }
    dim3 grid(mh_cdiv(m, 32), mh_cdiv(m, 32), BH);
    hipLaunchKernelGGL(pinv_z0_bwd_kernel, grid, dim3(256), 0, (hipStream_t)s, z0, dz0, (const unsigned long long*)stats64, dx, scratch1, m);
    hipLaunchKernelGGL(pinv_max_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, x, (const unsigned long long*)stats64, scratch1, dx, m);
    MH_LAUNCH_CHECK("mh_pinv_z0_bwd");
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
