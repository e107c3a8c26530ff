// BASELINE config 5: fp8 (OCP e4m3) MFMA operands for the forward projections of the WSI encoder (W1 `_fc1`, W4 `to_qkv`,
// W10 `to_out`, W14 retention embed / head; models/mirror.py:346, [3P] NystromAttention.to_qkv / to_out, :595-607), f32
// accumulate, bf16 / f32 results; everything else (backward, pinv, softmax, losses) stays in the bf16 policy.
//
//   quantise:  amax = max |x| over the tensor;  q = fp8(x * 448 / amax);  scale = amax / 448          (per-tensor scaling)
//   product :  C = act(scale_a * scale_b * (Aq Bq^T) + bias),  v_mfma_f32_32x32x16_fp8_fp8
//
// A lane feeds the MFMA 8 consecutive k of one row as one 8-byte register pair — the bf16 32x32x16 fragment with bytes
// instead of halfwords — so the tile images are [rows][64 bytes of k] with an 80-byte pitch.  This first version uses a
// 128 x 128 x 64 tile with 4 waves (wave tile 64 x 64); it is the parity / plumbing implementation of config 5, not yet
// a tuned kernel (the 2x MFMA rate of CDNA4 for fp8 needs the 32x32x64 f8f6f4 instruction and the 256-tile pipeline).
#include <cstdlib>
#include "gemm_kernel.h"

bool gemm_try_big_fp8(GemmArgs& a, int dtC, int batch, hipStream_t s);   // gemm_big.hip: 256-tile pipeline, f8f6f4 MFMA

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int F8_T = 128, F8_BK = 64, F8_P = 80;      // tile rows / cols, K bytes per step, LDS pitch in bytes
constexpr float F8_MAX = 448.f;

// ---- per-tensor absolute maximum (f32 bits are monotone for non-negative values: atomicMax on the bit pattern)
template <typename T>
__global__ __launch_bounds__(256) void amax_kernel(const T* __restrict__ x, long n, unsigned* __restrict__ amax_bits) {
    __shared__ float red[4];
    float m = 0.f;
    const long n4 = n / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
        const f4_t v = ld4(x + 4 * q);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(ldf(x + i)));
    m = block_max256(m, red);
    if (threadIdx.x == 0) atomicMax(amax_bits, __float_as_uint(m));
}

// q = fp8_e4m3(clamp(x * 448 / amax)); scale[0] = amax / 448 (1 when the tensor is all zeros)
template <typename T>
__global__ __launch_bounds__(256) void quant_fp8_kernel(const T* __restrict__ x, long n, const unsigned* __restrict__ amax_bits,
                                                        unsigned char* __restrict__ q, float* __restrict__ scale) {
    const float amax = __uint_as_float(*amax_bits);
    const float mul = amax > 0.f ? F8_MAX / amax : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) scale[0] = amax > 0.f ? amax / F8_MAX : 1.f;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f4_t v = ld4(x + 4 * i) * mul;
        unsigned w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[0], -F8_MAX), F8_MAX), fminf(fmaxf(v[1], -F8_MAX), F8_MAX), w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[2], -F8_MAX), F8_MAX), fminf(fmaxf(v[3], -F8_MAX), F8_MAX), w, true);
        reinterpret_cast<unsigned*>(q)[i] = w;
    }
}

// Delayed scaling: ONE pass.  q = e4m3(clamp(x * 448 / amax_prev)) with amax_prev = this tensor's max |x| of the previous step,
// while this step's max is gathered for the next one.  `ring` is 3 uint32 per call site: slot t % 3 collects step t, slot
// (t - 1) % 3 is read, slot (t + 1) % 3 is cleared by one thread (nobody else touches it during step t) — no pass over the
// tensor just to learn its scale, no grid-wide ordering inside the kernel.  t comes from DEVICE memory (`tick`, the optimizer's
// step counter), so a replayed HIP graph rotates the ring by itself.  margin > 1 leaves headroom for growth from step to
// step; values past +-448 saturate.
template <typename T>
__global__ __launch_bounds__(256) void quant_fp8_delayed_kernel(const T* __restrict__ x, long n, unsigned* __restrict__ ring,
                                                                const float* __restrict__ tick, float margin,
                                                                unsigned char* __restrict__ q, float* __restrict__ scale) {
    __shared__ float red[4];
    const int t = (int)tick[0];
    const int cur = t % 3, prev = (t + 2) % 3, nxt = (t + 1) % 3;
    const float amax = __uint_as_float(ring[prev]) * margin;
    const float mul = amax > 0.f ? F8_MAX / amax : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scale[0] = amax > 0.f ? amax / F8_MAX : 1.f;
        ring[nxt] = 0u;
    }
    float m = 0.f;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f4_t raw = ld4(x + 4 * i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(raw[0]), fabsf(raw[1]))), fmaxf(fabsf(raw[2]), fabsf(raw[3])));
        const f4_t v = raw * mul;
        unsigned w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[0], -F8_MAX), F8_MAX), fminf(fmaxf(v[1], -F8_MAX), F8_MAX), w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[2], -F8_MAX), F8_MAX), fminf(fmaxf(v[3], -F8_MAX), F8_MAX), w, true);
        reinterpret_cast<unsigned*>(q)[i] = w;
    }
    m = block_max256(m, red);
    if (threadIdx.x == 0) atomicMax(ring + cur, __float_as_uint(m));
}

// C[M, N] = act(sa * sb * A[M, K] B[N, K]^T + bias); A, B fp8 with K contiguous (lda, ldb bytes), K % 64 == 0, N % 128 == 0
template <typename TC>
__global__ __launch_bounds__(256) void gemm_fp8_kernel(const unsigned char* __restrict__ A, long lda, long a_bs,
                                                       const unsigned char* __restrict__ B, long ldb, TC* __restrict__ C, long ldc, long c_bs,
                                                       const float* __restrict__ sa, const float* __restrict__ sb,
                                                       const float* __restrict__ bias, int act, int M, int N, int K) {
    A += (long)blockIdx.z * a_bs;      // batch of row windows (`to_out(x)[:, -n:]`, `_fc1` into rows 1..N of the sequence buffer)
    C += (long)blockIdx.z * c_bs;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][F8_T * F8_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hl = lane >> 5;
    const int row0 = blockIdx.y * F8_T, col0 = blockIdx.x * F8_T;
    // staging: 128 rows x 64 bytes per operand = 512 chunks of 16 bytes, two per thread; rows past M are clamped (never stored)
    u32x4 ra[2], rb[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int cid = tid + 256 * i, rr = cid >> 2, cc = cid & 3;
            ra[i] = *reinterpret_cast<const u32x4*>(A + (long)min(row0 + rr, M - 1) * lda + k0 + 16 * cc);
            rb[i] = *reinterpret_cast<const u32x4*>(B + (long)(col0 + rr) * ldb + k0 + 16 * cc);
        }
    };
    auto store = [&](int st) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int cid = tid + 256 * i, rr = cid >> 2, cc = cid & 3;
            *reinterpret_cast<u32x4*>(&smem[st][0][rr * F8_P + 16 * cc]) = ra[i];
            *reinterpret_cast<u32x4*>(&smem[st][1][rr * F8_P + 16 * cc]) = rb[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    const int nt = K / F8_BK;
    load(0);
    store(0);
    if (nt > 1) load(F8_BK);
    __syncthreads();
    for (int t = 0; t < nt; t++) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            store(cur ^ 1);
            if (t + 2 < nt) load((t + 2) * F8_BK);
        }
        const unsigned char* at = smem[cur][0];
        const unsigned char* bt = smem[cur][1];
#pragma unroll
        for (int ks = 0; ks < F8_BK; ks += 16) {
            long af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; i++) af[i] = *reinterpret_cast<const long*>(at + (wm * 64 + 32 * i + r) * F8_P + ks + 8 * hl);
#pragma unroll
            for (int j = 0; j < 2; j++) bf[j] = *reinterpret_cast<const long*>(bt + (wn * 64 + 32 * j + r) * F8_P + ks + 8 * hl);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C[i][j]: column = lane & 31, rows 8 (e >> 2) + 4 hl + (e & 3)
    const float sc = sa[0] * sb[0];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int col = col0 + wn * 64 + 32 * j + r;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = row0 + wm * 64 + 32 * i + 8 * (e >> 2) + 4 * hl + (e & 3);
                if (row < M) {
                    float v = acc[i][j][e] * sc + bv;
                    if (act == MH_ACT_RELU) v = fmaxf(v, 0.f);
                    stf(C + (long)row * ldc + col, v);
                }
            }
    }
}

}  // namespace

extern "C" int mh_quant_fp8(const void* x, int64_t n, void* q, float* scale, unsigned* amax_scratch, int dt, mh_stream s) {
    if (n == 0) return MH_OK;
    MH_REQUIRE(n % 4 == 0 && mh_quad_ok(x, mh_dt_size(dt)) && ((uintptr_t)q & 3) == 0, "mh_quant_fp8: n must be a multiple of 4, buffers quad-aligned");
    (void)hipMemsetAsync(amax_scratch, 0, sizeof(unsigned), (hipStream_t)s);
    dim3 grid((unsigned)min((long)mh_cdiv(n / 4, 256), 4096L));
    if (dt == MH_F32) {
        hipLaunchKernelGGL((amax_kernel<float>), grid, dim3(256), 0, (hipStream_t)s, (const float*)x, (long)n, amax_scratch);
        hipLaunchKernelGGL((quant_fp8_kernel<float>), grid, dim3(256), 0, (hipStream_t)s, (const float*)x, (long)n, amax_scratch, (unsigned char*)q, scale);
    } else {
        hipLaunchKernelGGL((amax_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (long)n, amax_scratch);
        hipLaunchKernelGGL((quant_fp8_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (long)n, amax_scratch, (unsigned char*)q, scale);
    }
    MH_LAUNCH_CHECK("mh_quant_fp8");
    return MH_OK;
}

extern "C" int mh_quant_fp8_delayed(const void* x, int64_t n, void* q, float* scale, unsigned* ring, const float* tick, float margin,
                                    int dt, mh_stream s) {
    if (n == 0) return MH_OK;
    MH_REQUIRE(n % 4 == 0 && mh_quad_ok(x, mh_dt_size(dt)) && ((uintptr_t)q & 3) == 0, "mh_quant_fp8_delayed: n must be a multiple of 4, buffers quad-aligned");
    MH_REQUIRE(ring && tick && margin >= 1.f, "mh_quant_fp8_delayed: ring, tick and margin >= 1 are required");
    dim3 grid((unsigned)min((long)mh_cdiv(n / 4, 256), 4096L));
    if (dt == MH_F32)
        hipLaunchKernelGGL((quant_fp8_delayed_kernel<float>), grid, dim3(256), 0, (hipStream_t)s, (const float*)x, (long)n, ring, tick, margin, (unsigned char*)q, scale);
    else
        hipLaunchKernelGGL((quant_fp8_delayed_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (long)n, ring, tick, margin, (unsigned char*)q, scale);
    MH_LAUNCH_CHECK("mh_quant_fp8_delayed");
    return MH_OK;
}

extern "C" int mh_gemm_fp8(const void* A, int64_t lda, int64_t a_bs, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t c_bs,
                           int batch, const float* scale_a, const float* scale_b, const float* bias, int act, int M, int N, int K, int dt_c,
                           mh_stream s) {
    if (M == 0 || N == 0 || batch == 0) return MH_OK;
    MH_REQUIRE(batch >= 1 && batch <= 65535 && a_bs % 16 == 0, "mh_gemm_fp8: bad batch (%d) or batch stride", batch);
    MH_REQUIRE(K >= F8_BK && K % F8_BK == 0 && N % F8_T == 0, "mh_gemm_fp8: needs K %% 64 == 0 and N %% 128 == 0 (got N=%d K=%d)", N, K);
    MH_REQUIRE(lda % 16 == 0 && ldb % 16 == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0, "mh_gemm_fp8: operands must be 16-byte aligned");
    MH_REQUIRE(act == MH_ACT_NONE || act == MH_ACT_RELU, "mh_gemm_fp8: activation %d unsupported", act);
    // the large-tile pipeline (v_mfma_scale_f32_32x32x64_f8f6f4: twice the bf16 MFMA rate) when the shape fits it
    if (K % 128 == 0 && N % 256 == 0 && lda % 2 == 0 && ldb % 2 == 0 && a_bs % 2 == 0) {
        GemmArgs a = {};
        a.A = A; a.B = B; a.C = C; a.bias = bias;
        a.M = M; a.N = N; a.K = K / 2;
        a.lda = lda / 2; a.ldb = ldb / 2; a.ldc = ldc;
        a.sA1 = a_bs / 2; a.sC1 = c_bs; a.batch2 = 1;
        a.alpha = 1.f; a.act = act; a.split_k = 1; a.k_per_split = K / 2;
        a.vecA = a.vecB = 1;
        const int cvec = dt_c == MH_F32 ? 4 : 8;
        a.vecC = (((uintptr_t)C & 15) == 0) && ldc % cvec == 0 && c_bs % cvec == 0;
        a.scale_a = scale_a; a.scale_b = scale_b;
        if (gemm_try_big_fp8(a, dt_c, batch, (hipStream_t)s)) {
            MH_LAUNCH_CHECK("mh_gemm_fp8");
            return MH_OK;
        }
    }
    dim3 grid(N / F8_T, mh_cdiv(M, F8_T), batch);
    if (dt_c == MH_F32)
        hipLaunchKernelGGL((gemm_fp8_kernel<float>), grid, dim3(256), 0, (hipStream_t)s, (const unsigned char*)A, (long)lda, (long)a_bs,
                           (const unsigned char*)B, (long)ldb, (float*)C, (long)ldc, (long)c_bs, scale_a, scale_b, bias, act, M, N, K);
    else
        hipLaunchKernelGGL((gemm_fp8_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)s, (const unsigned char*)A, (long)lda, (long)a_bs,
                           (const unsigned char*)B, (long)ldb, (bf16_t*)C, (long)ldc, (long)c_bs, scale_a, scale_b, bias, act, M, N, K);
    MH_LAUNCH_CHECK("mh_gemm_fp8");
    return MH_OK;
}
