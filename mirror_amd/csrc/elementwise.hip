// Elementwise and small reduction kernels: add / cast / GELU / ReLU-bwd / Philox dropout / column sums,
// MAE-style masking (rank select + token fill + positional add), reparameterisation, RNA heads-axis attention.
#include "common.h"

#define EW_GRID(n) dim3((unsigned)min((long)mh_cdiv((n), 256), 16384L))
#define EW_LOOP(i, n) for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (n); i += (long)gridDim.x * 256)

template <typename TA, typename TB, typename TY>
__global__ __launch_bounds__(256) void add_kernel(const TA* a, const TB* b, TY* y, long n) {
    EW_LOOP(i, n) stf(y + i, ldf(a + i) + ldf(b + i));
}
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void cast_kernel(const TX* x, TY* y, long n) {
    EW_LOOP(i, n) stf(y + i, ldf(x + i));
}
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const TX* x, TY* y, long n) {
    EW_LOOP(i, n) stf(y + i, gelu_f(ldf(x + i)));
}
template <typename TX, typename TDY, typename TDX>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const TX* x, const TDY* dy, TDX* dx, long n) {
    EW_LOOP(i, n) stf(dx + i, ldf(dy + i) * gelu_grad_f(ldf(x + i)));
}
template <typename TY, typename TDY, typename TDX>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const TY* y, const TDY* dy, TDX* dx, long npb, long y_bs, long dy_bs, long dx_bs) {
    const long b = blockIdx.y;
    EW_LOOP(i, npb) stf(dx + b * dx_bs + i, ldf(y + b * y_bs + i) > 0.f ? ldf(dy + b * dy_bs + i) : 0.f);
}

// quad forms (n % 4 == 0, quad-aligned pointers): 16-byte f32 / 8-byte bf16 accesses
#define EW4_LOOP(q, n4) for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < (n4); q += (long)gridDim.x * 256)
template <typename TA, typename TB, typename TY>
__global__ __launch_bounds__(256) void add4_kernel(const TA* a, const TB* b, TY* y, long n4) {
    EW4_LOOP(q, n4) st4(y + 4 * q, ld4(a + 4 * q) + ld4(b + 4 * q));
}
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void cast4_kernel(const TX* x, TY* y, long n4) {
    EW4_LOOP(q, n4) st4(y + 4 * q, ld4(x + 4 * q));
}
template <typename TY, typename TDY, typename TDX>
__global__ __launch_bounds__(256) void relu_bwd4_kernel(const TY* y, const TDY* dy, TDX* dx, long npb4, long y_bs, long dy_bs, long dx_bs) {
    const long b = blockIdx.y;
    EW4_LOOP(q, npb4) {
        const f4_t yv = ld4(y + b * y_bs + 4 * q), g = ld4(dy + b * dy_bs + 4 * q);
        f4_t o = {yv[0] > 0.f ? g[0] : 0.f, yv[1] > 0.f ? g[1] : 0.f, yv[2] > 0.f ? g[2] : 0.f, yv[3] > 0.f ? g[3] : 0.f};
        st4(dx + b * dx_bs + 4 * q, o);
    }
}

#define DISPATCH2(dt0, dt1, MACRO)                                     \
    if ((dt0) == MH_F32 && (dt1) == MH_F32) { MACRO(float, float); }   \
    else if ((dt0) == MH_F32) { MACRO(float, bf16_t); }                \
    else if ((dt1) == MH_F32) { MACRO(bf16_t, float); }                \
    else { MACRO(bf16_t, bf16_t); }

extern "C" int mh_add(const void* a, const void* b, void* y, int64_t n, int dt_a, int dt_b, int dt_y, mh_stream s) {
    if (n == 0) return MH_OK;
    if (n % 4 == 0 && mh_quad_ok(a, mh_dt_size(dt_a)) && mh_quad_ok(b, mh_dt_size(dt_b)) && mh_quad_ok(y, mh_dt_size(dt_y))) {
#define ADD4_(TA, TB)                                                                                                    \
    if (dt_y == MH_F32) hipLaunchKernelGGL((add4_kernel<TA, TB, float>), EW_GRID(n / 4), dim3(256), 0, (hipStream_t)s, (const TA*)a, (const TB*)b, (float*)y, (long)(n / 4)); \
    else hipLaunchKernelGGL((add4_kernel<TA, TB, bf16_t>), EW_GRID(n / 4), dim3(256), 0, (hipStream_t)s, (const TA*)a, (const TB*)b, (bf16_t*)y, (long)(n / 4))
        DISPATCH2(dt_a, dt_b, ADD4_)
#undef ADD4_
        MH_LAUNCH_CHECK("mh_add");
        return MH_OK;
    }
#define ADD_(TA, TB)                                                                                                     \
    if (dt_y == MH_F32) hipLaunchKernelGGL((add_kernel<TA, TB, float>), EW_GRID(n), dim3(256), 0, (hipStream_t)s, (const TA*)a, (const TB*)b, (float*)y, (long)n); \
    else hipLaunchKernelGGL((add_kernel<TA, TB, bf16_t>), EW_GRID(n), dim3(256), 0, (hipStream_t)s, (const TA*)a, (const TB*)b, (bf16_t*)y, (long)n)
    DISPATCH2(dt_a, dt_b, ADD_)
#undef ADD_
    MH_LAUNCH_CHECK("mh_add");
    return MH_OK;
}

// The landmark gradient of a Nystrom layer written where to_qkv's backward reads it: out[r, 0:cols] = a[r] + b[r] (f32 partial sums of
// the attention kernels and of sim2's products, b may be NULL) rounded to bf16 at row stride out_ld, and out[r, cols:cols + zero_cols] = 0
// (the v columns of the landmark rows that sit behind the sequence in the [rows, 3D] gradient buffer: landmarks have no v).
__global__ __launch_bounds__(256) void lm_merge_kernel(const float* __restrict__ a, const float* __restrict__ b, bf16_t* __restrict__ out, long rows,
                                                       int cols4, int tot4, long out_ld) {
    EW4_LOOP(q, rows * tot4) {
        const long r = q / tot4;
        const int c = (int)(q - r * tot4);
        f4_t v = {0.f, 0.f, 0.f, 0.f};
        if (c < cols4) {
            v = ld4(a + (r * cols4 + c) * 4);
            if (b) v += ld4(b + (r * cols4 + c) * 4);
        }
        st4(out + r * out_ld + 4 * c, v);
    }
}

extern "C" int mh_lm_merge(const float* a, const float* b, void* out, int64_t rows, int cols, int64_t out_ld, int zero_cols, mh_stream s) {
    MH_REQUIRE(a && out && rows >= 0 && cols > 0 && cols % 4 == 0 && zero_cols >= 0 && zero_cols % 4 == 0 && out_ld >= cols + zero_cols && out_ld % 4 == 0,
               "mh_lm_merge: cols, zero_cols, out_ld must be multiples of 4 with out_ld >= cols + zero_cols");
    MH_REQUIRE(mh_quad_ok(a, 4) && (!b || mh_quad_ok(b, 4)) && mh_quad_ok(out, 2), "mh_lm_merge: unaligned buffer");
    if (rows == 0) return MH_OK;
    const int tot4 = (cols + zero_cols) / 4;
    hipLaunchKernelGGL(lm_merge_kernel, EW_GRID(rows * tot4), dim3(256), 0, (hipStream_t)s, a, b, (bf16_t*)out, (long)rows, cols / 4, tot4, (long)out_ld);
    MH_LAUNCH_CHECK("mh_lm_merge");
    return MH_OK;
}

extern "C" int mh_cast(const void* x, void* y, int64_t n, int dt_x, int dt_y, mh_stream s) {
    if (n == 0) return MH_OK;
    if (n % 4 == 0 && mh_quad_ok(x, mh_dt_size(dt_x)) && mh_quad_ok(y, mh_dt_size(dt_y))) {
#define CAST4_(TX, TY) hipLaunchKernelGGL((cast4_kernel<TX, TY>), EW_GRID(n / 4), dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, (long)(n / 4))
        DISPATCH2(dt_x, dt_y, CAST4_)
#undef CAST4_
        MH_LAUNCH_CHECK("mh_cast");
        return MH_OK;
    }
#define CAST_(TX, TY) hipLaunchKernelGGL((cast_kernel<TX, TY>), EW_GRID(n), dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, (long)n)
    DISPATCH2(dt_x, dt_y, CAST_)
#undef CAST_
    MH_LAUNCH_CHECK("mh_cast");
    return MH_OK;
}

extern "C" int mh_gelu_fwd(const void* x, void* y, int64_t n, int dt_x, int dt_y, mh_stream s) {
    if (n == 0) return MH_OK;
#define GELU_(TX, TY) hipLaunchKernelGGL((gelu_fwd_kernel<TX, TY>), EW_GRID(n), dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, (long)n)
    DISPATCH2(dt_x, dt_y, GELU_)
#undef GELU_
    MH_LAUNCH_CHECK("mh_gelu_fwd");
    return MH_OK;
}

extern "C" int mh_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dt_x, int dt_dy, int dt_dx, mh_stream s) {
    MH_REQUIRE(dt_dy == dt_dx, "mh_gelu_bwd: dy/dx dtype mismatch");
    if (n == 0) return MH_OK;
#define GELUB_(TX, TD) hipLaunchKernelGGL((gelu_bwd_kernel<TX, TD, TD>), EW_GRID(n), dim3(256), 0, (hipStream_t)s, (const TX*)x, (const TD*)dy, (TD*)dx, (long)n)
    DISPATCH2(dt_x, dt_dy, GELUB_)
#undef GELUB_
    MH_LAUNCH_CHECK("mh_gelu_bwd");
    return MH_OK;
}

extern "C" int mh_relu_bwd(const void* y, const void* dy, void* dx, int64_t n_per_batch, int batches, int64_t y_bs,
                           int64_t dy_bs, int64_t dx_bs, int dt_y, int dt_dy, int dt_dx, mh_stream s) {
    if (n_per_batch == 0 || batches == 0) return MH_OK;
    MH_REQUIRE(batches <= 65535, "mh_relu_bwd: too many batches");
    if (n_per_batch % 4 == 0 && y_bs % 4 == 0 && dy_bs % 4 == 0 && dx_bs % 4 == 0 && mh_quad_ok(y, mh_dt_size(dt_y)) &&
        mh_quad_ok(dy, mh_dt_size(dt_dy)) && mh_quad_ok(dx, mh_dt_size(dt_dx))) {
        dim3 g4((unsigned)min((long)mh_cdiv(n_per_batch / 4, 256), 4096L), batches);
#define RELUB42_(TY, TDY)                                                                                                  \
    if (dt_dx == MH_F32) hipLaunchKernelGGL((relu_bwd4_kernel<TY, TDY, float>), g4, dim3(256), 0, (hipStream_t)s, (const TY*)y, (const TDY*)dy, (float*)dx, (long)(n_per_batch / 4), (long)y_bs, (long)dy_bs, (long)dx_bs); \
    else hipLaunchKernelGGL((relu_bwd4_kernel<TY, TDY, bf16_t>), g4, dim3(256), 0, (hipStream_t)s, (const TY*)y, (const TDY*)dy, (bf16_t*)dx, (long)(n_per_batch / 4), (long)y_bs, (long)dy_bs, (long)dx_bs)
#define RELUB4_(TY, TD) RELUB42_(TY, TD)
        DISPATCH2(dt_y, dt_dy, RELUB4_)
#undef RELUB42_
#undef RELUB4_
        MH_LAUNCH_CHECK("mh_relu_bwd");
        return MH_OK;
    }
    dim3 grid((unsigned)min((long)mh_cdiv(n_per_batch, 256), 4096L), batches);
#define RELUB2_(TY, TDY)                                                                                                   \
    if (dt_dx == MH_F32) hipLaunchKernelGGL((relu_bwd_kernel<TY, TDY, float>), grid, dim3(256), 0, (hipStream_t)s, (const TY*)y, (const TDY*)dy, (float*)dx, (long)n_per_batch, (long)y_bs, (long)dy_bs, (long)dx_bs); \
    else hipLaunchKernelGGL((relu_bwd_kernel<TY, TDY, bf16_t>), grid, dim3(256), 0, (hipStream_t)s, (const TY*)y, (const TDY*)dy, (bf16_t*)dx, (long)n_per_batch, (long)y_bs, (long)dy_bs, (long)dx_bs)
#define RELUB_(TY, TD) RELUB2_(TY, TD)
    DISPATCH2(dt_y, dt_dy, RELUB_)
#undef RELUB2_
#undef RELUB_
    MH_LAUNCH_CHECK("mh_relu_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------ Philox4x32-10 dropout
// philox4x32_10: common.h

// element i uses word (i & 3) of the Philox block with counter (offset + i) >> 2: the mask is a pure
// function of (seed, offset, i), so the backward pass regenerates it instead of storing it.
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void dropout_kernel(const TX* x, TY* y, long n, float p, uint64_t seed, uint64_t offset,
                                                      const uint64_t* __restrict__ dev_base) {
    if (dev_base) offset += *dev_base & ~3ull;     // per-step base kept on the device (graph replays draw fresh masks)
    const float scale = 1.f / (1.f - p);
    const uint32_t thr = (uint32_t)fminf(p * 4294967296.f, 4294967295.f);
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q * 4 < n; q += (long)gridDim.x * 256) {
        const uint64_t blk = (offset >> 2) + (uint64_t)q;
        uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u};
        philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const long i = q * 4 + e;
            if (i < n) stf(y + i, ctr[e] >= thr ? ldf(x + i) * scale : 0.f);
        }
    }
}

// The step's four random draws in ONE launch on the dropout stream (models/mirror.py:630, :516, :832-833: torch.rand(B, N), torch.rand(B, D),
// torch.randn(B, L) twice): out[0, n_uniform) uniform in [0, 1) with 24 random bits like torch.rand's f32, out[n_uniform, + n_normal)
// standard normal (Box-Muller on word pairs of a block).  n_uniform % 4 == 0.  Element i = word (i & 3) of block (offset + i) >> 2.
// As torch draws they were four launches plus, under a captured graph, two generator-state fills in front of every replay.
__global__ __launch_bounds__(256) void noise_draws_kernel(float* __restrict__ out, long n_uniform, long n_total, uint64_t seed, uint64_t offset,
                                                          const uint64_t* __restrict__ dev_base) {
    if (dev_base) offset += *dev_base & ~3ull;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q * 4 < n_total; q += (long)gridDim.x * 256) {
        const uint64_t blk = (offset >> 2) + (uint64_t)q;
        uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u};
        philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
        float v[4];
        if (q * 4 < n_uniform) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (float)(ctr[e] >> 8) * 5.9604644775390625e-8f;        // k / 2^24, k < 2^24
        } else {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float u1 = ((float)(ctr[2 * h] >> 8) + 1.f) * 5.9604644775390625e-8f;           // (0, 1]
                const float u2 = (float)(ctr[2 * h + 1] >> 8) * 5.9604644775390625e-8f;
                const float r = sqrtf(-2.f * logf(u1));
                float sn, cs;
                sincosf(6.283185307179586f * u2, &sn, &cs);
                v[2 * h] = r * cs;
                v[2 * h + 1] = r * sn;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (q * 4 + e < n_total) out[q * 4 + e] = v[e];
    }
}

extern "C" int mh_noise_draws(float* out, int64_t n_uniform, int64_t n_normal, uint64_t seed, uint64_t offset, const uint64_t* dev_base,
                              mh_stream s) {
    MH_REQUIRE(out && n_uniform >= 0 && n_normal >= 0 && n_uniform % 4 == 0 && (offset & 3) == 0, "mh_noise_draws: n_uniform and offset are multiples of 4");
    const long n = n_uniform + n_normal;
    if (n == 0) return MH_OK;
    hipLaunchKernelGGL(noise_draws_kernel, dim3((unsigned)min((long)mh_cdiv(mh_cdiv(n, 4), 256), 2048L)), dim3(256), 0, (hipStream_t)s, out, (long)n_uniform, n, seed,
                       offset, dev_base);
    MH_LAUNCH_CHECK("mh_noise_draws");
    return MH_OK;
}

// quad form (n % 4 == 0, quad-aligned pointers): one Philox block per thread iteration = one 8 / 16-byte access per tensor;
// ADD: y = a + dropout(x) (the residual add behind to_out's Dropout, models/mirror.py:312 + [3P] to_out[1])
template <typename TX, typename TY, bool ADD>
__global__ __launch_bounds__(256) void dropout4_kernel(const TX* x, const float* a, TY* y, long n4, float p, uint64_t seed, uint64_t offset,
                                                       const uint64_t* __restrict__ dev_base) {
    if (dev_base) offset += *dev_base & ~3ull;
    const float scale = 1.f / (1.f - p);
    const uint32_t thr = (uint32_t)fminf(p * 4294967296.f, 4294967295.f);
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
        const uint64_t blk = (offset >> 2) + (uint64_t)q;
        uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u};
        philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
        const f4_t v = ld4(x + 4 * q);
        f4_t o = {ctr[0] >= thr ? v[0] * scale : 0.f, ctr[1] >= thr ? v[1] * scale : 0.f, ctr[2] >= thr ? v[2] * scale : 0.f,
                  ctr[3] >= thr ? v[3] * scale : 0.f};
        if (ADD) o += ld4(a + 4 * q);
        st4(y + 4 * q, o);
    }
}

// the lite stream (common.h drop16_*): 8 elements per Philox block, n % 8 == 0, 16 / 32-byte accesses per tensor
template <typename TX, typename TY, bool ADD>
__global__ __launch_bounds__(256) void dropout8_kernel(const TX* x, const float* a, TY* y, long n8, float p, uint64_t seed, uint64_t offset,
                                                       const uint64_t* __restrict__ dev_base) {
    if (dev_base) offset += *dev_base & ~7ull;
    const uint32_t thr = drop16_thr(p);
    const float scale = drop16_scale(thr);
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n8; q += (long)gridDim.x * 256) {
        const uint32_t keep = drop16_keep8((offset >> 3) + (uint64_t)q, seed, thr);
        const f4_t v0 = ld4(x + 8 * q), v1 = ld4(x + 8 * q + 4);
        f4_t o0, o1;
#pragma unroll
        for (int e = 0; e < 4; e++) {           // __fmul_rn / __fadd_rn: scale and residual add round separately (never an fma), so the
            o0[e] = (keep & (1u << e)) ? __fmul_rn(v0[e], scale) : 0.f;      // DROPADD projection epilogue reproduces this bit for bit
            o1[e] = (keep & (16u << e)) ? __fmul_rn(v1[e], scale) : 0.f;
        }
        if (ADD) {
            const f4_t a0 = ld4(a + 8 * q), a1 = ld4(a + 8 * q + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) { o0[e] = __fadd_rn(a0[e], o0[e]); o1[e] = __fadd_rn(a1[e], o1[e]); }
        }
        st4(y + 8 * q, o0);
        st4(y + 8 * q + 4, o1);
    }
}

// y = dropout(x) (a == NULL) or y = a + dropout(x) on the lite stream; n % 8 == 0, offset % 8 == 0, 16-byte aligned buffers
extern "C" int mh_dropout_lite(const float* a, const void* x, void* y, int64_t n, float p, uint64_t seed, uint64_t offset,
                               const uint64_t* dev_base, int dt_x, int dt_y, mh_stream s) {
    MH_REQUIRE(p >= 0.f && p < 1.f, "mh_dropout_lite: p=%f out of range", (double)p);
    MH_REQUIRE((offset & 7) == 0 && n % 8 == 0, "mh_dropout_lite: offset and n must be multiples of 8");
    MH_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)a) & 15) == 0, "mh_dropout_lite: buffers must be 16-byte aligned");
    MH_REQUIRE(!a || dt_y == MH_F32, "mh_dropout_lite: the residual form writes f32");
    if (n == 0) return MH_OK;
#define DROP8_(TX, TY, ADD) hipLaunchKernelGGL((dropout8_kernel<TX, TY, ADD>), EW_GRID(n / 8), dim3(256), 0, (hipStream_t)s, (const TX*)x, a, (TY*)y, (long)(n / 8), p, seed, offset, dev_base)
    if (a) { if (dt_x == MH_F32) DROP8_(float, float, true); else DROP8_(bf16_t, float, true); }
    else if (dt_x == MH_F32 && dt_y == MH_F32) DROP8_(float, float, false);
    else if (dt_x == MH_F32) DROP8_(float, bf16_t, false);
    else if (dt_y == MH_F32) DROP8_(bf16_t, float, false);
    else DROP8_(bf16_t, bf16_t, false);
#undef DROP8_
    MH_LAUNCH_CHECK("mh_dropout_lite");
    return MH_OK;
}

// dropout backward of a [rows, N] gradient on the lite stream (f32 in, bf16 out) that also leaves the column sums of what it wrote:
// the bias gradient of the Linear in front of the dropout ([3P] to_out = Sequential(Linear, Dropout), models/mirror.py:312) without
// a second pass over the bf16 gradient.  Thread (cg = tid % (N / 8), rl = tid / (N / 8)) owns 8 columns and every (256 / (N / 8))-th
// row of this workgroup's rows; column sums are folded through LDS and added to db with one atomic per column and workgroup.
__global__ __launch_bounds__(256) void dropout8_colsum_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long rows, int N, int rows_per_wg,
                                                              float p, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ dev_base,
                                                              float* __restrict__ db) {
    __shared__ float red[256 * 8];
    if (dev_base) offset += *dev_base & ~7ull;
    const uint32_t thr = drop16_thr(p);
    const float scale = drop16_scale(thr);
    const int groups = N >> 3, lanes = 256 / groups, cg = threadIdx.x % groups, rl = threadIdx.x / groups;
    const long r0 = (long)blockIdx.x * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;                         // rows in flight per thread
    auto body = [&](long rb, bool full) {
        f4_t v0[U], v1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long r = rb + (long)u * lanes;
            const long q = (full || r < r1 ? r : r1 - 1) * groups + cg;          // clamped: the value is not used
            v0[u] = ld4(x + 8 * q);
            v1[u] = ld4(x + 8 * q + 4);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long r = rb + (long)u * lanes;
            const bool on = full || r < r1;
            const long q = (on ? r : r1 - 1) * groups + cg;
            const uint32_t keep = drop16_keep8((offset >> 3) + (uint64_t)q, seed, thr);
            f4_t o0, o1;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                o0[e] = (keep & (1u << e)) ? __fmul_rn(v0[u][e], scale) : 0.f;
                o1[e] = (keep & (16u << e)) ? __fmul_rn(v1[u][e], scale) : 0.f;
            }
            if (on) {
                st4(y + 8 * q, o0);
                st4(y + 8 * q + 4, o1);
#pragma unroll
                for (int e = 0; e < 4; e++) {    // the sums are those of the ROUNDED values (what mh_colsum reads back)
                    acc[e] += bf2f(f2bf(o0[e]));
                    acc[4 + e] += bf2f(f2bf(o1[e]));
                }
            }
        }
    };
    long rb = r0 + rl;
    for (; rb + (long)(U - 1) * lanes < r1; rb += (long)U * lanes) body(rb, true);
    if (rb < r1) body(rb, false);
#pragma unroll
    for (int e = 0; e < 8; e++) red[rl * N + 8 * cg + e] = acc[e];
    __syncthreads();
    // one atomic per column and workgroup, consecutive columns in consecutive lanes (a lane adding its own 8 columns in turn spreads
    // every wave-instruction over 16 cache lines: 4.5x slower end to end)
    for (int c = threadIdx.x; c < N; c += 256) {
        float t = 0.f;
        for (int k = 0; k < lanes; k++) t += red[k * N + c];
        if (t != 0.f) atomicAdd(db + c, t);
    }
}

extern "C" int mh_dropout_lite_colsum(const float* x, void* y, int64_t rows, int N, float p, uint64_t seed, uint64_t offset,
                                      const uint64_t* dev_base, float* db, mh_stream s) {
    MH_REQUIRE(p >= 0.f && p < 1.f, "mh_dropout_lite_colsum: p=%f out of range", (double)p);
    MH_REQUIRE(N >= 8 && N % 8 == 0 && N <= 2048 && 256 % (N / 8) == 0 && (offset & 7) == 0, "mh_dropout_lite_colsum: N=%d needs N %% 8 == 0 and 256 %% (N / 8) == 0", N);
    MH_REQUIRE(db && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "mh_dropout_lite_colsum: aligned buffers and a bias-gradient destination");
    if (rows == 0) return MH_OK;
    constexpr int rows_per_wg = 192;      // rows per workgroup: 8 in flight per thread, one coalesced atomic per column and workgroup (64: 46 us, 128: 38, 192: 36, 512: 44; the two launches: 47)
    hipLaunchKernelGGL(dropout8_colsum_kernel, dim3((unsigned)mh_cdiv(rows, rows_per_wg)), dim3(256), 0, (hipStream_t)s, x, (bf16_t*)y, (long)rows, N,
                       rows_per_wg, p, seed, offset, dev_base, db);
    MH_LAUNCH_CHECK("mh_dropout_lite_colsum");
    return MH_OK;
}

extern "C" int mh_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, uint64_t offset, const uint64_t* dev_base,
                          int dt_x, int dt_y, mh_stream s) {
    MH_REQUIRE(p >= 0.f && p < 1.f, "mh_dropout: p=%f out of range", (double)p);
    MH_REQUIRE((offset & 3) == 0, "mh_dropout: offset must be a multiple of 4");
    if (n == 0) return MH_OK;
    if (n % 4 == 0 && mh_quad_ok(x, mh_dt_size(dt_x)) && mh_quad_ok(y, mh_dt_size(dt_y))) {
#define DROP4_(TX, TY) hipLaunchKernelGGL((dropout4_kernel<TX, TY, false>), EW_GRID(n / 4), dim3(256), 0, (hipStream_t)s, (const TX*)x, (const float*)nullptr, (TY*)y, (long)(n / 4), p, seed, offset, dev_base)
        DISPATCH2(dt_x, dt_y, DROP4_)
#undef DROP4_
        MH_LAUNCH_CHECK("mh_dropout");
        return MH_OK;
    }
#define DROP_(TX, TY) hipLaunchKernelGGL((dropout_kernel<TX, TY>), EW_GRID((n + 3) / 4), dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, (long)n, p, seed, offset, dev_base)
    DISPATCH2(dt_x, dt_y, DROP_)
#undef DROP_
    MH_LAUNCH_CHECK("mh_dropout");
    return MH_OK;
}

extern "C" int mh_dropout_add(const float* a, const void* x, float* y, int64_t n, float p, uint64_t seed, uint64_t offset,
                              const uint64_t* dev_base, int dt_x, mh_stream s) {
    MH_REQUIRE(p >= 0.f && p < 1.f, "mh_dropout_add: p=%f out of range", (double)p);
    MH_REQUIRE((offset & 3) == 0, "mh_dropout_add: offset must be a multiple of 4");
    MH_REQUIRE(n % 4 == 0 && mh_quad_ok(x, mh_dt_size(dt_x)) && mh_quad_ok(a, 4) && mh_quad_ok(y, 4),
               "mh_dropout_add: n must be a multiple of 4 and the buffers quad-aligned");
    if (n == 0) return MH_OK;
    if (dt_x == MH_F32)
        hipLaunchKernelGGL((dropout4_kernel<float, float, true>), EW_GRID(n / 4), dim3(256), 0, (hipStream_t)s, (const float*)x, a, y, (long)(n / 4), p, seed, offset, dev_base);
    else
        hipLaunchKernelGGL((dropout4_kernel<bf16_t, float, true>), EW_GRID(n / 4), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, a, y, (long)(n / 4), p, seed, offset, dev_base);
    MH_LAUNCH_CHECK("mh_dropout_add");
    return MH_OK;
}

// ------------------------------------------------------------------ column sums (bias gradients)
// block = 64 columns x 4 row-slices; partial sums over a band of rows, then one f32 atomic per column.
#define CS_BAND 512
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long rows, int cols, long ld) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const long r0 = (long)blockIdx.y * CS_BAND, r1 = min(rows, r0 + CS_BAND);
    float s = 0.f;
    if (c < cols)
        for (long r = r0 + slice; r < r1; r += 4) s += ldf(x + r * ld + c);
    red[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && c < cols) atomicAdd(out + c, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

// bf16, cols % 8 == 0, 16-byte aligned rows: a lane owns 8 columns (one 16-byte load per row), a wave covers 512 columns of
// a row and keeps 8 rows in flight; the four waves' partials meet in LDS and leave as one coalesced f32 atomic per column.
// (The 2-byte-per-lane form above reads at 1.7 TB/s; the bias gradients of the [65 k x 512] projections are 7 such passes.)
__global__ __launch_bounds__(256) void colsum_bf16_vec_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, long rows, int cols,
                                                              long ld, long band) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 512 + 8 * lane;
    const long r0 = (long)blockIdx.y * band, r1 = min(rows, r0 + band);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typedef uint32_t cq __attribute__((ext_vector_type(4)));
    if (c0 < cols) {
        for (long r = r0 + wave; r < r1; r += 32) {
            cq v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const long ru = r + 4 * u;
                v[u] = ru < r1 ? *reinterpret_cast<const cq*>(x + ru * ld + c0) : (cq){0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    s[2 * w] += __uint_as_float(v[u][w] << 16);
                    s[2 * w + 1] += __uint_as_float(v[u][w] & 0xffff0000u);
                }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) red[wave][8 * lane + e] = s[e];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int j = threadIdx.x + 256 * k, c = blockIdx.x * 512 + j;
        if (c < cols) {
            const float t = red[0][j] + red[1][j] + red[2][j] + red[3][j];
            if (t != 0.f) atomicAdd(out + c, t);
        }
    }
}

// f32 rows of quads (a table of per-block partial sums, an f32 gradient): lane = one quad of columns, the rows dealt to gridDim.y x 4
// waves with eight 16-byte loads in flight each, one atomic per column and workgroup.  (colsum_kernel<float> walks a 512 x 512 table
// in 8 workgroups of serial 4-byte loads: 32 us where this takes ~5.)
__global__ __launch_bounds__(256) void colsum_f32_vec_kernel(const float* __restrict__ x, float* __restrict__ out, long rows, int cols, long ld) {
    typedef float cf4 __attribute__((ext_vector_type(4)));
    __shared__ cf4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const bool live = c < cols;
    const long chunks = (long)gridDim.y * 4, per = (rows + chunks - 1) / chunks;
    const long r0 = ((long)blockIdx.y * 4 + wave) * per, r1 = min(rows, r0 + per);
    cf4 sum = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        long r = r0;
        for (; r + 8 <= r1; r += 8) {
            cf4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const cf4*>(x + (r + u) * ld + c);
#pragma unroll
            for (int u = 0; u < 8; u++) sum += v[u];
        }
        for (; r < r1; r++) sum += *reinterpret_cast<const cf4*>(x + r * ld + c);
    }
    red[wave][lane] = sum;
    __syncthreads();
    if (wave == 0 && live) {
        sum = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (sum[e] != 0.f) atomicAdd(out + c + e, sum[e]);
    }
}

extern "C" int mh_colsum(const void* x, float* out, int64_t rows, int cols, int64_t ld, int dt, mh_stream s) {
    if (rows == 0 || cols == 0) return MH_OK;
    if (dt == MH_F32 && cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0 && rows >= 32) {
        dim3 gv(mh_cdiv(cols / 4, 64), (unsigned)min((long)mh_cdiv(rows, 32), 64L));
        hipLaunchKernelGGL(colsum_f32_vec_kernel, gv, dim3(256), 0, (hipStream_t)s, (const float*)x, out, (long)rows, cols, (long)ld);
        MH_LAUNCH_CHECK("mh_colsum");
        return MH_OK;
    }
    if (dt == MH_BF16 && cols % 8 == 0 && ld % 8 == 0 && ((uintptr_t)x & 15) == 0 && rows >= 1024) {
        const int cb = mh_cdiv(cols, 512);
        long nb = max(1L, 512L / cb);
        long band = (mh_cdiv(rows, nb) + 31) / 32 * 32;
        dim3 gv(cb, mh_cdiv(rows, band));
        hipLaunchKernelGGL(colsum_bf16_vec_kernel, gv, dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, out, (long)rows, cols, (long)ld, band);
        MH_LAUNCH_CHECK("mh_colsum");
        return MH_OK;
    }
    dim3 grid(mh_cdiv(cols, 64), mh_cdiv(rows, CS_BAND));
    MH_DISPATCH_DT(dt, T, hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, (hipStream_t)s, (const T*)x, out, (long)rows, cols, (long)ld));
    MH_LAUNCH_CHECK("mh_colsum");
    return MH_OK;
}

// ------------------------------------------------------------------ masking
// rank by ascending noise with index tie-break == argsort(argsort(noise)) of the reference for distinct values.
// One workgroup per row sorts the row's (order-preserving key, index) pairs in LDS (bitonic network over the next power of two, the
// tail padded with maximal keys) and scatters mask[index] = position >= len_keep.  The earlier all-pairs count was N^2 compares per
// row: 16 x 4096 took 256 workgroups for ~150-400 us beside the WSI encoder's first layers; the network is N log^2 N / 2.
__global__ __launch_bounds__(1024) void rank_mask_kernel(const float* __restrict__ noise, float* __restrict__ mask, int N, int P, int len_keep) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const long b = blockIdx.x;
    const float* nb = noise + b * N;
    for (int j = threadIdx.x; j < P; j += 1024) {
        unsigned long long k = ~0ull;
        if (j < N) {
            unsigned u = __float_as_uint(nb[j]);
            u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;      // total order of the floats as unsigned integers
            k = ((unsigned long long)u << 32) | (unsigned)j;
        }
        keys[j] = k;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += 1024) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const unsigned long long a = keys[lo], c = keys[hi];
                const bool up = (lo & k) == 0;
                if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
            }
            __syncthreads();
        }
    for (int j = threadIdx.x; j < N; j += 1024) mask[b * N + (unsigned)(keys[j] & 0xffffffffu)] = j >= len_keep ? 1.f : 0.f;
}

extern "C" int mh_rank_mask(const float* noise, float* mask, int B, int N, int len_keep, mh_stream s) {
    MH_REQUIRE(N >= 1 && N <= 16384, "mh_rank_mask: N=%d unsupported (max 16384)", N);
    if (B == 0) return MH_OK;
    int P = 2;
    while (P < N) P <<= 1;
    static const bool big = [] {       // above 64 KB of dynamic LDS the kernel needs the opt-in, once per process
        return hipFuncSetAttribute(reinterpret_cast<const void*>(rank_mask_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8) == hipSuccess;
    }();
    MH_REQUIRE(big || P * 8 <= 65536, "mh_rank_mask: %d bytes of LDS refused", P * 8);
    hipLaunchKernelGGL(rank_mask_kernel, dim3(B), dim3(1024), (size_t)P * sizeof(unsigned long long), (hipStream_t)s, noise, mask, N, P, len_keep);
    MH_LAUNCH_CHECK("mh_rank_mask");
    return MH_OK;
}

// x [B,T,D]: rows t >= first take the mask token where mask[b,t-first] != 0; every row gets + pos[t]
template <typename T, typename TY>
__global__ __launch_bounds__(256) void mask_apply_fwd_kernel(const T* x, TY* y, const float* __restrict__ mask,
                                                             const float* __restrict__ token, const float* __restrict__ pos, int B, int Tn,
                                                             int D, int first, int token_scalar) {
    const long total = (long)B * Tn * D;
    EW_LOOP(idx, total) {
        const int c = idx % D;
        const long bt = idx / D;
        const int t = bt % Tn;
        const long b = bt / Tn;
        float v = ldf(x + idx);
        if (t >= first && mask[b * (Tn - first) + (t - first)] != 0.f) v = token[token_scalar ? 0 : c];
        stf(y + idx, v + pos[(long)t * D + c]);
    }
}

// f32 out, D % 4 == 0, per-channel token: one (b, t) row per wave iteration, quad accesses, no per-element div / mod
typedef float mf_f4 __attribute__((ext_vector_type(4)));
template <typename TX>
__global__ __launch_bounds__(256) void mask_apply_fwd_vec_kernel(const TX* x, float* y, const float* __restrict__ mask,
                                                                 const float* __restrict__ token, const float* __restrict__ pos, int B,
                                                                 int Tn, int D, int first) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rows = (long)B * Tn;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const int t = (int)(row % Tn);
        const long b = row / Tn;
        const bool masked = t >= first && mask[b * (Tn - first) + (t - first)] != 0.f;
        for (int c = 4 * lane; c < D; c += 256) {
            const mf_f4 v = masked ? *reinterpret_cast<const mf_f4*>(token + c) : ld4(x + row * D + c);
            *reinterpret_cast<mf_f4*>(y + row * D + c) = v + *reinterpret_cast<const mf_f4*>(pos + (long)t * D + c);
        }
    }
}

// block = 64 columns x 4 row-slices over a band of MB_BAND rows t; each thread loops over b, so dpos needs no
// atomics and the mask-token gradient costs one f32 atomic per column per block.
#define MB_BAND 64
template <typename T, typename TDX>
__global__ __launch_bounds__(256) void mask_apply_bwd_kernel(const T* dy, TDX* dx, const float* __restrict__ mask,
                                                             float* __restrict__ dtoken, float* __restrict__ dpos, int B, int Tn, int D,
                                                             int first, int token_scalar) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int t0 = blockIdx.y * MB_BAND, t1 = min(Tn, t0 + MB_BAND);
    float st = 0.f;
    if (c < D) {
        for (int t = t0 + slice; t < t1; t += 4) {
            float sp = 0.f;
            for (int b = 0; b < B; b++) {
                const long at = ((long)b * Tn + t) * D + c;
                const float g = ldf(dy + at);
                sp += g;
                const bool msk = t >= first && mask[(long)b * (Tn - first) + (t - first)] != 0.f;
                if (msk) st += g;
                stf(dx + at, msk ? 0.f : g);
            }
            dpos[(long)t * D + c] += sp;
        }
    }
    red[slice][lane] = st;
    __syncthreads();
    if (slice == 0 && c < D) {
        const float tot = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (tot != 0.f) atomicAdd(dtoken + (token_scalar ? 0 : c), tot);
    }
}

// f32 in, D % 4 == 0: lane owns 4 columns (quad accesses), a wave owns one t at a time and has 8 batch rows in flight
typedef float mb_f4 __attribute__((ext_vector_type(4)));
template <typename TDX>
__global__ __launch_bounds__(256) void mask_apply_bwd_vec_kernel(const float* dy, TDX* dx, const float* __restrict__ mask,
                                                                 float* __restrict__ dtoken, float* __restrict__ dpos, int B, int Tn, int D,
                                                                 int first, int token_scalar, int band, float* __restrict__ dbias) {
    // dbias [D] (round 5, may be null): += the column sums of dx AS STORED (rounded to TDX, what mh_colsum over dx would read): the
    // bias gradient of the projection in front, without that pass
    __shared__ mb_f4 red[4][64];
    __shared__ mb_f4 reda[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + 4 * lane;
    const int t0 = blockIdx.y * band, t1 = min(Tn, t0 + band);
    mb_f4 st = {0.f, 0.f, 0.f, 0.f}, sa = {0.f, 0.f, 0.f, 0.f};
    if (c < D) {
        for (int t = t0 + wave; t < t1; t += 4) {
            mb_f4 sp = {0.f, 0.f, 0.f, 0.f};
            for (int b0 = 0; b0 < B; b0 += 8) {
                mb_f4 g[8];
                float mk[8];       // requested with the gradient rows: eight dependent mask loads per step were a latency chain
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (b0 + u < B) {
                        g[u] = *reinterpret_cast<const mb_f4*>(dy + ((long)(b0 + u) * Tn + t) * D + c);
                        mk[u] = t >= first ? mask[(long)(b0 + u) * (Tn - first) + (t - first)] : 0.f;
                    }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (b0 + u >= B) break;
                    sp += g[u];
                    const bool msk = mk[u] != 0.f;
                    if (msk) st += g[u];
                    else if constexpr (sizeof(TDX) == 2) {
#pragma unroll
                        for (int e = 0; e < 4; e++) sa[e] += bf2f(f2bf(g[u][e]));
                    } else sa += g[u];
                    if (msk || (const void*)dx != (const void*)dy)
                        st4(dx + ((long)(b0 + u) * Tn + t) * D + c, msk ? (mb_f4){0.f, 0.f, 0.f, 0.f} : g[u]);
                }
            }
            mb_f4* dp = reinterpret_cast<mb_f4*>(dpos + (long)t * D + c);
            *dp = *dp + sp;
        }
    }
    red[wave][lane] = st;
    reda[wave][lane] = sa;
    __syncthreads();
    if (wave == 1 && c < D && dbias) {
        const mb_f4 un = reda[0][lane] + reda[1][lane] + reda[2][lane] + reda[3][lane];
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (un[e] != 0.f) atomicAdd(dbias + c + e, un[e]);
    }
    if (wave == 0 && c < D) {
        const mb_f4 tot = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (token_scalar) {
            const float v = tot[0] + tot[1] + tot[2] + tot[3];
            if (v != 0.f) atomicAdd(dtoken, v);
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (tot[e] != 0.f) atomicAdd(dtoken + c + e, tot[e]);
        }
    }
}

// D == 1 (the RNA channel axis is the masked axis): thread = position t, loop over the batch (coalesced across t)
template <typename T, typename TDX>
__global__ __launch_bounds__(256) void mask_apply_bwd_d1_kernel(const T* dy, TDX* dx, const float* __restrict__ mask,
                                                                float* __restrict__ dtoken, float* __restrict__ dpos, int B, int Tn,
                                                                int first) {
    __shared__ float red[4];
    const int t = blockIdx.x * 256 + threadIdx.x;
    float st = 0.f;
    if (t < Tn) {
        float sp = 0.f;
        for (int b = 0; b < B; b++) {
            const long at = (long)b * Tn + t;
            const float g = ldf(dy + at);
            sp += g;
            const bool msk = t >= first && mask[(long)b * (Tn - first) + (t - first)] != 0.f;
            if (msk) st += g;
            stf(dx + at, msk ? 0.f : g);
        }
        dpos[t] += sp;
    }
    st = block_sum256(st, red);
    if (threadIdx.x == 0 && st != 0.f) atomicAdd(dtoken, st);
}

extern "C" int mh_mask_apply_fwd(const void* x, void* y, const float* mask, const float* token, const float* pos, int B, int T, int D,
                                 int first, int token_scalar, int dt_x, int dt_y, mh_stream s) {
    const long total = (long)B * T * D;
    if (total == 0) return MH_OK;
    if (dt_y == MH_F32 && D % 4 == 0 && !token_scalar && mh_quad_ok(x, mh_dt_size(dt_x)) &&
        (((uintptr_t)y | (uintptr_t)token | (uintptr_t)pos) & 15) == 0) {
        dim3 gv((unsigned)min((long)mh_cdiv((long)B * T, 4), 16384L));
        if (dt_x == MH_F32) hipLaunchKernelGGL((mask_apply_fwd_vec_kernel<float>), gv, dim3(256), 0, (hipStream_t)s, (const float*)x, (float*)y, mask, token, pos, B, T, D, first);
        else hipLaunchKernelGGL((mask_apply_fwd_vec_kernel<bf16_t>), gv, dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (float*)y, mask, token, pos, B, T, D, first);
        MH_LAUNCH_CHECK("mh_mask_apply_fwd");
        return MH_OK;
    }
#define MAF_(TX, TY) hipLaunchKernelGGL((mask_apply_fwd_kernel<TX, TY>), EW_GRID(total), dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, mask, token, pos, B, T, D, first, token_scalar)
    DISPATCH2(dt_x, dt_y, MAF_)
#undef MAF_
    MH_LAUNCH_CHECK("mh_mask_apply_fwd");
    return MH_OK;
}

// the quad form below (the one that can leave the bias gradient too)
extern "C" int mh_mask_apply_bwd_dbias_ok(const void* dy, const void* dx, const float* dpos, int D, int dt_dy, int dt_dx) {
    return dt_dy == MH_F32 && D % 4 == 0 && D >= 256 && mh_quad_ok(dx, mh_dt_size(dt_dx)) && (((uintptr_t)dy | (uintptr_t)dpos) & 15) == 0;
}

extern "C" int mh_mask_apply_bwd(const void* dy, void* dx, const float* mask, float* dtoken, float* dpos, int B, int T, int D,
                                 int first, int token_scalar, int dt_dy, int dt_dx, float* dbias, mh_stream s) {
    if (T == 0 || D == 0 || B == 0) return MH_OK;
    MH_REQUIRE(!dbias || mh_mask_apply_bwd_dbias_ok(dy, dx, dpos, D, dt_dy, dt_dx),
               "mh_mask_apply_bwd: dbias needs the quad form (f32 dy, D %% 4 == 0, D >= 256, 16-byte aligned operands)");
    if (D == 1) {
#define MAB1_(TDY, TDX) hipLaunchKernelGGL((mask_apply_bwd_d1_kernel<TDY, TDX>), dim3(mh_cdiv(T, 256)), dim3(256), 0, (hipStream_t)s, (const TDY*)dy, (TDX*)dx, mask, dtoken, dpos, B, T, first)
        DISPATCH2(dt_dy, dt_dx, MAB1_)
#undef MAB1_
        MH_LAUNCH_CHECK("mh_mask_apply_bwd");
        return MH_OK;
    }
    if (mh_mask_apply_bwd_dbias_ok(dy, dx, dpos, D, dt_dy, dt_dx)) {
        const int band = 16;     // rows of t per block: T / 16 x D / 256 blocks, 4 waves x 8 rows of 1 KiB in flight each
        dim3 gv(mh_cdiv(D, 256), mh_cdiv(T, band));
        if (dt_dx == MH_F32) hipLaunchKernelGGL((mask_apply_bwd_vec_kernel<float>), gv, dim3(256), 0, (hipStream_t)s, (const float*)dy, (float*)dx, mask, dtoken, dpos, B, T, D, first, token_scalar, band, dbias);
        else hipLaunchKernelGGL((mask_apply_bwd_vec_kernel<bf16_t>), gv, dim3(256), 0, (hipStream_t)s, (const float*)dy, (bf16_t*)dx, mask, dtoken, dpos, B, T, D, first, token_scalar, band, dbias);
        MH_LAUNCH_CHECK("mh_mask_apply_bwd");
        return MH_OK;
    }
    dim3 grid(mh_cdiv(D, 64), mh_cdiv(T, MB_BAND));
#define MAB_(TDY, TDX) hipLaunchKernelGGL((mask_apply_bwd_kernel<TDY, TDX>), grid, dim3(256), 0, (hipStream_t)s, (const TDY*)dy, (TDX*)dx, mask, dtoken, dpos, B, T, D, first, token_scalar)
    DISPATCH2(dt_dy, dt_dx, MAB_)
#undef MAB_
    MH_LAUNCH_CHECK("mh_mask_apply_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------ per-row scale (key-padding mask: zeroed rows, masked-mean landmarks)
// y[r, :] = x[r, :] * s[r]; one wave per row
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void row_scale_kernel(const T* x, const float* __restrict__ sc, T* y, long rows, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        const float f = sc[r];
        if (VEC) {
            for (int c = 4 * lane; c < D; c += 256) st4(y + r * D + c, ld4(x + r * D + c) * f);
        } else {
            for (int c = lane; c < D; c += 64) stf(y + r * D + c, ldf(x + r * D + c) * f);
        }
    }
}
extern "C" int mh_row_scale(const void* x, const float* scale, void* y, int64_t rows, int D, int dt, mh_stream s) {
    if (rows == 0 || D == 0) return MH_OK;
    dim3 grid((unsigned)min((long)mh_cdiv(rows, 4), 16384L));
    const bool vec = D % 4 == 0 && mh_quad_ok(x, mh_dt_size(dt)) && mh_quad_ok(y, mh_dt_size(dt));
#define RS_(T)                                                                                                                 \
    if (vec) hipLaunchKernelGGL((row_scale_kernel<T, true>), grid, dim3(256), 0, (hipStream_t)s, (const T*)x, scale, (T*)y, (long)rows, D); \
    else hipLaunchKernelGGL((row_scale_kernel<T, false>), grid, dim3(256), 0, (hipStream_t)s, (const T*)x, scale, (T*)y, (long)rows, D)
    if (dt == MH_F32) { RS_(float); } else { RS_(bf16_t); }
#undef RS_
    MH_LAUNCH_CHECK("mh_row_scale");
    return MH_OK;
}

// ------------------------------------------------------------------ key-padding plan of a Nystrom layer (BASELINE config 4)
// From the [B, n_src] bool mask of the patches, everything a layer's masked attention needs, in one launch (was ~10 ATen
// launches per layer): the f32 row mask of the front-padded sequence [pad zeros | lead ones | mask | mask[:, :wrap]], and per
// landmark group of l rows the valid flag (count > 0) and l * (1 / (count + 1e-8)).  One WAVE per group: its l rows are l consecutive
// floats of mrow (one coalesced store; one thread per group walked its 33 rows alone: 14 us on the critical path of config 4's first layer).
__global__ __launch_bounds__(256) void keymask_plan_kernel(const unsigned char* __restrict__ mask, float* __restrict__ mrow,
                                                           float* __restrict__ mlm, float* __restrict__ lscale, long B, long n_src,
                                                           int lead, int wrap, int pad, int l, long m) {
    const int lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= B * m) return;                                  // wave-uniform
    const long b = g / m, j0 = (g % m) * l;
    const unsigned char* mb = mask + b * n_src;
    float* out = mrow + b * (m * l) + j0;
    float cnt = 0.f;
    for (int i = lane; i < l; i += 64) {
        const long j = j0 + i - pad;                         // position in [lead ones | mask | wrapped head of the mask]
        float v = 0.f;
        if (j >= 0) {
            if (j < lead) v = 1.f;
            else if (j - lead < n_src) v = mb[j - lead] ? 1.f : 0.f;
            else v = mb[j - lead - n_src] ? 1.f : 0.f;
        }
        out[i] = v;
        cnt += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);      // counts: exact in any order
    if (lane == 0) {
        mlm[g] = cnt > 0.f ? 1.f : 0.f;
        lscale[g] = (float)l * (1.0f / (cnt + 1e-8f));
    }
}
extern "C" int mh_keymask_plan(const unsigned char* mask, float* mrow, float* mlm, float* lscale, int64_t B, int64_t n_src, int lead,
                               int wrap, int pad, int l, mh_stream s) {
    if (B == 0) return MH_OK;
    MH_REQUIRE(l > 0 && lead >= 0 && wrap >= 0 && pad >= 0 && wrap <= n_src, "mh_keymask_plan: l > 0, 0 <= wrap <= n_src, lead / pad >= 0");
    const long n_tot = (long)pad + lead + n_src + wrap;
    MH_REQUIRE(n_tot % l == 0, "mh_keymask_plan: pad + lead + n_src + wrap = %ld is no multiple of l = %d", n_tot, l);
    const long m = n_tot / l;
    hipLaunchKernelGGL(keymask_plan_kernel, dim3((unsigned)mh_cdiv(B * m, 4)), dim3(256), 0, (hipStream_t)s, mask, mrow, mlm, lscale, (long)B,
                       (long)n_src, lead, wrap, pad, l, m);
    MH_LAUNCH_CHECK("mh_keymask_plan");
    return MH_OK;
}

// ------------------------------------------------------------------ data feed: fixed-N resampling (datasets/dataset_pretrain.py:150-167)
// out[r, :] = src[row[r], :]: the gather behind `wsi_feature[sampled_indices]`, for a whole batch at once (row = global row
// index into the bank of concatenated slides).  One wave per row, 16-byte pieces; rows are F * esz bytes.
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const long* __restrict__ row, uint4* __restrict__ out,
                                                          long R, int chunks, long src_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
        long q = row[r];
        q = q < 0 ? 0 : (q >= src_rows ? src_rows - 1 : q);       // never read outside the bank
        for (int c = lane; c < chunks; c += 64) out[r * chunks + c] = src[q * chunks + c];
    }
}
__global__ __launch_bounds__(256) void gather_rows_bytes_kernel(const unsigned char* __restrict__ src, const long* __restrict__ row,
                                                                unsigned char* __restrict__ out, long R, long row_bytes, long src_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
        long q = row[r];
        q = q < 0 ? 0 : (q >= src_rows ? src_rows - 1 : q);
        for (long c = lane; c < row_bytes; c += 64) out[r * row_bytes + c] = src[q * row_bytes + c];
    }
}

extern "C" int mh_gather_rows(const void* src, const int64_t* row, void* out, int64_t R, int64_t F, int64_t src_rows, int dt, mh_stream s) {
    if (R == 0 || F == 0) return MH_OK;
    MH_REQUIRE(src_rows > 0, "mh_gather_rows: empty source bank");
    const long row_bytes = F * mh_dt_size(dt);
    dim3 grid((unsigned)min((long)mh_cdiv(R, 4), 16384L));
    if (row_bytes % 16 == 0 && (((uintptr_t)src | (uintptr_t)out) & 15) == 0)
        hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(256), 0, (hipStream_t)s, (const uint4*)src, (const long*)row, (uint4*)out, (long)R,
                           (int)(row_bytes / 16), (long)src_rows);
    else
        hipLaunchKernelGGL(gather_rows_bytes_kernel, grid, dim3(256), 0, (hipStream_t)s, (const unsigned char*)src, (const long*)row,
                           (unsigned char*)out, (long)R, row_bytes, (long)src_rows);
    MH_LAUNCH_CHECK("mh_gather_rows");
    return MH_OK;
}

// ------------------------------------------------------------------ encoder-output gradient fan-in
// one wave per (b, t) row; VEC: D % 4 == 0 and quad-aligned pointers
template <typename TX, bool VEC>
__global__ __launch_bounds__(256) void fanout_bwd_kernel(const float* __restrict__ gfull, const TX* __restrict__ x, float alpha,
                                                         const float* __restrict__ c, float* __restrict__ dE, int B, int Tn, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rows = (long)B * Tn;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const int t = (int)(row % Tn);
        const long b = row / Tn;
        const float* gr = gfull ? gfull + row * D : nullptr;
        const TX* xr = (x && t >= 1) ? x + (b * (Tn - 1) + (t - 1)) * D : nullptr;
        const float* cr = (c && t == 0) ? c + b * D : nullptr;
        if (VEC) {
            for (int k = 4 * lane; k < D; k += 256) {
                f4_t v = {0.f, 0.f, 0.f, 0.f};
                if (gr) v = ld4(gr + k);
                if (xr) v += ld4(xr + k) * alpha;
                if (cr) v += ld4(cr + k);
                st4(dE + row * D + k, v);
            }
        } else {
            for (int k = lane; k < D; k += 64) {
                float v = gr ? gr[k] : 0.f;
                if (xr) v += alpha * ldf(xr + k);
                if (cr) v += cr[k];
                dE[row * D + k] = v;
            }
        }
    }
}

extern "C" int mh_fanout_bwd(const float* gfull, const void* x, float alpha, const float* c, float* dE, int B, int T, int D, int dt_x,
                             mh_stream s) {
    if (B == 0 || T == 0 || D == 0) return MH_OK;
    const bool vec = D % 4 == 0 && mh_quad_ok(gfull, 4) && mh_quad_ok(x, mh_dt_size(dt_x)) && mh_quad_ok(c, 4) && mh_quad_ok(dE, 4);
    dim3 grid((unsigned)min((long)mh_cdiv((long)B * T, 4), 16384L));
#define FAN_(TX)                                                                                                                   \
    if (vec) hipLaunchKernelGGL((fanout_bwd_kernel<TX, true>), grid, dim3(256), 0, (hipStream_t)s, gfull, (const TX*)x, alpha, c, dE, B, T, D); \
    else hipLaunchKernelGGL((fanout_bwd_kernel<TX, false>), grid, dim3(256), 0, (hipStream_t)s, gfull, (const TX*)x, alpha, c, dE, B, T, D)
    if (dt_x == MH_F32) { FAN_(float); } else { FAN_(bf16_t); }
#undef FAN_
    MH_LAUNCH_CHECK("mh_fanout_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------ reparameterisation (models/mirror.py:830-833)
__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float* mu, const float* ls, const float* eps, float* z, long n) {
    EW_LOOP(i, n) z[i] = mu[i] + eps[i] * __expf(0.5f * ls[i]);
}
// add_mu / add_ls (nullable): gradients that reached mu / logstd by another road (the KL term reads both): summed in here, so
// autograd has nothing to add afterwards
__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* ls, const float* eps, const float* dz, const float* add_mu,
                                                          const float* add_ls, float* dmu, float* dls, long n) {
    EW_LOOP(i, n) {
        const float g = dz[i];
        dmu[i] = g + (add_mu ? add_mu[i] : 0.f);
        dls[i] = g * eps[i] * 0.5f * __expf(0.5f * ls[i]) + (add_ls ? add_ls[i] : 0.f);
    }
}
extern "C" int mh_reparam_fwd(const float* mu, const float* logstd, const float* eps, float* z, int64_t n, mh_stream s) {
    if (n == 0) return MH_OK;
    hipLaunchKernelGGL(reparam_fwd_kernel, EW_GRID(n), dim3(256), 0, (hipStream_t)s, mu, logstd, eps, z, (long)n);
    MH_LAUNCH_CHECK("mh_reparam_fwd");
    return MH_OK;
}
extern "C" int mh_reparam_bwd(const float* logstd, const float* eps, const float* dz, const float* add_mu, const float* add_logstd,
                              float* dmu, float* dlogstd, int64_t n, mh_stream s) {
    if (n == 0) return MH_OK;
    hipLaunchKernelGGL(reparam_bwd_kernel, EW_GRID(n), dim3(256), 0, (hipStream_t)s, logstd, eps, dz, add_mu, add_logstd, dmu, dlogstd,
                       (long)n);
    MH_LAUNCH_CHECK("mh_reparam_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------ exp of a small f32 tensor (logit_scale.exp(), models/mirror.py:911)
__global__ __launch_bounds__(256) void exp_fwd_kernel(const float* x, float* y, long n) {
    EW_LOOP(i, n) y[i] = expf(x[i]);
}
// dx (+)= dy * y: with acc the parameter's gradient is summed straight into its arena slot (no mul + add pair on the autograd side)
__global__ __launch_bounds__(256) void exp_bwd_kernel(const float* dy, const float* y, float* dx, long n, int acc) {
    EW_LOOP(i, n) {
        const float g = dy[i] * y[i];
        dx[i] = acc ? dx[i] + g : g;
    }
}
extern "C" int mh_exp_fwd(const float* x, float* y, int64_t n, mh_stream s) {
    if (n == 0) return MH_OK;
    hipLaunchKernelGGL(exp_fwd_kernel, EW_GRID(n), dim3(256), 0, (hipStream_t)s, x, y, (long)n);
    MH_LAUNCH_CHECK("mh_exp_fwd");
    return MH_OK;
}
extern "C" int mh_exp_bwd(const float* dy, const float* y, float* dx, int64_t n, int accumulate, mh_stream s) {
    if (n == 0) return MH_OK;
    hipLaunchKernelGGL(exp_bwd_kernel, EW_GRID(n), dim3(256), 0, (hipStream_t)s, dy, y, dx, (long)n, accumulate);
    MH_LAUNCH_CHECK("mh_exp_bwd");
    return MH_OK;
}

// ------------------------------------------------------------------ RNA encoder attention (models/mirror.py:77-102)
// The reference feeds a 2-D [B, D] tensor: q,k,v are [B, H, hd] and SDPA attends over the HEADS axis (an H x H
// matrix per sample); the result is permuted by transpose(1,2).reshape(B, D): out[b, d*H + h] = o[h][d].
#define HA_MAXH 16
template <typename T>
__global__ __launch_bounds__(256) void headattn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ attn,
                                                           int H, int hd) {
    extern __shared__ float sm[];  // q,k,v [3][H*hd] + a [H*H]
    const int D = H * hd;
    const long b = blockIdx.x;
    float* q = sm; float* k = sm + D; float* v = sm + 2 * D; float* a = sm + 3 * D;
    for (int i = threadIdx.x; i < 3 * D; i += 256) sm[i] = ldf(qkv + b * 3 * D + i);
    __syncthreads();
    const float scale = rsqrtf((float)hd);
    for (int p = threadIdx.x; p < H * H; p += 256) {
        const int h = p / H, g = p % H;
        float s = 0.f;
        for (int d = 0; d < hd; d++) s += q[h * hd + d] * k[g * hd + d];
        a[p] = s * scale;
    }
    __syncthreads();
    if (threadIdx.x < H) {
        float* ar = a + threadIdx.x * H;
        float m = -INFINITY;
        for (int g = 0; g < H; g++) m = fmaxf(m, ar[g]);
        float sum = 0.f;
        for (int g = 0; g < H; g++) { ar[g] = __expf(ar[g] - m); sum += ar[g]; }
        for (int g = 0; g < H; g++) { ar[g] /= sum; attn[b * H * H + threadIdx.x * H + g] = ar[g]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256) {
        const int d = i / H, h = i % H;  // output column i = d*H + h
        float s = 0.f;
        for (int g = 0; g < H; g++) s += a[h * H + g] * v[g * hd + d];
        stf(out + b * D + i, s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void headattn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ attn,
                                                           const T* __restrict__ dout, T* __restrict__ dqkv, int H, int hd) {
    extern __shared__ float sm[];  // q,k,v [3D], do [D] (as [h][d]), a [H*H], ds [H*H]
    const int D = H * hd;
    const long b = blockIdx.x;
    float* q = sm; float* k = sm + D; float* v = sm + 2 * D; float* dO = sm + 3 * D;
    float* a = sm + 4 * D; float* ds = a + H * H;
    for (int i = threadIdx.x; i < 3 * D; i += 256) sm[i] = ldf(qkv + b * 3 * D + i);
    for (int i = threadIdx.x; i < D; i += 256) { const int d = i / H, h = i % H; dO[h * hd + d] = ldf(dout + b * D + i); }
    for (int i = threadIdx.x; i < H * H; i += 256) a[i] = attn[b * H * H + i];
    __syncthreads();
    // da[h][g] = sum_d dO[h][d] v[g][d]
    for (int p = threadIdx.x; p < H * H; p += 256) {
        const int h = p / H, g = p % H;
        float s = 0.f;
        for (int d = 0; d < hd; d++) s += dO[h * hd + d] * v[g * hd + d];
        ds[p] = s;
    }
    __syncthreads();
    if (threadIdx.x < H) {
        const int h = threadIdx.x;
        float dot = 0.f;
        for (int g = 0; g < H; g++) dot += ds[h * H + g] * a[h * H + g];
        for (int g = 0; g < H; g++) ds[h * H + g] = a[h * H + g] * (ds[h * H + g] - dot);
    }
    __syncthreads();
    const float scale = rsqrtf((float)hd);
    for (int i = threadIdx.x; i < D; i += 256) {
        const int h = i / hd, d = i % hd;
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int g = 0; g < H; g++) {
            dq += ds[h * H + g] * k[g * hd + d];
            dk += ds[g * H + h] * q[g * hd + d];
            dv += a[g * H + h] * dO[g * hd + d];
        }
        stf(dqkv + b * 3 * D + i, dq * scale);
        stf(dqkv + b * 3 * D + D + i, dk * scale);
        stf(dqkv + b * 3 * D + 2 * D + i, dv);
    }
}

extern "C" int mh_headattn_fwd(const void* qkv, void* out, float* attn, int B, int H, int hd, int dt, mh_stream s) {
    MH_REQUIRE(H >= 1 && H <= 64 && (long)H * hd <= 4096, "mh_headattn_fwd: H=%d hd=%d unsupported", H, hd);
    if (B == 0) return MH_OK;
    const size_t lds = (3 * (size_t)H * hd + (size_t)H * H) * sizeof(float);
    MH_DISPATCH_DT(dt, T, hipLaunchKernelGGL((headattn_fwd_kernel<T>), dim3(B), dim3(256), lds, (hipStream_t)s, (const T*)qkv, (T*)out, attn, H, hd));
    MH_LAUNCH_CHECK("mh_headattn_fwd");
    return MH_OK;
}

extern "C" int mh_headattn_bwd(const void* qkv, const float* attn, const void* dout, void* dqkv, int B, int H, int hd, int dt,
                               mh_stream s) {
    MH_REQUIRE(H >= 1 && H <= 64 && (long)H * hd <= 4096, "mh_headattn_bwd: H=%d hd=%d unsupported", H, hd);
    if (B == 0) return MH_OK;
    const size_t lds = (4 * (size_t)H * hd + 2 * (size_t)H * H) * sizeof(float);
    MH_DISPATCH_DT(dt, T, hipLaunchKernelGGL((headattn_bwd_kernel<T>), dim3(B), dim3(256), lds, (hipStream_t)s, (const T*)qkv, attn, (const T*)dout, (T*)dqkv, H, hd));
    MH_LAUNCH_CHECK("mh_headattn_bwd");
    return MH_OK;
}

__global__ void timestamp_kernel(unsigned long long* dst) { *dst = wall_clock64(); }

extern "C" int mh_timestamp(uint64_t* dst, mh_stream s) {
    MH_REQUIRE(dst != nullptr, "mh_timestamp: null destination");
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, (unsigned long long*)dst);
    MH_LAUNCH_CHECK("mh_timestamp");
    return MH_OK;
}
