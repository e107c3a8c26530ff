// Shared device/host helpers for libmirror_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mirror_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

void mh_set_error(const char* fmt, ...);

#define MH_REQUIRE(cond, ...)                  \
    do {                                       \
        if (!(cond)) {                         \
            mh_set_error(__VA_ARGS__);         \
            return MH_EINVAL;                  \
        }                                      \
    } while (0)

#define MH_LAUNCH_CHECK(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            mh_set_error("%s: HIP launch error: %s", name, hipGetErrorString(e_));   \
            return MH_EHIP;                                                          \
        }                                                                            \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(bf16_t, b);
}
// two f32 -> one dword of bf16 (lo in bits 0-15): ONE v_cvt_pk_bf16_f32 (f2bf(lo) | f2bf(hi) << 16 costs two of them, a shift and an or)
typedef __bf16 mh_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mh_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const mh_f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, mh_bf16x2));
}

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// four consecutive elements as f32 (16-byte f32 / 8-byte bf16 accesses; the pointer must be aligned to that)
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ f4_t ld4(const T* p);
template <> __device__ __forceinline__ f4_t ld4<float>(const float* p) { return *reinterpret_cast<const f4_t*>(p); }
template <> __device__ __forceinline__ f4_t ld4<bf16_t>(const bf16_t* p) {
    const u2_t u = *reinterpret_cast<const u2_t*>(p);
    f4_t r = {__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u), __uint_as_float(u[1] << 16), __uint_as_float(u[1] & 0xffff0000u)};
    return r;
}
template <typename T> __device__ __forceinline__ void st4(T* p, f4_t v);
template <> __device__ __forceinline__ void st4<float>(float* p, f4_t v) { *reinterpret_cast<f4_t*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f4_t v) {
    u2_t u = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    *reinterpret_cast<u2_t*>(p) = u;
}
// host side: can `p` be accessed as quads of `esz`-byte elements?
static inline bool mh_quad_ok(const void* p, int esz) { return ((uintptr_t)p % (4 * (uintptr_t)esz)) == 0; }
static inline int mh_dt_size(int dt) { return dt == MH_F32 ? 4 : 2; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide reductions for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS.
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max256(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Philox4x32-10 (dropout masks): element i of a tensor uses word (i & 3) of the block with counter (offset + i) >> 2
__device__ __forceinline__ void philox4x32_10(uint32_t (&ctr)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ k1;
        ctr[0] = n0; ctr[1] = (uint32_t)p1; ctr[2] = n2; ctr[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// "Lite" dropout stream for the [B, n, D]-sized dropouts of the WSI layers ([3P] to_out[1], models/mirror.py:312): Philox4x32 with
// 7 rounds (the shortest variant that passes BigCrush in Salmon et al.) and 16 random bits per element, i.e. one block serves 8
// elements: element i uses 16-bit field (i & 7) of the block with counter (offset + i) >> 3 (field f = low / high half of word
// f >> 1 for even / odd f); it is kept when field >= thr16 = round(p * 65536), scaled by 65536 / (65536 - thr16) (the exact keep
// probability, so the estimator stays unbiased).  3x cheaper per element than the 10-round / 32-bit stream: with the mask drawn
// in a GEMM epilogue there is no HBM time to hide the integer multiplies under.
__device__ __forceinline__ void philox4x32_7(uint32_t (&ctr)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 7; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ k1;
        ctr[0] = n0; ctr[1] = (uint32_t)p1; ctr[2] = n2; ctr[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ uint32_t drop16_thr(float p) { return (uint32_t)fminf(p * 65536.f + 0.5f, 65535.f); }
__device__ __forceinline__ float drop16_scale(uint32_t thr16) { return 65536.f / (float)(65536u - thr16); }
// keep flags of the 8 elements of block `blk` as a bit mask (bit e = element e is kept)
__device__ __forceinline__ uint32_t drop16_keep8(uint64_t blk, uint64_t seed, uint32_t thr16) {
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u};
    philox4x32_7(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t m = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        m |= ((ctr[w] & 0xffffu) >= thr16 ? 1u : 0u) << (2 * w);
        m |= ((ctr[w] >> 16) >= thr16 ? 1u : 0u) << (2 * w + 1);
    }
    return m;
}

static inline int mh_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Dispatch a functor-style macro over one runtime dtype.
#define MH_DISPATCH_DT(dt, T, ...)                              \
    if ((dt) == MH_F32) { using T = float; __VA_ARGS__; }       \
    else { using T = bf16_t; __VA_ARGS__; }
