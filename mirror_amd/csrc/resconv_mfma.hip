// [3P] NystromAttention.res_conv (33-tap depthwise conv along the sequence, one filter per head; called at
// models/mirror.py:312) on the matrix cores — bf16 policy, dh = 64, taps = 33.
//
// As VALU code the conv is 33 MACs per element = 1.2 G FMA per call at c2: ~36 us of pure issue, 90 us measured.
// As a banded Toeplitz product it is 8 MFMAs per 32 x 64 output block and purely memory bound:
//     out[t0 + i, c] = sum_k A[i][k] v[t0 - 16 + k, c],     A[i][k] = w[k - i] for 0 <= k - i <= 32 (else 0),  k < 64
// The accumulators hold out^T (c in registers, t in lanes); they pass through an f32 LDS image so that the read-modify-
// write of `out` moves whole 128-byte rows in 16-byte pieces (per-lane 8-byte pieces of 32 different rows ran at a
// quarter of the HBM rate); the v tile (128 + 32 halo rows of one head) sits in LDS and is read with
// ds_read_b64_tr_b16 in the accumulator's k order (see nystrom_fused.hip).
// The weight gradient dw[j] = sum_{b,t,c} dout[t, c] v[t + j - 16, c] is the diagonal sums of
//     S[i][k] = sum_c dout[t0 + i, c] v[t0 - 16 + k, c]
// accumulated over every row block: both operands are channel-contiguous rows, read straight from HBM as fragments.
#include <cstdlib>
#include "gemm_kernel.h"

namespace {

constexpr int RM_P = 72;       // LDS pitch (bf16) of the [rows][64] image
constexpr int RM_T = 128;      // output rows per workgroup
constexpr int RM_HALO = 16;    // taps / 2
constexpr int RM_TAPS = 33;
constexpr int RM_DH = 64;
constexpr int RM_SP = 68;      // LDS pitch (f32) of the [128][64] output image

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// fragment with free index = image column (col0 + lane&31), contraction index = image rows kb + 4hl + {0..3}, kb + 8 + 4hl + {0..3}
__device__ __forceinline__ bf16x8 rm_frag_tr(const bf16_t* img, int col0, int kb, int lane) {
    const int g16 = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const bf16_t* a0 = img + (kb + 4 * (g16 >> 1) + q) * RM_P + col0 + 16 * (g16 & 1) + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 8 * RM_P));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// grid (n_p / 128, heads, B)
__global__ __launch_bounds__(256) void resconv_mfma_kernel(const bf16_t* __restrict__ v, long ldv, long v_bs, const float* __restrict__ w,
                                                           bf16_t* out, long ldo, long o_bs, int n_p, int transpose, int accumulate) {
    static_assert(64 * RM_SP * 4 <= (RM_T + 2 * RM_HALO) * RM_P * 2, "half of the f32 output image must fit the v tile");
    __shared__ __attribute__((aligned(16))) bf16_t img[(RM_T + 2 * RM_HALO) * RM_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 5, r = lane & 31;
    const int t0 = blockIdx.x * RM_T, h = blockIdx.y, b = blockIdx.z;
    const bf16_t* vb = v + (long)b * v_bs + h * RM_DH;
    // image row q <-> sequence position t0 - 16 + q, zero outside [0, n_p)
#pragma unroll
    for (int i = 0; i < (RM_T + 2 * RM_HALO) * 8 / 256; i++) {
        const int cid = tid + i * 256, q = cid >> 3, c = cid & 7;
        const int t = t0 - RM_HALO + q;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (t >= 0 && t < n_p) val = *reinterpret_cast<const u32x4*>(vb + (long)t * ldv + c * 8);
        *reinterpret_cast<u32x4*>(img + q * RM_P + c * 8) = val;
    }
    // accumulate: the addend rows are requested here, beside the v tile, not behind the MFMAs (one more dependent round trip per
    // workgroup otherwise: the transposed pass of the backward ran 125 us against the forward's 75 us beside the chain)
    bf16_t* ob = out + (long)b * o_bs + h * RM_DH;
    u32x4 old[2][64 * 8 / 256];
    if (accumulate == 2) {
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
            for (int i = 0; i < 64 * 8 / 256; i++) {
                const int cid = tid + i * 256, q = cid >> 3, c = cid & 7;
                const int t = t0 + 64 * half + q;
                old[half][i] = (u32x4){0u, 0u, 0u, 0u};
                if (t < n_p) old[half][i] = *reinterpret_cast<const u32x4*>(ob + (long)t * ldo + 8 * c);
            }
    }
    // Toeplitz operand W^T[k][t = r] = w[k - r], k in the accumulator order of k-step ks
    const float* wh = w + h * RM_TAPS;
    bf16x8 wf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int k = 16 * ks + (e < 4 ? 4 * hl + e : 8 + 4 * hl + (e - 4));
            const int j = k - r;
            const float x = (j >= 0 && j < RM_TAPS) ? wh[transpose ? RM_TAPS - 1 - j : j] : 0.f;
            wf[ks][e] = (__bf16)x;
        }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
#pragma unroll
        for (int e = 0; e < 16; e++) acc[nb][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) acc[nb] = MFMA(rm_frag_tr(img, 32 * nb, 32 * wave + 16 * ks, lane), wf[ks], acc[nb]);
    }
    // out^T accumulators -> f32 [t][c] image -> whole 128-byte rows read-modify-written with 16-byte accesses.  The image
    // reuses the v tile's LDS (64 rows at a time: 17 KB inside the 23 KB tile) so that 6 workgroups fit a CU: the kernel is a
    // chain of dependent memory round trips per workgroup and lives on occupancy.
    float* stage = reinterpret_cast<float*>(img);
    __syncthreads();                                  // every wave is done reading the v tile
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if ((wave >> 1) == half) {
            float* srow = stage + (32 * (wave & 1) + r) * RM_SP + 4 * hl;
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f4_t x = {acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]};
                    *reinterpret_cast<f4_t*>(srow + 32 * nb + 8 * g) = x;
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 64 * 8 / 256; i++) {
            const int cid = tid + i * 256, q = cid >> 3, c = cid & 7;
            const int t = t0 + 64 * half + q;
            if (t >= n_p) continue;
            const float* sp = stage + q * RM_SP + 8 * c;
            f4_t lo = *reinterpret_cast<const f4_t*>(sp), hi = *reinterpret_cast<const f4_t*>(sp + 4);
            bf16_t* p = ob + (long)t * ldo + 8 * c;
            if (accumulate) {
                const u32x4 od = accumulate == 2 ? old[half][i] : *reinterpret_cast<const u32x4*>(p);
                lo[0] += __uint_as_float(od[0] << 16); lo[1] += __uint_as_float(od[0] & 0xffff0000u);
                lo[2] += __uint_as_float(od[1] << 16); lo[3] += __uint_as_float(od[1] & 0xffff0000u);
                hi[0] += __uint_as_float(od[2] << 16); hi[1] += __uint_as_float(od[2] & 0xffff0000u);
                hi[2] += __uint_as_float(od[3] << 16); hi[3] += __uint_as_float(od[3] & 0xffff0000u);
            }
            u32x4 o;
            o[0] = pack_bf2(lo[0], lo[1]);
            o[1] = pack_bf2(lo[2], lo[3]);
            o[2] = pack_bf2(hi[0], hi[1]);
            o[3] = pack_bf2(hi[2], hi[3]);
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(p));      // written once, whole 128-byte rows: must not evict the v rows the next row tile's halo reads
        }
        if (half == 0) __syncthreads();               // the first half is consumed before the second overwrites it
    }
}

// grid (splits, heads, B); wave w of the block walks 32-row blocks blockIdx.x * 4 + w, + 4 gridDim.x, ...
__global__ __launch_bounds__(256) void resconv_wgrad_mfma_kernel(const bf16_t* __restrict__ v, long ldv, long v_bs,
                                                                 const bf16_t* __restrict__ dout, long ldo, long o_bs,
                                                                 float* __restrict__ dw, int n_p) {
    __shared__ float gsum[64];
    __shared__ float s_S[4 * 32 * 65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 5, r = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    if (tid < 64) gsum[tid] = 0.f;
    const bf16_t* vb = v + (long)b * v_bs + h * RM_DH + 8 * hl;
    const bf16_t* gb = dout + (long)b * o_bs + h * RM_DH + 8 * hl;
    f32x16 acc[2];     // S[i = output row (registers)][k = input row (lanes)], k block 0 / 1
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[kb][e] = 0.f;
    const int nblk = (n_p + 31) / 32;
    // the fragments of block tb + stride are requested before the MFMAs of block tb: a wave walks ~17 blocks, and with one block's
    // loads in flight at a time every step stood at the full memory latency (38 us alone, 143 us beside the chain's traffic)
    auto fetch = [&](int tb, bf16x8 (&af)[4], bf16x8 (&bf)[2][4]) {
        const int t = 32 * tb + r;
        const bool tok = t < n_p;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            u32x4 a = {0u, 0u, 0u, 0u};
            if (tok) a = *reinterpret_cast<const u32x4*>(gb + (long)t * ldo + 16 * ks);
            af[ks] = __builtin_bit_cast(bf16x8, a);
        }
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            const int q = 32 * tb - RM_HALO + 32 * kb + r;
            const bool qok = q >= 0 && q < n_p;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                u32x4 x = {0u, 0u, 0u, 0u};
                if (qok) x = *reinterpret_cast<const u32x4*>(vb + (long)q * ldv + 16 * ks);
                bf[kb][ks] = __builtin_bit_cast(bf16x8, x);
            }
        }
    };
    auto mac = [&](const bf16x8 (&af)[4], const bf16x8 (&bf)[2][4]) {
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) acc[kb] = MFMA(af[ks], bf[kb][ks], acc[kb]);
    };
    const int stride = 4 * gridDim.x;
    int tb = blockIdx.x * 4 + wave;
    bf16x8 a0[4], b0[2][4], a1[4], b1[2][4];
    if (tb < nblk) fetch(tb, a0, b0);
    while (tb < nblk) {
        if (tb + stride < nblk) fetch(tb + stride, a1, b1);
        mac(a0, b0);
        tb += stride;
        if (tb >= nblk) break;
        if (tb + stride < nblk) fetch(tb + stride, a0, b0);
        mac(a1, b1);
        tb += stride;
    }
    __syncthreads();
    // diagonal sums: S[i][k] belongs to tap j = k - i.  The 32 x 64 result of every wave goes through LDS once and 33 lanes add up one
    // diagonal each (2048 LDS atomics per wave onto 33 addresses serialised: most of the kernel's 38 us)
    float* S = s_S + wave * (32 * 65);
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int i = 8 * (e >> 2) + 4 * hl + (e & 3);
            S[i * 65 + 32 * kb + r] = acc[kb][e];
        }
    __syncthreads();
    if (lane < RM_TAPS) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; i++) t += S[i * 65 + lane + i];
        atomicAdd(&gsum[lane], t);
    }
    __syncthreads();
    if (tid < RM_TAPS) atomicAdd(dw + h * RM_TAPS + tid, gsum[tid]);
}

// ============================================================================ backward in ONE pass (round 5)
// dv[t, c] += sum_j w[j] dout[t - j + 16, c]   (the adjoint conv, read-modify-write of the v columns of d qkv)   and
// dw[j]    += sum_{t, c} dout[t, c] v[t + j - 16, c]
// were two launches that both streamed dout: 213 + 142 MB and, beside the half-chip pinv chain, 95 + 119 us of the backward window's main
// side.  The adjoint-conv kernel above already has the dout tile (+ halo) of one head in LDS; the tap gradient needs, per 32-row block,
//     S[i][k] = sum_c dout[t0 + i, c] v[t0 - 16 + k, c]      (i < 32, k < 64; tap j = k - i)
// = 8 more MFMAs whose A fragments are rows of that image and whose B fragments (64 rows of v) come straight from HBM beside the tile.
// The diagonal sums stay in registers: lane k of row i holds tap (k - i) mod 32 (column block 0 for k >= i, block 1 for k < i), so one
// ds_bpermute per accumulator register rotates row i by i lanes and the 32 rows add up lane-wise (no LDS image, no LDS atomics:
// the kernel lives on occupancy); tap 32 is the diagonal of column block 1.  A workgroup leaves its 33 sums in `part`
// [B][heads][tiles][33] and resconv_bwd_fold_kernel adds them to dw (4352 workgroups x 33 same-address atomics would serialise).
// grid (n_p / 128, heads, B)
__global__ __launch_bounds__(256, 3) void resconv_bwd_kernel(const bf16_t* __restrict__ dout, long ldo, long o_bs, const bf16_t* __restrict__ v,
                                                          long ldv, long v_bs, const float* __restrict__ w, bf16_t* dv, long lddv, long dv_bs,
                                                          float* __restrict__ part, int n_p) {
    static_assert(64 * RM_SP * 4 <= (RM_T + 2 * RM_HALO) * RM_P * 2, "half of the f32 output image must fit the dout tile");
    __shared__ __attribute__((aligned(16))) bf16_t img[(RM_T + 2 * RM_HALO) * RM_P];
    __shared__ float s_part[4][RM_TAPS + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 5, r = lane & 31;
    const int t0 = blockIdx.x * RM_T, h = blockIdx.y, b = blockIdx.z;
    const bf16_t* gb = dout + (long)b * o_bs + h * RM_DH;
    // image row q <-> sequence position t0 - 16 + q of dout, zero outside [0, n_p)
#pragma unroll
    for (int i = 0; i < (RM_T + 2 * RM_HALO) * 8 / 256; i++) {
        const int cid = tid + i * 256, q = cid >> 3, c = cid & 7;
        const int t = t0 - RM_HALO + q;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (t >= 0 && t < n_p) val = *reinterpret_cast<const u32x4*>(gb + (long)t * ldo + c * 8);
        *reinterpret_cast<u32x4*>(img + q * RM_P + c * 8) = val;
    }
    // the addend rows (the attention kernels' dv) and the v fragments of the tap gradient are requested beside the tile
    bf16_t* ob = dv + (long)b * dv_bs + h * RM_DH;
    u32x4 old[2][64 * 8 / 256];
#pragma unroll
    for (int half = 0; half < 2; half++)
#pragma unroll
        for (int i = 0; i < 64 * 8 / 256; i++) {
            const int cid = tid + i * 256, q = cid >> 3, c = cid & 7;
            const int t = t0 + 64 * half + q;
            old[half][i] = (u32x4){0u, 0u, 0u, 0u};
            if (t < n_p) old[half][i] = *reinterpret_cast<const u32x4*>(ob + (long)t * lddv + 8 * c);
        }
    const bf16_t* vb = v + (long)b * v_bs + h * RM_DH + 8 * hl;
    bf16x8 bfv[2][4];       // B fragments: v rows t0 + 32 wave - 16 + 32 kb + r
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
        const int q = t0 + 32 * wave - RM_HALO + 32 * kb + r;
        const bool qok = q >= 0 && q < n_p;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            u32x4 x = {0u, 0u, 0u, 0u};
            if (qok) x = *reinterpret_cast<const u32x4*>(vb + (long)q * ldv + 16 * ks);
            bfv[kb][ks] = __builtin_bit_cast(bf16x8, x);
        }
    }
    // Toeplitz operand of the ADJOINT conv: W^T[k][t = r] = w[32 - (k - r)], k in the accumulator order of k-step ks
    const float* wh = w + h * RM_TAPS;
    bf16x8 wf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int k = 16 * ks + (e < 4 ? 4 * hl + e : 8 + 4 * hl + (e - 4));
            const int j = k - r;
            const float x = wh[RM_TAPS - 1 - min(max(j, 0), RM_TAPS - 1)];      // unconditional load + select: no branch per element
            wf[ks][e] = (__bf16)((j >= 0 && j < RM_TAPS) ? x : 0.f);
        }
    __syncthreads();
    f32x16 acc[2], S[2];
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
#pragma unroll
        for (int e = 0; e < 16; e++) acc[nb][e] = S[nb][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) acc[nb] = MFMA(rm_frag_tr(img, 32 * nb, 32 * wave + 16 * ks, lane), wf[ks], acc[nb]);
    }
    // S[i (registers)][k (lanes)]: A = this wave's 32 rows of dout (image rows 16 + 32 wave + r, channel-contiguous), B = the v rows
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(img + (RM_HALO + 32 * wave + r) * RM_P + 16 * ks + 8 * hl);
#pragma unroll
        for (int kb = 0; kb < 2; kb++) S[kb] = MFMA(af, bfv[kb][ks], S[kb]);
    }
    // diagonal sums in registers (see the header of this kernel)
    float tsum = 0.f, t32 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int i = 8 * (e >> 2) + 4 * hl + (e & 3);
        const float val = (r >= i) ? S[0][e] : S[1][e];                   // tap (r - i) mod 32
        t32 += (r == i) ? S[1][e] : 0.f;                                  // tap 32: k' = 32 + i
        const int src = 32 * hl + ((r + i) & 31);                         // the lane that holds THIS lane's tap in row i
        tsum += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, val)));
    }
    tsum += __shfl_xor(tsum, 32, 64);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t32 += __shfl_xor(t32, o, 64);
    if (lane < 32) s_part[wave][lane] = tsum;
    if (lane == 0) s_part[wave][32] = t32;
    // out^T accumulators -> f32 [t][c] image -> whole 128-byte rows read-modify-written with 16-byte accesses (as resconv_mfma_kernel)
    float* stage = reinterpret_cast<float*>(img);
    __syncthreads();                                  // every wave is done reading the dout tile; s_part is complete
    if (tid < RM_TAPS)
        part[(((long)b * gridDim.y + h) * gridDim.x + blockIdx.x) * RM_TAPS + tid] = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid];
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if ((wave >> 1) == half) {
            float* srow = stage + (32 * (wave & 1) + r) * RM_SP + 4 * hl;
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f4_t x = {acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]};
                    *reinterpret_cast<f4_t*>(srow + 32 * nb + 8 * g) = x;
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 64 * 8 / 256; i++) {
            const int cid = tid + i * 256, q = cid >> 3, c = cid & 7;
            const int t = t0 + 64 * half + q;
            if (t >= n_p) continue;
            const float* sp = stage + q * RM_SP + 8 * c;
            f4_t lo = *reinterpret_cast<const f4_t*>(sp), hi = *reinterpret_cast<const f4_t*>(sp + 4);
            const u32x4 od = old[half][i];
            lo[0] += __uint_as_float(od[0] << 16); lo[1] += __uint_as_float(od[0] & 0xffff0000u);
            lo[2] += __uint_as_float(od[1] << 16); lo[3] += __uint_as_float(od[1] & 0xffff0000u);
            hi[0] += __uint_as_float(od[2] << 16); hi[1] += __uint_as_float(od[2] & 0xffff0000u);
            hi[2] += __uint_as_float(od[3] << 16); hi[3] += __uint_as_float(od[3] & 0xffff0000u);
            u32x4 o;
            o[0] = pack_bf2(lo[0], lo[1]);
            o[1] = pack_bf2(lo[2], lo[3]);
            o[2] = pack_bf2(hi[0], hi[1]);
            o[3] = pack_bf2(hi[2], hi[3]);
            *reinterpret_cast<u32x4*>(ob + (long)t * lddv + 8 * c) = o;      // read next by to_qkv's data gradient: ordinary stores
        }
        if (half == 0) __syncthreads();               // the first half is consumed before the second overwrites it
    }
}

// dw[h][j] += sum over the (b, tile) partial sums of resconv_bwd_kernel.  grid (heads, B): a workgroup adds up the tiles of one (b, h)
// — 4 quarter-sums per tap, <= 16 independent loads per thread — and leaves with one atomic per tap (B per address).  (As 8 workgroups whose
// threads walked all B x tiles partials one dependent load at a time this fold took longer than the pass it follows.)
__global__ __launch_bounds__(256) void resconv_bwd_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int tiles) {
    __shared__ float red[4][64];
    const int h = blockIdx.x, b = blockIdx.y, j = threadIdx.x & 63, q = threadIdx.x >> 6;
    const float* pb = part + ((long)b * gridDim.x + h) * tiles * RM_TAPS;
    float s = 0.f;
    if (j < RM_TAPS) {
        int t = q;
        for (; t + 12 < tiles; t += 16) {
            const float a0 = pb[(long)t * RM_TAPS + j], a1 = pb[(long)(t + 4) * RM_TAPS + j], a2 = pb[(long)(t + 8) * RM_TAPS + j],
                        a3 = pb[(long)(t + 12) * RM_TAPS + j];
            s += (a0 + a1) + (a2 + a3);
        }
        for (; t < tiles; t += 4) s += pb[(long)t * RM_TAPS + j];
    }
    red[q][j] = s;
    __syncthreads();
    if (q == 0 && j < RM_TAPS) atomicAdd(dw + h * RM_TAPS + j, red[0][j] + red[1][j] + red[2][j] + red[3][j]);
}

}  // namespace

// floats of `part` for resconv_bwd_try_mfma (0: the shape is not taken)
long resconv_bwd_part_floats(int B, int n_p, int heads, int dh, int taps) {
    if (dh != RM_DH || taps != RM_TAPS || B <= 0 || n_p <= 0) return 0;
    return (long)B * heads * mh_cdiv(n_p, RM_T) * RM_TAPS;
}

// adjoint conv (accumulated into dv) + tap gradient (accumulated into dw) in one pass over dout; true when the MFMA path took the launch
bool resconv_bwd_try_mfma(const void* dout, long ldo, long o_bs, const void* v, long ldv, long v_bs, const float* w, void* dv, long lddv,
                          long dv_bs, float* dw, float* part, long part_floats, int B, int n_p, int heads, int dh, int taps, int dt,
                          hipStream_t s) {
    if (dt != MH_BF16 || dh != RM_DH || taps != RM_TAPS) return false;
    if (ldv % 8 || v_bs % 8 || ldo % 8 || o_bs % 8 || lddv % 8 || dv_bs % 8 || ((uintptr_t)v & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dv & 15))
        return false;
    if (!part || part_floats < resconv_bwd_part_floats(B, n_p, heads, dh, taps)) return false;
    const int tiles = mh_cdiv(n_p, RM_T);
    hipLaunchKernelGGL(resconv_bwd_kernel, dim3(tiles, heads, B), dim3(256), 0, s, (const bf16_t*)dout, ldo, o_bs, (const bf16_t*)v, ldv, v_bs,
                       w, (bf16_t*)dv, lddv, dv_bs, part, n_p);
    hipLaunchKernelGGL(resconv_bwd_fold_kernel, dim3(heads, B), dim3(256), 0, s, (const float*)part, dw, tiles);
    return true;
}

// true when the MFMA path took the launch (bf16 in / bf16 out, dh = 64, 33 taps, 16-byte aligned rows)
bool resconv_try_mfma(const void* v, long ldv, long v_bs, const float* w, void* out, long ldo, long o_bs, int B, int n_p, int heads,
                      int dh, int taps, int transpose, int accumulate, int dt_v, int dt_o, hipStream_t s) {
    if (dt_v != MH_BF16 || dt_o != MH_BF16 || dh != RM_DH || taps != RM_TAPS) return false;
    if (ldv % 8 || v_bs % 8 || ldo % 8 || o_bs % 8 || ((uintptr_t)v & 15) || ((uintptr_t)out & 15)) return false;
    dim3 grid(mh_cdiv(n_p, RM_T), heads, B);
    constexpr bool early = true;      // the addend rows are requested beside the v tile (-0.17 % step time, round 3)
    hipLaunchKernelGGL(resconv_mfma_kernel, grid, dim3(256), 0, s, (const bf16_t*)v, ldv, v_bs, w, (bf16_t*)out, ldo, o_bs, n_p,
                       transpose, accumulate ? (early ? 2 : 1) : 0);
    return true;
}

bool resconv_wgrad_try_mfma(const void* v, long ldv, long v_bs, const void* dout, long ldo, long o_bs, float* dw, int B, int n_p,
                            int heads, int dh, int taps, int dt_v, int dt_o, hipStream_t s) {
    if (dt_v != MH_BF16 || dt_o != MH_BF16 || dh != RM_DH || taps != RM_TAPS) return false;
    if (ldv % 8 || v_bs % 8 || ldo % 8 || o_bs % 8 || ((uintptr_t)v & 15) || ((uintptr_t)dout & 15)) return false;
    const int nblk = mh_cdiv(n_p, 32);
    int splits = 1;
    // ~one workgroup per CU: every extra split adds a round of LDS + global atomics (16 splits: 64 us, 2: 37 us at c2)
    constexpr int target = 256;      // workgroups aimed at
    while (splits * 2 * heads * B <= target && splits * 2 * 4 <= nblk) splits *= 2;
    dim3 grid(splits, heads, B);
    hipLaunchKernelGGL(resconv_wgrad_mfma_kernel, grid, dim3(256), 0, s, (const bf16_t*)v, ldv, v_bs, (const bf16_t*)dout, ldo, o_bs,
                       dw, n_p);
    return true;
}
