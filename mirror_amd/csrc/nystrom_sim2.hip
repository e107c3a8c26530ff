// attn2 of [3P] NystromAttention and everything the Moore-Penrose chain needs from it, in ONE launch (m = 256, dh = 64):
//     sim2 = scale q_l k_l^T ; attn2 = softmax(sim2, -1)                          (models/mirror.py:312 -> nystrom_attention)
//     moore_penrose_iter_pinv's start: z0 = attn2^T / (max_i sum_j |attn2| * max_j sum_i |attn2|), maxima over the WHOLE tensor
// The composed path ran five dependent launches between the landmarks and the chain's fork (batched 256 x 256 x 64 GEMM, row
// softmax, abs-sum maxima, z0 / panel packing): ~80 us per layer of tiny kernels with the chip idle and the chain waiting.
// One 256-thread workgroup per (batch, head); wave w owns rows [64 w, 64 w + 64) of the 256 x 256 result:
//   pass 1  S = q_l k_l^T (64 MFMAs per wave, operands straight from global as fragments), row softmax in the accumulators
//           (row = registers, column = lane: maxima / sums over the 8 column blocks + a 32-lane butterfly), row / column abs sums
//           -> packed (value, index) maxima by atomicMax, attn2 (f32, row-major: the backward's operand) and PN(attn2) (bf16, the
//           chain's X operand: the accumulator layout IS the panel layout, pinv_panel.hip);
//   pass 2  S^T = k_l q_l^T with the same fragments in the other roles (bit-identical dot products), exponentiated against pass 1's
//           row statistics (through LDS) -> PN(attn2^T) in f32, UNSCALED: the chain forward multiplies by 1 / (c r) when it loads
//           z_0 (the maxima are only complete when every workgroup of this launch has finished).
#include <cstdlib>
#include "gemm_kernel.h"

namespace {

#ifndef SIM2_EXP
#define SIM2_EXP 0               // timing experiments only (tools/exp/time_sim2.sh rebuilds this file with -DSIM2_EXP=n; results wrong for n > 1)
#endif
constexpr int SM = 256;          // landmarks
constexpr int SDH = 64;          // head dim
constexpr long SMAT = (long)SM * SM;

// Reductions over the 32 lanes of this lane's half.  __shfl_xor is a ds_bpermute (an LDS-pipe round trip) per step and five dependent
// steps per reduction, 96 reductions per wave: ~15 of the kernel's ~54 us.  The first four steps as DPP moves (VALU rate): xor 1 / xor 2
// inside a quad, then the mirror of a row half and of a row (every lane already holds its group's sum, so a mirror is as good as a
// butterfly); the last one (lanes 16 apart) as ds_swizzle.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float swz16_f(float v) {          // lane ^ 16 (bit-mask mode: and 0x1f, or 0, xor 0x10)
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
__device__ __forceinline__ float half_max(float v) {      // over the 32 lanes of this lane's half
    v = fmaxf(v, dpp_f<0xB1>(v));       // quad_perm [1, 0, 3, 2]
    v = fmaxf(v, dpp_f<0x4E>(v));       // quad_perm [2, 3, 0, 1]
    v = fmaxf(v, dpp_f<0x141>(v));      // row_half_mirror
    v = fmaxf(v, dpp_f<0x140>(v));      // row_mirror
    return fmaxf(v, swz16_f(v));
}
__device__ __forceinline__ float half_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return v + swz16_f(v);
}
__device__ __forceinline__ u32x4 pack8(const f32x16& a, int t) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; w++) o[w] = pack_bf2(a[8 * t + 2 * w], a[8 * t + 2 * w + 1]);
    return o;
}

// MASKED (round 5, BASELINE config 4): mlm [B, m] holds the valid-landmark flags of the key-padding mask; an entry whose row or column
// landmark is invalid is filled with -FLT_MAX in front of the softmax ([3P] `sim2.masked_fill_(~(mask_l[..., None] * mask_l[..., None, :]),
// -finfo.max)`), so a fully masked row comes out uniform — what mh_softmax_masked_fwd does on the composed path
template <bool MASKED>
__global__ __launch_bounds__(256) void nys_sim2_kernel(const bf16_t* __restrict__ lm, long ld, int D, int heads, float sl2, float* __restrict__ a2,
                                                       bf16_t* __restrict__ xp, float* __restrict__ z0f, unsigned long long* __restrict__ stats,
                                                       const float* __restrict__ mlm) {
    __shared__ float s_max[SM], s_inv[SM];
    __shared__ float s_mlm[MASKED ? SM : 1];
    __shared__ float s_col[4][SM];
    __shared__ unsigned long long s_best[2];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hl = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const bf16_t* ql = lm + (long)b * SM * ld + h * SDH;
    const bf16_t* kl = ql + D;
    if (tid < 2) s_best[tid] = 0ull;
    if constexpr (MASKED) s_mlm[tid] = mlm[(long)b * SM + tid];
    __syncthreads();

    // fragments: lane (r, hl) holds k = 16 ks + 8 hl .. + 7 of row 32 blk + r (the A and the B operand of 32x32x16 read a
    // row-major [row][k] source the same way)
    bf16x8 kf[8][4], qa[2][4];
#pragma unroll
    for (int cb = 0; cb < 8; cb++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) kf[cb][ks] = *reinterpret_cast<const bf16x8*>(kl + (long)(32 * cb + r) * ld + 16 * ks + 8 * hl);
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qa[rb][ks] = *reinterpret_cast<const bf16x8*>(ql + (long)(32 * (2 * wave + rb) + r) * ld + 16 * ks + 8 * hl);

    f32x16 acc[2][8];
    // ---------------------------------------------------------------- pass 1: S, row softmax
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int cb = 0; cb < 8; cb++) {
            f32x16 c;
#pragma unroll
            for (int e = 0; e < 16; e++) c[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[rb][ks], kf[cb][ks], c, 0, 0, 0);
            acc[rb][cb] = c * sl2;                        // log2 domain: exp2 below
        }
    unsigned long long best_r = 0ull;
    float cs[8];
#pragma unroll
    for (int cb = 0; cb < 8; cb++) cs[cb] = 0.f;
    unsigned colv = 0xffu;            // MASKED: bit cb = this lane's column 32 cb + r is a valid landmark
    if constexpr (MASKED) {
        colv = 0;
#pragma unroll
        for (int cb = 0; cb < 8; cb++) colv |= (s_mlm[32 * cb + r] != 0.f ? 1u : 0u) << cb;
    }
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            if constexpr (MASKED) {
                const bool rv = s_mlm[32 * (2 * wave + rb) + (reg & 3) + 8 * (reg >> 2) + 4 * hl] != 0.f;
#pragma unroll
                for (int cb = 0; cb < 8; cb++)
                    if (!(rv && ((colv >> cb) & 1u))) acc[rb][cb][reg] = -3.402823466e38f;
            }
            float mx = acc[rb][0][reg];
#pragma unroll
            for (int cb = 1; cb < 8; cb++) mx = fmaxf(mx, acc[rb][cb][reg]);
            if (SIM2_EXP != 2) mx = half_max(mx);
            float sm = 0.f;
#pragma unroll
            for (int cb = 0; cb < 8; cb++) {
                const float e = __builtin_amdgcn_exp2f(acc[rb][cb][reg] - mx);
                acc[rb][cb][reg] = e;
                sm += e;
            }
            if (SIM2_EXP != 2) sm = half_sum(sm);
            const float inv = 1.f / sm;
#pragma unroll
            for (int cb = 0; cb < 8; cb++) {
                const float p = acc[rb][cb][reg] * inv;
                acc[rb][cb][reg] = p;
                cs[cb] += p;
            }
            const float rs = sm * inv;                    // sum_j |attn2[i][j]| = 1 within rounding, as the sum of the rounded
                                                          // probabilities is (a third butterfly per row bought nothing: 5 us)
            const int row = 32 * (2 * wave + rb) + (reg & 3) + 8 * (reg >> 2) + 4 * hl;
            if (r == 0) { s_max[row] = mx; s_inv[row] = inv; }
            const unsigned long long pk = ((unsigned long long)__float_as_uint(rs) << 32) | (unsigned)(bh * SM + row);
            best_r = pk > best_r ? pk : best_r;
        }
    // column sums: this wave's 64 rows -> LDS, folded over the four waves below
#pragma unroll
    for (int cb = 0; cb < 8; cb++) {
        const float t = cs[cb] + __shfl_xor(cs[cb], 32, 64);
        if (hl == 0) s_col[wave][32 * cb + r] = t;
    }
    // outputs of pass 1
    float* a2b = a2 + bh * SMAT;
    bf16_t* xpb = xp + bh * SMAT;
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int cb = 0; cb < 8; cb++) {
            if (SIM2_EXP == 4) continue;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = 32 * (2 * wave + rb) + (reg & 3) + 8 * (reg >> 2) + 4 * hl;
                a2b[(long)row * SM + 32 * cb + r] = acc[rb][cb][reg];
            }
            // panel native: PN[jblk = cb][T][lane][8], T = 16-row k-step of the chain's products: rows 16 T + 4 hl + {0..3}, + 8
#pragma unroll
            for (int t = 0; t < 2; t++)
                *reinterpret_cast<u32x4*>(xpb + ((cb * 16 + 2 * (2 * wave + rb) + t) * 512) + (lane << 3)) = pack8(acc[rb][cb], t);
        }
    if (r == 0) atomicMax(&s_best[0], best_r);
    __syncthreads();
    {
        const float tot = s_col[0][tid] + s_col[1][tid] + s_col[2][tid] + s_col[3][tid];      // sum_i |attn2[i][tid]|
        unsigned long long pk = ((unsigned long long)__float_as_uint(tot) << 32) | (unsigned)(bh * SM + tid);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(pk, o, 64);
            pk = other > pk ? other : pk;
        }
        if (lane == 0) atomicMax(&s_best[1], pk);
    }
    // ---------------------------------------------------------------- pass 2: S^T against pass 1's row statistics
    // A = k_l rows [64 w, +64) = kf[2 w + rb] ; B = q_l, all 256 rows (read again: kf's registers are reused)
    bf16x8 ka[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) ka[rb][ks] = *reinterpret_cast<const bf16x8*>(kl + (long)(32 * (2 * wave + rb) + r) * ld + 16 * ks + 8 * hl);
#pragma unroll
    for (int cb = 0; cb < 8; cb++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) kf[cb][ks] = *reinterpret_cast<const bf16x8*>(ql + (long)(32 * cb + r) * ld + 16 * ks + 8 * hl);
    float* z0b = z0f + bh * SMAT;
#pragma unroll
    for (int cb = 0; cb < 8; cb++) {
        if (SIM2_EXP == 3 || z0f == nullptr) break;        // z0f NULL: the chain forward reads attn2's rows itself (mh_pinv_chain_fwd z0_rowmajor)
        const float mxi = s_max[32 * cb + r], ivi = s_inv[32 * cb + r];      // statistics of attn2's row i = this lane's column
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            f32x16 c;
#pragma unroll
            for (int e = 0; e < 16; e++) c[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[rb][ks], kf[cb][ks], c, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; e++) c[e] = __builtin_amdgcn_exp2f(c[e] * sl2 - mxi) * ivi;          // attn2[i][j] at (row j, column i)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                float* dst = z0b + ((long)(cb * 16 + 2 * (2 * wave + rb) + t) * 64 + lane) * 8;
                *reinterpret_cast<f32x4*>(dst) = f32x4{c[8 * t], c[8 * t + 1], c[8 * t + 2], c[8 * t + 3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{c[8 * t + 4], c[8 * t + 5], c[8 * t + 6], c[8 * t + 7]};
            }
        }
    }
    __syncthreads();
    if (tid < 2) atomicMax(stats + tid, s_best[tid]);
}

// ------------------------------------------------------------------------------------------------------------------------------
// The two 256 x 256 x 64 products that open NystromAttention's backward around the Moore-Penrose chain, one launch per pass instead
// of a batched GEMM + a packing pass + another batched GEMM (out = attn1 (Z (attn3 v)): w2 = Z av):
//     dZ = dW2 av^T  -> the chain's input U = dZ^T = av dW2^T, written panel native in bf16 (as in pass 1 above, the accumulator
//                       layout of the product IS the panel layout: no f32 dZ in HBM, no packing launch);
//     dAV = Z^T dW2  =  zfT dW2   (zfT[j][i] = Z[i][j]: the chain's column-major output).
// dW2, av: f32 [B h, 256, 64] (rounded to bf16 as MFMA operands, what mh_gemm does with f32 operands); dAV: bf16 [B h, 256, 64].
constexpr int DWP = 72;          // pitch of the [256][64] bf16 image of dW2 (K-contiguous fragments, conflict free)
constexpr int DTP = 264;         // pitch of its transpose [64][256]
__device__ __forceinline__ bf16x8 cvt8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; e++) { o[e] = (__bf16)a[e]; o[4 + e] = (__bf16)b[e]; }
    return o;
}
__global__ __launch_bounds__(256) void nys_dz_dav_kernel(const float* __restrict__ dw2, const float* __restrict__ av,
                                                         const bf16_t* __restrict__ zfT, bf16_t* __restrict__ up, bf16_t* __restrict__ dav,
                                                         float* __restrict__ delta3) {
    __shared__ __attribute__((aligned(16))) bf16_t s_dw[SM * DWP];
    __shared__ __attribute__((aligned(16))) bf16_t s_dt[SDH * DTP];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hl = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.x;
    const int rb = blockIdx.y;          // two workgroups per matrix (256 for B h = 128: every CU), one of a wave's two row blocks each
    const float* dwb = dw2 + (long)bh * SM * SDH;
    const float* avb = av + (long)bh * SM * SDH;
    // this wave's A operands are requested first: av rows (pass 1) and zfT rows (pass 2) of its two row blocks
    bf16x8 af[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) af[ks] = cvt8(avb + (long)(32 * (2 * wave + rb) + r) * SDH + 16 * ks + 8 * hl);
    bf16x8 zf[16];
#pragma unroll
    for (int ks = 0; ks < 16; ks++)
        zf[ks] = *reinterpret_cast<const bf16x8*>(zfT + (long)bh * SMAT + (long)(32 * (2 * wave + rb) + r) * SM + 16 * ks + 8 * hl);
    // dW2 -> bf16 images, [i][d] and [d][i]
#pragma unroll
    for (int n = 0; n < 16; n++) {
        const int e0 = 4 * (tid + 256 * n), i = e0 >> 6, d = e0 & 63;
        const f32x4 v = *reinterpret_cast<const f32x4*>(dwb + e0);
        unsigned short h[4];
#pragma unroll
        for (int e = 0; e < 4; e++) h[e] = f2bf(v[e]);
        *reinterpret_cast<u32x2*>(s_dw + i * DWP + d) = u32x2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
#pragma unroll
        for (int e = 0; e < 4; e++) s_dt[(d + e) * DTP + i] = h[e];
    }
    __syncthreads();
    // pass 1: U = av dW2^T, rows i of this wave's two row blocks x all 256 columns j
    bf16_t* upb = up + (long)bh * SMAT;
#pragma unroll
        for (int cb = 0; cb < 8; cb++) {
            f32x16 c;
#pragma unroll
            for (int e = 0; e < 16; e++) c[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(s_dw + (32 * cb + r) * DWP + 16 * ks + 8 * hl);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], b, c, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 2; t++)
                *reinterpret_cast<u32x4*>(upb + ((cb * 16 + 2 * (2 * wave + rb) + t) * 512) + (lane << 3)) = pack8(c, t);
        }
    // pass 2: dAV = zfT dW2, rows j of the same row blocks x 64 columns d
    bf16_t* davb = dav + (long)bh * SM * SDH;
    {
        float dl[16];       // delta3[j] = sum_d dAV[j][d] av[j][d] with the rounded dAV (what mh_nys_attn3_bwd's first launch computes)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) dl[reg] = 0.f;
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
            f32x16 c;
#pragma unroll
            for (int e = 0; e < 16; e++) c[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 16; ks++) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(s_dt + (32 * nb + r) * DTP + 16 * ks + 8 * hl);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[ks], b, c, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = 32 * (2 * wave + rb) + (reg & 3) + 8 * (reg >> 2) + 4 * hl;
                const unsigned short o = f2bf(c[reg]);
                davb[(long)row * SDH + 32 * nb + r] = o;
                if (delta3) dl[reg] += __uint_as_float((unsigned)o << 16) * avb[(long)row * SDH + 32 * nb + r];
            }
        }
        if (delta3) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const float d = half_sum(dl[reg]);
                if (r == 0) delta3[(long)bh * SM + 32 * (2 * wave + rb) + (reg & 3) + 8 * (reg >> 2) + 4 * hl] = d;
            }
        }
    }
}

}  // namespace

extern "C" int mh_nys_sim2(const void* lm, float* a2, void* xp, float* z0f, uint64_t* stats64, int B, int m, int D, int heads, float scale,
                           int64_t lm_ld, const float* mlm, mh_stream s) {
    MH_REQUIRE(lm_ld == 0 || (lm_ld >= 2L * D && lm_ld % 8 == 0), "mh_nys_sim2: lm_ld must be 0 (contiguous [B, m, 2D]) or a multiple of 8 >= 2 D");
    MH_REQUIRE(m == SM && heads >= 1 && D == heads * SDH, "mh_nys_sim2: built for m = %d landmarks and dh = %d (m=%d, D=%d, heads=%d)", SM, SDH, m, D, heads);
    MH_REQUIRE(lm && a2 && xp && stats64 && (((uintptr_t)lm | (uintptr_t)a2 | (uintptr_t)xp | (uintptr_t)z0f) & 15) == 0,
               "mh_nys_sim2: null / unaligned buffer");
    MH_REQUIRE((long)B * heads * m < (1L << 31), "mh_nys_sim2: index overflow");
    if (B == 0) return MH_OK;
    if (mlm) {
        MH_REQUIRE(z0f == nullptr, "mh_nys_sim2: the masked form has no second pass (z0f must be NULL: the chain forms z_0 from attn2's rows)");
        hipLaunchKernelGGL(nys_sim2_kernel<true>, dim3(B * heads, 1), dim3(256), 0, (hipStream_t)s, (const bf16_t*)lm, lm_ld > 0 ? (long)lm_ld : 2L * D, D, heads, scale * 1.4426950408889634f,
                           a2, (bf16_t*)xp, z0f, (unsigned long long*)stats64, mlm);
        MH_LAUNCH_CHECK("mh_nys_sim2");
        return MH_OK;
    }
    hipLaunchKernelGGL(nys_sim2_kernel<false>, dim3(B * heads, 1), dim3(256), 0, (hipStream_t)s, (const bf16_t*)lm, lm_ld > 0 ? (long)lm_ld : 2L * D, D, heads, scale * 1.4426950408889634f,
                       a2, (bf16_t*)xp, z0f, (unsigned long long*)stats64, (const float*)nullptr);
    MH_LAUNCH_CHECK("mh_nys_sim2");
    return MH_OK;
}

extern "C" int mh_nys_dz_dav(const float* dw2, const float* av, const void* zfT, void* up, void* dav, float* delta3, int BH, int m, int dh,
                             mh_stream s) {
    MH_REQUIRE(m == SM && dh == SDH, "mh_nys_dz_dav: built for m = %d landmarks and dh = %d (m=%d, dh=%d)", SM, SDH, m, dh);
    MH_REQUIRE(dw2 && av && zfT && up && dav && (((uintptr_t)dw2 | (uintptr_t)av | (uintptr_t)zfT | (uintptr_t)up | (uintptr_t)dav) & 15) == 0,
               "mh_nys_dz_dav: null / unaligned buffer");
    if (BH == 0) return MH_OK;
    hipLaunchKernelGGL(nys_dz_dav_kernel, dim3(BH, 2), dim3(256), 0, (hipStream_t)s, dw2, av, (const bf16_t*)zfT, (bf16_t*)up, (bf16_t*)dav, delta3);
    MH_LAUNCH_CHECK("mh_nys_dz_dav");
    return MH_OK;
}
