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
#include "gemm_kernel.h"

namespace {

constexpr int SM = 256;          // landmarks
constexpr int SDH = 64;          // head dim
constexpr long SMAT = (long)SM * SM;

__device__ __forceinline__ float half_max(float v) {      // over the 32 lanes of this lane's half
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ u32x4 pack8(const f32x16& a, int t) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; w++) o[w] = (unsigned)f2bf(a[8 * t + 2 * w]) | ((unsigned)f2bf(a[8 * t + 2 * w + 1]) << 16);
    return o;
}

__global__ __launch_bounds__(256) void nys_sim2_kernel(const bf16_t* __restrict__ lm, int D, int heads, float sl2, float* __restrict__ a2,
                                                       bf16_t* __restrict__ xp, float* __restrict__ z0f, unsigned long long* __restrict__ stats) {
    __shared__ float s_max[SM], s_inv[SM];
    __shared__ float s_col[4][SM];
    __shared__ unsigned long long s_best[2];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hl = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const long ld = 2L * D;
    const bf16_t* ql = lm + (long)b * SM * ld + h * SDH;
    const bf16_t* kl = ql + D;
    if (tid < 2) s_best[tid] = 0ull;
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
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            float mx = acc[rb][0][reg];
#pragma unroll
            for (int cb = 1; cb < 8; cb++) mx = fmaxf(mx, acc[rb][cb][reg]);
            mx = half_max(mx);
            float sm = 0.f;
#pragma unroll
            for (int cb = 0; cb < 8; cb++) {
                const float e = exp2f(acc[rb][cb][reg] - mx);
                acc[rb][cb][reg] = e;
                sm += e;
            }
            sm = half_sum(sm);
            const float inv = 1.f / sm;
            float rs = 0.f;
#pragma unroll
            for (int cb = 0; cb < 8; cb++) {
                const float p = acc[rb][cb][reg] * inv;
                acc[rb][cb][reg] = p;
                rs += p;
                cs[cb] += p;
            }
            rs = half_sum(rs);                            // sum_j |attn2[i][j]| (probabilities: no abs needed)
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
        const float mxi = s_max[32 * cb + r], ivi = s_inv[32 * cb + r];      // statistics of attn2's row i = this lane's column
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            f32x16 c;
#pragma unroll
            for (int e = 0; e < 16; e++) c[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[rb][ks], kf[cb][ks], c, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; e++) c[e] = exp2f(c[e] * sl2 - mxi) * ivi;          // attn2[i][j] at (row j, column i)
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

}  // namespace

extern "C" int mh_nys_sim2(const void* lm, float* a2, void* xp, float* z0f, uint64_t* stats64, int B, int m, int D, int heads, float scale,
                           mh_stream s) {
    MH_REQUIRE(m == SM && heads >= 1 && D == heads * SDH, "mh_nys_sim2: built for m = %d landmarks and dh = %d (m=%d, D=%d, heads=%d)", SM, SDH, m, D, heads);
    MH_REQUIRE(lm && a2 && xp && z0f && stats64 && (((uintptr_t)lm | (uintptr_t)a2 | (uintptr_t)xp | (uintptr_t)z0f) & 15) == 0,
               "mh_nys_sim2: null / unaligned buffer");
    MH_REQUIRE((long)B * heads * m < (1L << 31), "mh_nys_sim2: index overflow");
    if (B == 0) return MH_OK;
    hipLaunchKernelGGL(nys_sim2_kernel, dim3(B * heads), dim3(256), 0, (hipStream_t)s, (const bf16_t*)lm, D, heads, scale * 1.4426950408889634f,
                       a2, (bf16_t*)xp, z0f, (unsigned long long*)stats64);
    MH_LAUNCH_CHECK("mh_nys_sim2");
    return MH_OK;
}
