// Fused Nystrom attention sides ([3P] NystromAttention.forward, called at models/mirror.py:312) for the TransMIL
// geometry dh = 64, m = 256 landmarks, bf16 operands.
//
// The two big similarity matrices  sim1 = q k_l^T  [n_p x m]  and  sim3 = q_l k^T  [m x n_p]  are never written to
// HBM.  As separate GEMM / softmax / GEMM launches each of them costs ~1.7 GB of traffic per layer at B = 16
// (f32 logits out, softmax in/out, probabilities in again) for 36 GFLOP of work; fused, the probabilities live in
// MFMA accumulators from the first product to the second and the traffic is q, k, v, out (~0.3 GB).
//
//   attn1 side:  out = softmax_m(scale q k_l^T) w2                (w2 = pinv(attn2) (attn3 v), [m x dh])
//   attn3 side:  av  = softmax_n(scale q_l k^T) v                 (online softmax over the n_p keys)
//
// Register-only data flow: v_mfma_f32_32x32x16_bf16 leaves C[i][j] with j = lane & 31 and i in registers
// (i = 8 (r >> 2) + 4 (lane >> 5) + (r & 3)).  That is exactly the operand layout of a matrix whose CONTRACTION index
// is i, so a tile of probabilities is fed straight back as an operand: registers 8t..8t+7 form the bf16x8 fragment of
// k-step t, and the other operand is read from LDS in the same permuted k order (rows kb + 4 hl + {0..3} and
// kb + 8 + 4 hl + {0..3}) with ds_read_b64_tr_b16.  Which index must be contracted decides the orientation:
//   N kernels (the landmark images are staged once per workgroup, then every wave walks its own 32-row blocks of the
//   sequence: n in lanes, the 256 landmarks pass through registers 32 at a time):
//       attn1 fwd (online softmax over landmark blocks), attn1 bwd dq (+ delta), attn3 bwd dk/dv
//   L kernels (wave owns 64 landmark columns in lanes and walks 128-row tiles of the sequence, n in registers; the
//   sequence is cut into ranges over several workgroups per (b, h)):
//       attn3 fwd (online softmax + nys_a3_combine), attn1 bwd dw2/dk_l, attn3 bwd dq_l
// Every kernel has a MASKED instance for the package's key-padding mask (BASELINE config 4), see Geo.
// Layout (SURVEY.md §8 / DESIGN.md §4): qkv [B, n_p, 3D] bf16, heads are 64-wide column slices; landmarks lm
// [B, m, 2D] = q_l | k_l; w2, av, dav [B, h, m, 64]; out / dout [B, n_p, D].
#include <cstdlib>
#include <cstring>
#include "gemm_kernel.h"

namespace {

constexpr int NM = 256;   // landmarks
constexpr int ND = 64;    // head dim
constexpr int NP = 72;    // LDS pitch in bf16 of every [rows][64] image: ds_read_b128 fragments conflict free
constexpr int NT = 256;   // threads per workgroup (4 waves)
constexpr int TR = 128;   // sequence rows per tile

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef short s16x8 __attribute__((ext_vector_type(8)));

// A/B fragment, contraction index contiguous in the image: row (row0 + lane&31), k = k0 + 8 hl + {0..7}
__device__ __forceinline__ bf16x8 frag_kc(const bf16_t* img, int row0, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + (row0 + (lane & 31)) * NP + k0 + 8 * (lane >> 5));
}
// A/B fragment, contraction index = image ROW, free index = image column (col0 + lane&31), in the accumulator's k
// order: element j <-> image row kb + 4 hl + j (j < 4), kb + 8 + 4 hl + (j - 4) (j >= 4)
__device__ __forceinline__ bf16x8 frag_tr(const bf16_t* img, int col0, int kb, int lane) {
    const int g16 = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const bf16_t* a0 = img + (kb + 4 * (g16 >> 1) + q) * NP + col0 + 16 * (g16 & 1) + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 8 * NP));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// fragment straight from global memory: 8 consecutive bf16 of a row
__device__ __forceinline__ bf16x8 frag_g(const bf16_t* rowptr, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(rowptr + k0 + 8 * (lane >> 5));
}
// accumulator registers 8t..8t+7 -> bf16x8 operand fragment of k-step t
template <int T>
__device__ __forceinline__ bf16x8 pack8(const f32x16& a) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = (__bf16)a[8 * T + j];
    return r;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0.f;
    return z;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// The softmax arithmetic between the two products is what these kernels are bound by (the first version spent ~27 VALU
// slots per MFMA): logits are scaled by scale * log2(e) so the exponential is a bare v_exp_f32, whole accumulator tiles
// are processed as vectors so the compiler emits v_pk_fma / v_pk_mul / v_pk_add, and the file is built with
// -mllvm -amdgpu-mfma-vgpr-form so accumulators stay in VGPRs instead of bouncing through v_accvgpr_read / write.
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
typedef float f32x8v __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vsum16(const f32x16& a) {
    const f32x8v t8 = __builtin_shufflevector(a, a, 0, 1, 2, 3, 4, 5, 6, 7) + __builtin_shufflevector(a, a, 8, 9, 10, 11, 12, 13, 14, 15);
    const f32x4 t4 = __builtin_shufflevector(t8, t8, 0, 1, 2, 3) + __builtin_shufflevector(t8, t8, 4, 5, 6, 7);
    const f32x2v t2 = __builtin_shufflevector(t4, t4, 0, 1) + __builtin_shufflevector(t4, t4, 2, 3);
    return t2[0] + t2[1];
}
__device__ __forceinline__ float vmax16(const f32x16& a) {
    float m0 = fmaxf(fmaxf(a[0], a[1]), a[2]), m1 = fmaxf(fmaxf(a[3], a[4]), a[5]), m2 = fmaxf(fmaxf(a[6], a[7]), a[8]);
    float m3 = fmaxf(fmaxf(a[9], a[10]), a[11]), m4 = fmaxf(fmaxf(a[12], a[13]), a[14]);
    return fmaxf(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), fmaxf(m4, a[15]));
}
__device__ __forceinline__ void exp2_16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = __builtin_amdgcn_exp2f(a[r]);
}
// a * m + b with m, b the same for all 16 elements: pairs of elements, so the compiler emits 8 v_pk_fma_f32 (written as
// `a * m - b` it first splats b into 16 registers and negates every one of them)
__device__ __forceinline__ f32x16 fma_splat(const f32x16& a, float m, float b) {
    const f32x2v mm = {m, m}, bb = {b, b};
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        f32x2v x = {a[i], a[i + 1]};
        x = __builtin_elementwise_fma(x, mm, bb);
        r[i] = x[0];
        r[i + 1] = x[1];
    }
    return r;
}
constexpr float NEG_BIG = -1e30f;   // "minus infinity" of the running maxima (the file is built with -fno-honor-nans)
// accumulator registers <-> rows 8 (r >> 2) + 4 hl + (r & 3): the 16 per-row values of `src` (LDS, f32) for this lane
__device__ __forceinline__ f32x16 rowvals16(const float* src, int hl) {
    f32x16 v;
#pragma unroll
    for (int gq = 0; gq < 4; gq++) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(src + 8 * gq + 4 * hl);
        v[4 * gq] = x[0]; v[4 * gq + 1] = x[1]; v[4 * gq + 2] = x[2]; v[4 * gq + 3] = x[3];
    }
    return v;
}

// ROWS x 64 bf16 (row stride ld) -> pitch-NP image
template <int ROWS>
__device__ __forceinline__ void stage_rows(bf16_t* img, const bf16_t* __restrict__ src, long ld, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; i++) {
        const int cid = tid + i * NT, r = cid >> 3, c = cid & 7;
        *reinterpret_cast<u32x4*>(img + r * NP + c * 8) = *reinterpret_cast<const u32x4*>(src + (long)r * ld + c * 8);
    }
}
// the same in two halves so the next tile can wait in registers while the current one is consumed
template <int ROWS>
__device__ __forceinline__ void tile_load(u32x4 (&regs)[ROWS * 8 / NT], const bf16_t* __restrict__ src, long ld, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; i++) {
        const int cid = tid + i * NT, r = cid >> 3, c = cid & 7;
        regs[i] = *reinterpret_cast<const u32x4*>(src + (long)r * ld + c * 8);
    }
}
template <int ROWS>
__device__ __forceinline__ void tile_store(const u32x4 (&regs)[ROWS * 8 / NT], bf16_t* img, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; i++) {
        const int cid = tid + i * NT, r = cid >> 3, c = cid & 7;
        *reinterpret_cast<u32x4*>(img + r * NP + c * 8) = regs[i];
    }
}
// All 16 accumulator rows of a [d][n]-oriented result (d = 32 nb ..) -> this lane's row p[0 .. 31] as TWO 16-byte stores instead of four
// 8-byte ones: lanes c and c + 32 hold columns 8 gq + {0..3} and 8 gq + 4 + {0..3} of the same row, so they trade packed words
// (v_permlane32_swap: the upper lane half of one register against the lower half of another) until the lower lane owns column groups
// 0 and 2 whole and the upper lane groups 1 and 3.  (The outputs of these kernels leave as per-lane pieces of 32 different rows; the
// L2 has to merge them into lines, and the store tail is issue bound: half the instructions.)
__device__ __forceinline__ void store_row8(bf16_t* p, const f32x16& a, int hl) {
    unsigned w[4][2];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        w[g][0] = pack_bf2(a[4 * g + 0], a[4 * g + 1]);
        w[g][1] = pack_bf2(a[4 * g + 2], a[4 * g + 3]);
    }
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {          // column groups (2 pr, 2 pr + 1)
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(w[2 * pr][0], w[2 * pr + 1][0], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(w[2 * pr][1], w[2 * pr + 1][1], false, false);
        // lower lane: s?[0] = its own words of group 2 pr, s?[1] = the upper lane's words of that group; upper lane: s?[0] = the lower
        // lane's words of group 2 pr + 1, s?[1] = its own
        const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
        *reinterpret_cast<u32x4*>(p + 8 * (2 * pr + hl)) = v;
    }
}

// Key-padding mask ([3P] NystromAttention.forward(x, mask=...), BASELINE config 4): mrow [B, n_p] and mlm [B, m] hold 0 / 1
// floats (valid sequence rows / landmark groups with at least one valid row); an entry of sim1 (row n, landmark l) or
// sim3 (landmark l, key n) is valid iff mrow[n] and mlm[l].  Invalid logits take NEG_FILL — the package's
// masked_fill_(-finfo.max) in log2 units, small enough in magnitude that `fill + log2(row sum)` stays exact, so the
// backward's exp2(fill - lse) reproduces the uniform row of a fully masked query — and their gradient is zero.
constexpr float NEG_FILL = -1.0e6f;
__device__ __forceinline__ void mask_fill16(f32x16& l2, const f32x16& vr, float vl) {
#pragma unroll
    for (int e = 0; e < 16; e++) l2[e] = (vr[e] * vl != 0.f) ? l2[e] : NEG_FILL;
}
__device__ __forceinline__ void mask_zero16(f32x16& d, const f32x16& vr, float vl) {
#pragma unroll
    for (int e = 0; e < 16; e++) d[e] = (vr[e] * vl != 0.f) ? d[e] : 0.f;
}

// e4m3 copy of attn1's output for the fp8 forward of to_out (BASELINE config 5): delayed per-tensor scaling as in
// mh_quant_fp8_delayed (ring of three per-site maxima rotated by the device-side step counter)
struct Q8Out {
    unsigned char* q;
    unsigned* ring;
    const float* tick;
    float margin;
    float* scale;
};

struct Geo {
    int h, n_p, D;
    float scale, scale2;    // scale2 = scale * log2(e)
    const float* mrow;      // key-padding mask rows [B, n_p] (NULL: no mask)
    const float* mlm;       // valid landmarks [B, m]
    int accumulate;     // attn1 forward: add to `out` (the res_conv term is already there) instead of overwriting it
    long lm_ld;         // row stride of `lm` in elements (2 D for a contiguous [B, m, 2D]; 3 D when the landmark rows live behind the
                        // sequence in the to_qkv output buffer); batch b's landmarks start at lm + b * m * lm_ld
    const float* rc_w;  // attn3 forward only: the 33-tap res_conv filters [h][33] (NULL: no res_conv) and the [B, n_p, D] buffer that
    bf16_t* rc_out;     // receives res_conv(v) (attn1's forward then adds its product to it)
};
constexpr int RC_TAPS = 33, RC_HALO = 16;

// ============================================================================ attn1 forward (N kernel)
// grid (splits, B h).  out[b, n, hd*64 + d] = sum_l softmax_l(scale q k_l^T)[n, l] w2[l, d];  lse1 = row logsumexp.
// The landmark images (k_l, w2: 64 KB) are staged ONCE per workgroup; after that every wave walks its own 32-row blocks
// of the sequence (block = first + i * 4 * splits) with the q fragments read straight from HBM one block ahead — no
// LDS writes and no barriers in the loop.  (One 128-row tile per workgroup spent 3x longer staging than computing.)
template <bool MASKED, bool Q8 = false>
__global__ __launch_bounds__(NT) void nys_a1_fwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                        const bf16_t* __restrict__ w2, bf16_t* __restrict__ out,
                                                        float* __restrict__ lse1, Geo g, Q8Out q8 = Q8Out{}, bf16_t* __restrict__ o1 = nullptr) {
    __shared__ __attribute__((aligned(16))) bf16_t s_kl_[NM * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_w2_[NM * NP];
    __shared__ __attribute__((aligned(16))) float s_mlm_[NM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    const long LD = g.lm_ld;
    float q8mul = 1.f, q8max = 0.f;
    int q8cur = 0;
    if constexpr (Q8) {
        const int t = (int)q8.tick[0];
        q8cur = t % 3;
        const float amax = __uint_as_float(q8.ring[(t + 2) % 3]) * q8.margin;
        q8mul = amax > 0.f ? 448.f / amax : 1.f;
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
            q8.scale[0] = amax > 0.f ? amax / 448.f : 1.f;
            q8.ring[(t + 1) % 3] = 0u;
        }
    }
    stage_rows<NM>(s_kl_, lm + (long)b * NM * LD + D + hd * ND, LD, tid);
    stage_rows<NM>(s_w2_, w2 + (long)bh * NM * ND, ND, tid);
    constexpr bool masked = MASKED;
    s_mlm_[tid] = masked ? g.mlm[(long)b * NM + tid] : 1.f;
    const int nblk = g.n_p / 32, stride = 4 * gridDim.x;
    int rb = 4 * blockIdx.x + wave;
    const bf16_t* qb = qkv + (long)b * g.n_p * 3 * D + hd * ND;
    bf16x8 qn[4];
    if (rb < nblk) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qn[ks] = frag_g(qb + (long)(32 * rb + c) * 3 * D, 16 * ks, lane);
    }
    __syncthreads();
#pragma unroll 1
    for (; rb < nblk; rb += stride) {
        const long row = 32L * rb + c;
        // the landmark fragments do not depend on the block: without an opaque base the compiler hoists all 64 LDS reads
        // out of the loop and keeps them in 256 registers
        int opq = 0;
        asm volatile("" : "+v"(opq));
        const bf16_t* s_kl = s_kl_ + opq;
        const bf16_t* s_w2 = s_w2_ + opq;
        const float* s_mlm = s_mlm_ + opq;
        const float mr = masked ? g.mrow[(long)b * g.n_p + row] : 1.f;
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qf[ks] = qn[ks];
        if (rb + stride < nblk) {
#pragma unroll
            for (int ks = 0; ks < 4; ks++) qn[ks] = frag_g(qb + (long)(32 * (rb + stride) + c) * 3 * D, 16 * ks, lane);
        }
        bf16_t* orow = out + ((long)b * g.n_p + row) * D + hd * ND;
        u32x2 old[2][4];      // the res_conv term already in `out`: fetched now, needed at the end of the block
        if (g.accumulate) {
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) old[nb][gq] = *reinterpret_cast<const u32x2*>(orow + 32 * nb + 8 * gq + 4 * hl);
        }
        // online softmax over the 8 landmark blocks: 16 logits live at a time instead of 128 (two waves per SIMD fit)
        float mrun = NEG_BIG, lrun = 0.f;
        f32x16 o[2] = {zero16(), zero16()};   // O^T[d][q row]
        // the logits of block blk + 1 are issued before the softmax arithmetic of block blk: the MFMA pipe works on them
        // while the VALU does max / exp2 / sum / pack (a wave is otherwise strictly MFMA -> VALU -> MFMA, and only one
        // other wave shares the SIMD)
        f32x16 snext = zero16();              // S^T[landmark 32 blk ..][q row]
#pragma unroll
        for (int ks = 0; ks < 4; ks++) snext = MFMA(frag_kc(s_kl, 0, 16 * ks, lane), qf[ks], snext);
#pragma unroll
        for (int blk = 0; blk < 8; blk++) {
            f32x16 sb = snext;
            if (blk + 1 < 8) {
                snext = zero16();
#pragma unroll
                for (int ks = 0; ks < 4; ks++) snext = MFMA(frag_kc(s_kl, 32 * (blk + 1), 16 * ks, lane), qf[ks], snext);
            }
            float sc2 = g.scale2;
            if (masked) {                                          // logits to log2 units first, invalid ones to NEG_FILL
                sb = sb * g.scale2;
                mask_fill16(sb, rowvals16(s_mlm + 32 * blk, hl), mr);
                sc2 = 1.f;
            }
            float mx = vmax16(sb);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(mrun, mx * sc2);              // running max in log2 units
            if (__builtin_amdgcn_ballot_w64(mnew != mrun)) {       // rescale only when some row's max moved
                const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
                lrun *= alpha;
                o[0] *= alpha;
                o[1] *= alpha;
                mrun = mnew;
            }
            sb = fma_splat(sb, sc2, -mrun);
            exp2_16(sb);
            lrun += vsum16(sb);               // per lane half; the halves are joined once at the end
            const bf16x8 p0 = pack8<0>(sb), p1 = pack8<1>(sb);
#pragma unroll
            for (int nb = 0; nb < 2; nb++) {
                o[nb] = MFMA(frag_tr(s_w2, 32 * nb, 32 * blk, lane), p0, o[nb]);
                o[nb] = MFMA(frag_tr(s_w2, 32 * nb, 32 * blk + 16, lane), p1, o[nb]);
            }
        }
        const float sum = lrun + __shfl_xor(lrun, 32, 64);
        if (hl == 0) lse1[(long)bh * g.n_p + row] = (mrun + __log2f(sum)) * LN2;
        const float inv = 1.f / sum;
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
            o[nb] *= inv;
            // o1: attn1's OWN rows (without the res_conv addend), bf16: the backward takes delta[n] = sum_l P dP = sum_d dO[n, d] O1[n, d]
            // from them (attn1 backward, dw2 kernel), so the dq kernel needs no first pass over dP and can run after the pinv chain's fork
            if (o1) store_row8(o1 + ((long)b * g.n_p + row) * D + hd * ND + 32 * nb, o[nb], hl);
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                if (g.accumulate) {
                    o[nb][4 * gq + 0] += __uint_as_float(old[nb][gq][0] << 16);
                    o[nb][4 * gq + 1] += __uint_as_float(old[nb][gq][0] & 0xffff0000u);
                    o[nb][4 * gq + 2] += __uint_as_float(old[nb][gq][1] << 16);
                    o[nb][4 * gq + 3] += __uint_as_float(old[nb][gq][1] & 0xffff0000u);
                }
                if constexpr (Q8) {       // the e4m3 copy of exactly the bf16 values stored below
                    float v4[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        v4[e] = bf2f(f2bf(o[nb][4 * gq + e]));
                        q8max = fmaxf(q8max, fabsf(v4[e]));
                        v4[e] = fminf(fmaxf(v4[e] * q8mul, -448.f), 448.f);
                    }
                    unsigned w = 0;
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(v4[0], v4[1], w, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(v4[2], v4[3], w, true);
                    *reinterpret_cast<unsigned*>(q8.q + ((long)b * g.n_p + row) * D + hd * ND + 32 * nb + 8 * gq + 4 * hl) = w;
                }
            }
            store_row8(orow + 32 * nb, o[nb], hl);
        }
    }
    if constexpr (Q8) {       // one atomic per wave that raises the step's maximum
        float m = q8max;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0 && m > 0.f) atomicMax(q8.ring + q8cur, __float_as_uint(m));
    }
}

// ============================================================================ attn1 backward, dq (N kernel)
// grid (splits, B h), same walk as the forward.  dS1 = P1 o (dO w2^T - delta) scale,  dq = dS1 k_l.
// delta[n] = sum_l P1 dP1 is an INPUT (round 5: the dw2 kernel below computes it as sum_d dO[n, d] O1[n, d] from attn1's own saved output
// rows, the flash-attention identity): with it known a 32-landmark block is finished in one go — S, dP, dS, dq — instead of all 8 blocks'
// probabilities waiting in 128 registers for a first pass over dP (3 products instead of 4), and nothing the pinv chain needs comes out of
// this kernel any more, so it runs BESIDE the chain (NystromCoreFn.backward) instead of in front of its fork.
template <bool MASKED>
__global__ __launch_bounds__(NT, 2) void nys_a1_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                           const bf16_t* __restrict__ w2, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ lse1, const float* __restrict__ delta1,
                                                           bf16_t* __restrict__ dqkv, Geo g) {
    __shared__ __attribute__((aligned(16))) bf16_t s_kl_[NM * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_w2_[NM * NP];
    __shared__ __attribute__((aligned(16))) float s_mlm_[NM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    const long LD = g.lm_ld;
    stage_rows<NM>(s_kl_, lm + (long)b * NM * LD + D + hd * ND, LD, tid);
    stage_rows<NM>(s_w2_, w2 + (long)bh * NM * ND, ND, tid);
    constexpr bool masked = MASKED;
    s_mlm_[tid] = masked ? g.mlm[(long)b * NM + tid] : 1.f;
    const int nblk = g.n_p / 32, stride = 4 * gridDim.x;
    int rb = 4 * blockIdx.x + wave;
    const bf16_t* qb = qkv + (long)b * g.n_p * 3 * D + hd * ND;
    const bf16_t* gb = dout + (long)b * g.n_p * D + hd * ND;
    bf16x8 qn[4], gn[4];
    float lsen = 0.f, deln = 0.f;
    if (rb < nblk) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            qn[ks] = frag_g(qb + (long)(32 * rb + c) * 3 * D, 16 * ks, lane);
            gn[ks] = frag_g(gb + (long)(32 * rb + c) * D, 16 * ks, lane);
        }
        lsen = lse1[(long)bh * g.n_p + 32 * rb + c];
        deln = delta1[(long)bh * g.n_p + 32 * rb + c];
    }
    __syncthreads();
#pragma unroll 1
    for (; rb < nblk; rb += stride) {
        const long row = 32L * rb + c;
        int opq = 0;                      // see nys_a1_fwd_kernel
        asm volatile("" : "+v"(opq));
        const bf16_t* s_kl = s_kl_ + opq;
        const bf16_t* s_w2 = s_w2_ + opq;
        const float* s_mlm = s_mlm_ + opq;
        const float mr = masked ? g.mrow[(long)b * g.n_p + row] : 1.f;
        bf16x8 qf[4], gf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) { qf[ks] = qn[ks]; gf[ks] = gn[ks]; }
        const float lse2 = lsen * LOG2E, dsc = deln * g.scale;
        if (rb + stride < nblk) {
            const long nrow = 32L * (rb + stride) + c;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                qn[ks] = frag_g(qb + nrow * 3 * D, 16 * ks, lane);
                gn[ks] = frag_g(gb + nrow * D, 16 * ks, lane);
            }
            lsen = lse1[(long)bh * g.n_p + nrow];
            deln = delta1[(long)bh * g.n_p + nrow];
        }
        f32x16 dq[2] = {zero16(), zero16()};   // dq^T[d][q row]
        // the two logits products of block blk + 1 are issued before the softmax arithmetic of block blk (as the forward does)
        f32x16 sn = zero16(), dpn = zero16();   // S^T, dP^T [landmark 32 blk ..][q row]
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            sn = MFMA(frag_kc(s_kl, 0, 16 * ks, lane), qf[ks], sn);
            dpn = MFMA(frag_kc(s_w2, 0, 16 * ks, lane), gf[ks], dpn);
        }
#pragma unroll
        for (int blk = 0; blk < 8; blk++) {
            f32x16 sb = sn, dp = dpn;
            if (blk + 1 < 8) {
                sn = zero16();
                dpn = zero16();
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    sn = MFMA(frag_kc(s_kl, 32 * (blk + 1), 16 * ks, lane), qf[ks], sn);
                    dpn = MFMA(frag_kc(s_w2, 32 * (blk + 1), 16 * ks, lane), gf[ks], dpn);
                }
            }
            if (masked) {
                sb = sb * g.scale2;
                mask_fill16(sb, rowvals16(s_mlm + 32 * blk, hl), mr);
                sb = fma_splat(sb, 1.f, -lse2);
            } else {
                sb = fma_splat(sb, g.scale2, -lse2);
            }
            exp2_16(sb);
            dp = sb * fma_splat(dp, g.scale, -dsc);
            if (masked) mask_zero16(dp, rowvals16(s_mlm + 32 * blk, hl), mr);
            const bf16x8 d0 = pack8<0>(dp), d1 = pack8<1>(dp);
#pragma unroll
            for (int nb = 0; nb < 2; nb++) {
                dq[nb] = MFMA(frag_tr(s_kl, 32 * nb, 32 * blk, lane), d0, dq[nb]);
                dq[nb] = MFMA(frag_tr(s_kl, 32 * nb, 32 * blk + 16, lane), d1, dq[nb]);
            }
        }
        bf16_t* drow = dqkv + ((long)b * g.n_p + row) * 3 * D + hd * ND;
#pragma unroll
        for (int nb = 0; nb < 2; nb++) store_row8(drow + 32 * nb, dq[nb], hl);
    }
}

// accumulator [rows in registers][cols in lanes] -> f32 atomics into dst[row * ld + col]
__device__ __forceinline__ void atomic_tile(float* dst, long ld, const f32x16& a, int hl, int c) {
#pragma unroll
    for (int r = 0; r < 16; r++) atomicAdd(dst + (long)(8 * (r >> 2) + 4 * hl + (r & 3)) * ld + c, a[r]);
}

// ============================================================================ attn1 backward, dw2 + dk_l (L kernel)
// grid (splits, B h); wave w owns landmarks [64 w, 64 w + 64).  dw2 = P1^T dO,  dk_l = dS1^T q  (f32 atomics).
// Also the producer of delta1[n] = sum_l P1 dP1 = sum_d dO[n, d] O1[n, d] (O1 = attn1's own output rows, saved by the forward): eight
// threads share a row of the dO / O1 tiles they stage anyway.  It runs FIRST (dw2 is what the pinv chain's backward waits for).
// 8 waves (two per SIMD, 32 landmarks each: round 5 — as 4 waves of 64 landmarks the kernel ran at one wave per SIMD with nothing to
// cover a wave's softmax arithmetic and LDS traffic).
constexpr int NTW = 512;
template <bool MASKED>
__global__ __launch_bounds__(NTW) void nys_a1_bwd_dw_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                            const bf16_t* __restrict__ w2, const bf16_t* __restrict__ dout,
                                                            const float* __restrict__ lse1, const bf16_t* __restrict__ o1,
                                                            float* __restrict__ delta1,
                                                            float* __restrict__ dw2, float* __restrict__ dlm, Geo g,
                                                            int tiles_per_wg) {
    __shared__ __attribute__((aligned(16))) bf16_t s_q[TR * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_g[TR * NP];
    __shared__ __attribute__((aligned(16))) float s_lse[TR];
    __shared__ __attribute__((aligned(16))) float s_del[TR];
    __shared__ __attribute__((aligned(16))) float s_mr[TR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    const long LD = g.lm_ld;
    const int ntiles = g.n_p / TR;
    const int t0 = blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, ntiles);
    if (t0 >= t1) return;
    constexpr bool masked = MASKED;
    const int lq = 32 * wave + c;                  // this lane's landmark
    const float ml = masked ? g.mlm[(long)b * NM + lq] : 1.f;
    const bf16_t* klb = lm + (long)b * NM * LD + D + hd * ND;
    const bf16_t* w2b = w2 + (long)bh * NM * ND;
    bf16x8 klf[4], w2f[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        klf[ks] = frag_g(klb + (long)lq * LD, 16 * ks, lane);
        w2f[ks] = frag_g(w2b + (long)lq * ND, 16 * ks, lane);
    }
    const bf16_t* qb = qkv + (long)b * g.n_p * 3 * D + hd * ND;
    const bf16_t* gb = dout + (long)b * g.n_p * D + hd * ND;
    const bf16_t* ob = o1 + (long)b * g.n_p * D + hd * ND;
    f32x16 adw[2] = {zero16(), zero16()}, adk[2] = {zero16(), zero16()};
    constexpr int NCH = TR * 8 / NTW;       // 16-byte pieces of a [128 x 64] tile per thread: rows (tid >> 3) + 64 i, columns 8 (tid & 7) ..
    const int pr = tid >> 3, pc = tid & 7;
    u32x4 rq[NCH], rg[NCH], ro[NCH];
    float rl = 0.f;
    auto load_tiles = [&](int t) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const long row = (long)t * TR + pr + 64 * i;
            rq[i] = *reinterpret_cast<const u32x4*>(qb + row * 3 * D + pc * 8);
            rg[i] = *reinterpret_cast<const u32x4*>(gb + row * D + pc * 8);
            ro[i] = *reinterpret_cast<const u32x4*>(ob + row * D + pc * 8);
        }
        if (tid < TR) rl = lse1[(long)bh * g.n_p + (long)t * TR + tid];
    };
    load_tiles(t0);
#pragma unroll 1
    for (int t = t0; t < t1; t++) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int r = pr + 64 * i;
            *reinterpret_cast<u32x4*>(s_q + r * NP + pc * 8) = rq[i];
            *reinterpret_cast<u32x4*>(s_g + r * NP + pc * 8) = rg[i];
            // delta of row r: eight threads (pc = 0 .. 7) hold its 64 columns of both the dO and the O1 tile
            float d = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                d += __uint_as_float(rg[i][w] << 16) * __uint_as_float(ro[i][w] << 16);
                d += __uint_as_float(rg[i][w] & 0xffff0000u) * __uint_as_float(ro[i][w] & 0xffff0000u);
            }
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            if (pc == 0) {
                s_del[r] = -d * g.scale;            // staged negated and scaled: the tile arithmetic is two multiply-adds per element
                delta1[(long)bh * g.n_p + (long)t * TR + r] = d;
            }
        }
        if (tid < TR) {
            s_lse[tid] = -rl * LOG2E;
            s_mr[tid] = masked ? g.mrow[(long)b * g.n_p + (long)t * TR + tid] : 1.f;
        }
        __syncthreads();
        if (t + 1 < t1) load_tiles(t + 1);
#pragma unroll
        for (int i = 0; i < 4; i++) {   // 32 q rows at a time
            const f32x16 lv = rowvals16(s_lse + 32 * i, hl), dv = rowvals16(s_del + 32 * i, hl);      // -lse1 log2(e), -delta1 scale
            f32x16 s = zero16(), dp = zero16();   // S[q row][landmark], dP[q row][landmark]
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                s = MFMA(frag_kc(s_q, 32 * i, 16 * ks, lane), klf[ks], s);
                dp = MFMA(frag_kc(s_g, 32 * i, 16 * ks, lane), w2f[ks], dp);
            }
            if (masked) {
                const f32x16 vr = rowvals16(s_mr + 32 * i, hl);
                s = s * g.scale2;
                mask_fill16(s, vr, ml);
                s = s + lv;
                exp2_16(s);
                dp = s * (dp * g.scale + dv);
                mask_zero16(dp, vr, ml);
            } else {
                s = s * g.scale2 + lv;
                exp2_16(s);
                dp = s * (dp * g.scale + dv);
            }
            const bf16x8 p0 = pack8<0>(s), p1 = pack8<1>(s), d0 = pack8<0>(dp), d1 = pack8<1>(dp);
#pragma unroll
            for (int nb = 0; nb < 2; nb++) {
                adw[nb] = MFMA(p0, frag_tr(s_g, 32 * nb, 32 * i, lane), adw[nb]);
                adw[nb] = MFMA(p1, frag_tr(s_g, 32 * nb, 32 * i + 16, lane), adw[nb]);
                adk[nb] = MFMA(d0, frag_tr(s_q, 32 * nb, 32 * i, lane), adk[nb]);
                adk[nb] = MFMA(d1, frag_tr(s_q, 32 * nb, 32 * i + 16, lane), adk[nb]);
            }
        }
    }
    float* dwb = dw2 + (long)bh * NM * ND;
    float* dkb = dlm + (long)b * NM * 2 * D + D + hd * ND;
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
        atomic_tile(dwb + (long)(32 * wave) * ND + 32 * nb, ND, adw[nb], hl, c);
        atomic_tile(dkb + (long)(32 * wave) * 2 * D + 32 * nb, 2 * D, adk[nb], hl, c);
    }
}

// ============================================================================ attn3 forward (L kernel, online softmax)
// grid (splits, B h).  av[bh, l, d] = sum_n softmax_n(scale q_l k^T)[l, n] v[n, d];  lse3[bh, l].
// One workgroup per (b, h) would leave half of the CUs idle (128 workgroups), so the sequence is cut into `splits`
// ranges of tiles: each workgroup leaves its unnormalised O, running max and sum in `part` and nys_a3_combine_kernel
// merges them (splits == 1: finished here, `part` unused).
constexpr int A3_PART = NM * (ND + 2);     // floats per (b, h, split): O [256][64], m [256], l [256]
// RC: [3P] `out += self.res_conv(v)` (called at models/mirror.py:312) computed HERE, from the v tile this kernel stages anyway: the
// 33-tap depthwise conv along the sequence is the banded Toeplitz product of resconv_mfma.hip (8 MFMAs per 32 x 64 output block against
// the tile's rows, read with the same transposing fragments as the P V product); the tile carries a 16-row halo on both sides and
// every wave writes the 32 rows of `rc_out` it owns.  Replaces a launch that re-read v (71 MB) beside the half-chip pinv chain.
template <bool MASKED, bool RC = false>
__global__ __launch_bounds__(NT, 2) void nys_a3_fwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                        float* __restrict__ av, float* __restrict__ lse3, float* __restrict__ part,
                                                        Geo g, int tiles_per_wg) {
    constexpr int VH = RC ? RC_HALO : 0;          // halo rows in front of (and behind) the v tile's image
    __shared__ __attribute__((aligned(16))) bf16_t s_k[TR * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_v_[(TR + 2 * VH) * NP];
    __shared__ __attribute__((aligned(16))) float s_mr[TR];
    bf16_t* const s_v = s_v_ + VH * NP;          // the tile proper: image row q <-> sequence row 128 t + q, q in [-VH, 128 + VH)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    const long LD = g.lm_ld;
    const int t0 = blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, g.n_p / TR);
    constexpr bool masked = MASKED;
    float ml[2] = {1.f, 1.f};
    if (masked) { ml[0] = g.mlm[(long)b * NM + 64 * wave + c]; ml[1] = g.mlm[(long)b * NM + 64 * wave + 32 + c]; }
    const bf16_t* qlb = lm + (long)b * NM * LD + hd * ND;
    bf16x8 qlf[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qlf[j][ks] = frag_g(qlb + (long)(64 * wave + 32 * j + c) * LD, 16 * ks, lane);
    const bf16_t* kb = qkv + (long)b * g.n_p * 3 * D + D + hd * ND;
    const bf16_t* vb = kb + D;
    float mrun[2] = {NEG_BIG, NEG_BIG}, lrun[2] = {0.f, 0.f};
    f32x16 o[2][2];   // O^T[d (nb)][landmark (j)]
#pragma unroll
    for (int nb = 0; nb < 2; nb++)
#pragma unroll
        for (int j = 0; j < 2; j++) o[nb][j] = zero16();
    u32x4 rk[TR * 8 / NT], rv[TR * 8 / NT];
    tile_load<TR>(rk, kb + (long)t0 * TR * 3 * D, 3 * D, tid);
    tile_load<TR>(rv, vb + (long)t0 * TR * 3 * D, 3 * D, tid);
    // RC: thread tid carries one 16-byte piece of the halo: rows 128 t - 16 + hq (hq < 16) in front, 128 t + 128 + (hq - 16) behind
    u32x4 rh = {0u, 0u, 0u, 0u};
    const int hq = tid >> 3, hc = tid & 7, hrow = hq < RC_HALO ? hq - RC_HALO : TR + hq - RC_HALO;
    auto halo_load = [&](int t) {
        const long sr = (long)t * TR + hrow;
        rh = (u32x4){0u, 0u, 0u, 0u};
        if (sr >= 0 && sr < g.n_p) rh = *reinterpret_cast<const u32x4*>(vb + sr * 3 * D + hc * 8);
    };
    // Toeplitz operand W^T[k][row c] = w[k - c], k in the accumulator order of k-step ks (resconv_mfma.hip): the same 4 fragments for
    // every tile and every wave, parked in LDS (16 registers that this 256-register kernel does not have)
    __shared__ __attribute__((aligned(16))) bf16x8 s_wf_[RC ? 4 * 64 : 1];
    if constexpr (RC) {
        halo_load(t0);
        if (wave == 0) {
            const float* wh = g.rc_w + hd * RC_TAPS;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                bf16x8 wv;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int k = 16 * ks + (e < 4 ? 4 * hl + e : 8 + 4 * hl + (e - 4));
                    const int jt = k - c;
                    const float x = wh[min(max(jt, 0), RC_TAPS - 1)];      // unconditional load + select: no branch per element
                    wv[e] = (__bf16)((jt >= 0 && jt < RC_TAPS) ? x : 0.f);
                }
                s_wf_[ks * 64 + lane] = wv;
            }
        }
    }
#pragma unroll 1
    for (int t = t0; t < t1; t++) {
        __syncthreads();
        tile_store<TR>(rk, s_k, tid);
        tile_store<TR>(rv, s_v, tid);
        if constexpr (RC) *reinterpret_cast<u32x4*>(s_v + hrow * NP + hc * 8) = rh;
        if (tid < TR) s_mr[tid] = masked ? g.mrow[(long)b * g.n_p + (long)t * TR + tid] : 1.f;
        __syncthreads();
        if (t + 1 < t1) {
            tile_load<TR>(rk, kb + (long)(t + 1) * TR * 3 * D, 3 * D, tid);
            tile_load<TR>(rv, vb + (long)(t + 1) * TR * 3 * D, 3 * D, tid);
            if constexpr (RC) halo_load(t + 1);
        }
        if constexpr (RC) {
            // rows 128 t + 32 wave + c of res_conv(v): out^T[d][row] = sum_k V^T[d][k] W^T[k][row], k = image rows 32 wave - 16 + {0..63}
            bf16_t* orow = g.rc_out + ((long)b * g.n_p + (long)t * TR + 32 * wave + c) * D + hd * ND;
            int opq = 0;
            asm volatile("" : "+v"(opq));          // per tile: no hoisting of the four fragment reads out of the loop
            const bf16x8* s_wf = s_wf_ + opq;
#pragma unroll
            for (int nb = 0; nb < 2; nb++) {
                f32x16 a = zero16();
#pragma unroll
                for (int ks = 0; ks < 4; ks++) a = MFMA(frag_tr(s_v, 32 * nb, 32 * wave - RC_HALO + 16 * ks, lane), s_wf[ks * 64 + lane], a);
                store_row8(orow + 32 * nb, a, hl);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {      // 64 keys at a time: 32 logits live instead of 64 (two waves per SIMD)
                f32x16 s[2];   // S3^T[key][landmark]
                float mx = NEG_BIG;
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    s[i] = zero16();
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) s[i] = MFMA(frag_kc(s_k, 64 * hf + 32 * i, 16 * ks, lane), qlf[j][ks], s[i]);
                    if (masked) {
                        s[i] = s[i] * g.scale2;
                        mask_fill16(s[i], rowvals16(s_mr + 64 * hf + 32 * i, hl), ml[j]);
                    }
                    mx = fmaxf(mx, vmax16(s[i]));
                }
                const float sc2 = masked ? 1.f : g.scale2;
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(mrun[j], mx * sc2);           // running max in log2 units
                if (__builtin_amdgcn_ballot_w64(mnew != mrun[j])) {    // rescale only when some landmark's max moved
                    const float alpha = __builtin_amdgcn_exp2f(mrun[j] - mnew);
                    lrun[j] *= alpha;
#pragma unroll
                    for (int nb = 0; nb < 2; nb++) o[nb][j] *= alpha;
                    mrun[j] = mnew;
                }
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    s[i] = fma_splat(s[i], sc2, -mrun[j]);
                    exp2_16(s[i]);
                }
                lrun[j] += vsum16(s[0] + s[1]);    // per lane half; the halves are joined once at the end
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const bf16x8 p0 = pack8<0>(s[i]), p1 = pack8<1>(s[i]);
#pragma unroll
                    for (int nb = 0; nb < 2; nb++) {
                        o[nb][j] = MFMA(frag_tr(s_v, 32 * nb, 64 * hf + 32 * i, lane), p0, o[nb][j]);
                        o[nb][j] = MFMA(frag_tr(s_v, 32 * nb, 64 * hf + 32 * i + 16, lane), p1, o[nb][j]);
                    }
                }
            }
    }
    const bool whole = gridDim.x == 1;
    float* pb = part + ((long)bh * gridDim.x + blockIdx.x) * A3_PART;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const float l = lrun[j] + __shfl_xor(lrun[j], 32, 64);
        const float inv = whole ? 1.f / l : 1.f;
        const int lq = 64 * wave + 32 * j + c;
        if (hl == 0) {
            if (whole) lse3[(long)bh * NM + lq] = (mrun[j] + __log2f(l)) * LN2;
            else { pb[NM * ND + lq] = mrun[j]; pb[NM * ND + NM + lq] = l; }
        }
        float* arow = whole ? av + ((long)bh * NM + lq) * ND : pb + (long)lq * ND;
#pragma unroll
        for (int nb = 0; nb < 2; nb++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; e++) w[e] = o[nb][j][4 * gq + e] * inv;
                *reinterpret_cast<f32x4*>(arow + 32 * nb + 8 * gq + 4 * hl) = w;
            }
    }
}

// grid (4, B h): thread = (landmark, 16-wide slice of d).  av = sum_s O_s 2^(m_s - M) / sum_s l_s 2^(m_s - M)
__global__ __launch_bounds__(256) void nys_a3_combine_kernel(const float* __restrict__ part, float* __restrict__ av,
                                                             float* __restrict__ lse3, int splits) {
    const int bh = blockIdx.y, l = threadIdx.x, d0 = 16 * blockIdx.x;
    const float* pb = part + (long)bh * splits * A3_PART;
    float M = NEG_BIG;
    for (int sp = 0; sp < splits; sp++) M = fmaxf(M, pb[(long)sp * A3_PART + NM * ND + l]);
    float L = 0.f;
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int sp = 0; sp < splits; sp++) {
        const float* ps = pb + (long)sp * A3_PART;
        const float w = __builtin_amdgcn_exp2f(ps[NM * ND + l] - M);
        L += ps[NM * ND + NM + l] * w;
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] += *reinterpret_cast<const f32x4*>(ps + (long)l * ND + d0 + 4 * q) * w;
    }
    const float inv = 1.f / L;
    if (blockIdx.x == 0) lse3[(long)bh * NM + l] = (M + __log2f(L)) * LN2;
#pragma unroll
    for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4*>(av + ((long)bh * NM + l) * ND + d0 + 4 * q) = acc[q] * inv;
}

// delta3[bh, l] = sum_d dav[l, d] av[l, d]: once per (b, h) instead of once per 128-row tile (34x the reads)
__global__ __launch_bounds__(256) void nys_delta3_kernel(const float* __restrict__ av, const bf16_t* __restrict__ dav,
                                                         float* __restrict__ delta3) {
    const long row = (long)blockIdx.x * NM + threadIdx.x;
    const float* ar = av + row * ND;
    const bf16_t* gr = dav + row * ND;
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < ND; e += 8) {
        const u32x4 gv = *reinterpret_cast<const u32x4*>(gr + e);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ar + e), a1 = *reinterpret_cast<const f32x4*>(ar + e + 4);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float lo = __uint_as_float(gv[w] << 16), hi = __uint_as_float(gv[w] & 0xffff0000u);
            const float x0 = w < 2 ? a0[2 * w] : a1[2 * w - 4], x1 = w < 2 ? a0[2 * w + 1] : a1[2 * w - 3];
            d += lo * x0 + hi * x1;
        }
    }
    delta3[row] = d;
}

// ============================================================================ attn3 backward, dk + dv (N kernel)
// grid (splits, B h), same walk as attn1.  P3 = exp(scale q_l k^T - lse3), dv = P3^T dav,
// dS3 = P3 o (dav v^T - delta3) scale, dk = dS3^T q_l
template <bool MASKED>
__global__ __launch_bounds__(NT, 2) void nys_a3_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                            const float* __restrict__ delta3, const bf16_t* __restrict__ dav,
                                                            const float* __restrict__ lse3, bf16_t* __restrict__ dqkv, Geo g) {
    __shared__ __attribute__((aligned(16))) bf16_t s_ql_[NM * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_g_[NM * NP];
    __shared__ __attribute__((aligned(16))) float s_lse_[NM];
    __shared__ __attribute__((aligned(16))) float s_del_[NM];
    __shared__ __attribute__((aligned(16))) float s_mlm_[NM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    const long LD = g.lm_ld;
    stage_rows<NM>(s_ql_, lm + (long)b * NM * LD + hd * ND, LD, tid);
    stage_rows<NM>(s_g_, dav + (long)bh * NM * ND, ND, tid);
    s_del_[tid] = -delta3[(long)bh * NM + tid] * g.scale;      // thread = landmark; staged negated (a subtraction of per-element values
    s_lse_[tid] = -lse3[(long)bh * NM + tid] * LOG2E;          // costs a sign flip per element on top of the multiply-add)
    constexpr bool masked = MASKED;
    s_mlm_[tid] = masked ? g.mlm[(long)b * NM + tid] : 1.f;
    const int nblk = g.n_p / 32, stride = 4 * gridDim.x;
    int rb = 4 * blockIdx.x + wave;
    const bf16_t* kb = qkv + (long)b * g.n_p * 3 * D + D + hd * ND;
    bf16x8 kn[4], vn[4];
    if (rb < nblk) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            kn[ks] = frag_g(kb + (long)(32 * rb + c) * 3 * D, 16 * ks, lane);
            vn[ks] = frag_g(kb + (long)(32 * rb + c) * 3 * D + D, 16 * ks, lane);
        }
    }
    __syncthreads();
#pragma unroll 1
    for (; rb < nblk; rb += stride) {
        const long row = 32L * rb + c;
        int opq = 0;                      // see nys_a1_fwd_kernel
        asm volatile("" : "+v"(opq));
        const bf16_t* s_ql = s_ql_ + opq;
        const bf16_t* s_g = s_g_ + opq;
        const float* s_lse = s_lse_ + opq;
        const float* s_del = s_del_ + opq;
        const float* s_mlm = s_mlm_ + opq;
        const float mr = masked ? g.mrow[(long)b * g.n_p + row] : 1.f;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
        if (rb + stride < nblk) {
            const bf16_t* nr = kb + (32L * (rb + stride) + c) * 3 * D;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                kn[ks] = frag_g(nr, 16 * ks, lane);
                vn[ks] = frag_g(nr + D, 16 * ks, lane);
            }
        }
        f32x16 adv[2] = {zero16(), zero16()}, adk[2] = {zero16(), zero16()};   // dv^T[d][key], dk^T[d][key]
#pragma unroll
        for (int blk = 0; blk < 8; blk++) {
            f32x16 s = zero16(), dp = zero16();   // S3[landmark][key], dP3[landmark][key]
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                s = MFMA(frag_kc(s_ql, 32 * blk, 16 * ks, lane), kf[ks], s);
                dp = MFMA(frag_kc(s_g, 32 * blk, 16 * ks, lane), vf[ks], dp);
            }
            if (masked) {
                const f32x16 vr = rowvals16(s_mlm + 32 * blk, hl);
                s = s * g.scale2;
                mask_fill16(s, vr, mr);
                s = s + rowvals16(s_lse + 32 * blk, hl);
                exp2_16(s);
                dp = s * (dp * g.scale + rowvals16(s_del + 32 * blk, hl));
                mask_zero16(dp, vr, mr);
            } else {
                s = s * g.scale2 + rowvals16(s_lse + 32 * blk, hl);      // staged as -lse3 * log2(e)
                exp2_16(s);
                dp = s * (dp * g.scale + rowvals16(s_del + 32 * blk, hl));   // staged as -delta3 * scale
            }
            const bf16x8 p0 = pack8<0>(s), p1 = pack8<1>(s), d0 = pack8<0>(dp), d1 = pack8<1>(dp);
#pragma unroll
            for (int nb = 0; nb < 2; nb++) {
                adv[nb] = MFMA(frag_tr(s_g, 32 * nb, 32 * blk, lane), p0, adv[nb]);
                adv[nb] = MFMA(frag_tr(s_g, 32 * nb, 32 * blk + 16, lane), p1, adv[nb]);
                adk[nb] = MFMA(frag_tr(s_ql, 32 * nb, 32 * blk, lane), d0, adk[nb]);
                adk[nb] = MFMA(frag_tr(s_ql, 32 * nb, 32 * blk + 16, lane), d1, adk[nb]);
            }
        }
        bf16_t* dkrow = dqkv + ((long)b * g.n_p + row) * 3 * D + D + hd * ND;
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
            store_row8(dkrow + 32 * nb, adk[nb], hl);
            store_row8(dkrow + D + 32 * nb, adv[nb], hl);
        }
    }
}

// ============================================================================ attn3 backward, dq_l (L kernel)
// grid (splits, B h).  dq_l[l, d] += sum_n dS3[l, n] k[n, d]   (f32 atomics into the q_l half of dlm)
template <bool MASKED>
__global__ __launch_bounds__(NT) void nys_a3_bwd_dql_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                            const float* __restrict__ delta3, const bf16_t* __restrict__ dav,
                                                            const float* __restrict__ lse3, float* __restrict__ dlm, Geo g,
                                                            int tiles_per_wg) {
    __shared__ __attribute__((aligned(16))) bf16_t s_k[TR * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_v[TR * NP];
    __shared__ __attribute__((aligned(16))) float s_mr[TR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    const long LD = g.lm_ld;
    const int ntiles = g.n_p / TR;
    const int t0 = blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, ntiles);
    if (t0 >= t1) return;
    constexpr bool masked = MASKED;
    float ml[2] = {1.f, 1.f};
    if (masked) { ml[0] = g.mlm[(long)b * NM + 64 * wave + c]; ml[1] = g.mlm[(long)b * NM + 64 * wave + 32 + c]; }
    const bf16_t* qlb = lm + (long)b * NM * LD + hd * ND;
    bf16x8 qlf[2][4], gf[2][4];
    float lsev[2], delv[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int lq = 64 * wave + 32 * j + c;
        const bf16_t* gr = dav + ((long)bh * NM + lq) * ND;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            qlf[j][ks] = frag_g(qlb + (long)lq * LD, 16 * ks, lane);
            gf[j][ks] = frag_g(gr, 16 * ks, lane);
        }
        delv[j] = delta3[(long)bh * NM + lq] * g.scale;
        lsev[j] = lse3[(long)bh * NM + lq] * LOG2E;
    }
    const bf16_t* kb = qkv + (long)b * g.n_p * 3 * D + D + hd * ND;
    const bf16_t* vb = kb + D;
    f32x16 acc[2][2];   // dq_l[landmark (j block, registers)][d (nb block, lanes)]
#pragma unroll
    for (int nb = 0; nb < 2; nb++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[nb][j] = zero16();
    u32x4 rk[TR * 8 / NT], rv[TR * 8 / NT];
    tile_load<TR>(rk, kb + (long)t0 * TR * 3 * D, 3 * D, tid);
    tile_load<TR>(rv, vb + (long)t0 * TR * 3 * D, 3 * D, tid);
#pragma unroll 1
    for (int t = t0; t < t1; t++) {
        __syncthreads();
        tile_store<TR>(rk, s_k, tid);
        tile_store<TR>(rv, s_v, tid);
        if (tid < TR) s_mr[tid] = masked ? g.mrow[(long)b * g.n_p + (long)t * TR + tid] : 1.f;
        __syncthreads();
        if (t + 1 < t1) {
            tile_load<TR>(rk, kb + (long)(t + 1) * TR * 3 * D, 3 * D, tid);
            tile_load<TR>(rv, vb + (long)(t + 1) * TR * 3 * D, 3 * D, tid);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                f32x16 s = zero16(), dp = zero16();   // S3^T[key][landmark], dP3^T[key][landmark]
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    s = MFMA(frag_kc(s_k, 32 * i, 16 * ks, lane), qlf[j][ks], s);
                    dp = MFMA(frag_kc(s_v, 32 * i, 16 * ks, lane), gf[j][ks], dp);
                }
                if (masked) {
                    const f32x16 vr = rowvals16(s_mr + 32 * i, hl);
                    s = s * g.scale2;
                    mask_fill16(s, vr, ml[j]);
                    s = fma_splat(s, 1.f, -lsev[j]);
                    exp2_16(s);
                    dp = s * fma_splat(dp, g.scale, -delv[j]);
                    mask_zero16(dp, vr, ml[j]);
                } else {
                    s = fma_splat(s, g.scale2, -lsev[j]);
                    exp2_16(s);
                    dp = s * fma_splat(dp, g.scale, -delv[j]);
                }
                // dS3^T in the accumulator layout IS dS3 as an A operand (row = landmark = lane, k = keys): the product
                // comes out as dq_l[landmark (registers)][d (lanes)], so the final atomics are 128-byte coalesced
                const bf16x8 d0 = pack8<0>(dp), d1 = pack8<1>(dp);
#pragma unroll
                for (int nb = 0; nb < 2; nb++) {
                    acc[nb][j] = MFMA(d0, frag_tr(s_k, 32 * nb, 32 * i, lane), acc[nb][j]);
                    acc[nb][j] = MFMA(d1, frag_tr(s_k, 32 * nb, 32 * i + 16, lane), acc[nb][j]);
                }
            }
    }
    float* dqb = dlm + (long)b * NM * 2 * D + hd * ND;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int nb = 0; nb < 2; nb++)
            atomic_tile(dqb + (long)(64 * wave + 32 * j) * 2 * D + 32 * nb, 2 * D, acc[nb][j], hl, c);
}

// ============================================================================ attn3 backward in ONE pass (L kernel + LDS hand-over)
// dk, dv (contraction over the landmarks) and dq_l (contraction over the keys) need the probabilities in two orientations; as two
// kernels every logit, every dP and every exponential was computed twice (28 MFMAs per 32 x 32 tile instead of 20) and k / v were read
// twice.  Here wave w owns landmarks [64 w, 64 w + 64) as in the dq_l kernel: S^T and dP^T [key][landmark] come out with the landmark
// on the lane, dS^T is at once the A operand of dq_l += dS k (registers), and P / dS ALSO go to two LDS images [landmark][key] (the
// lane's own row, four consecutive keys per 8-byte store).  Behind a barrier the same four waves switch roles: wave w takes 32 of the
// 64 keys and 32 of the 64 channels and contracts over ALL 256 landmarks,  dv^T = dav^T P,  dk^T = q_l^T dS,  both operands read with
// the transposing fragments (contraction index = image row).  One workgroup per CU (157 KB of LDS), 64 keys per step, 2 barriers per step;
// the v rows are A fragments straight from HBM / L2 (the image budget has no room for a v tile).
// grid (splits, B h): range of 64-key steps per workgroup; dq_l leaves as f32 atomics into the q_l half of dlm (as nys_a3_bwd_dql_kernel).
constexpr int HT = 64;   // keys per step
constexpr int NT8 = 512;  // 8 waves: two per SIMD, so one wave's softmax arithmetic / LDS traffic runs under the other's MFMAs
// (first version: 4 waves of 64 landmarks, one per SIMD — 183 us alone against 192 for the two kernels, and 0.75 % SLOWER in the step)
template <bool MASKED>
__global__ __launch_bounds__(NT8) void nys_a3_bwd_one_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ lm,
                                                             const float* __restrict__ delta3, const bf16_t* __restrict__ dav,
                                                             const float* __restrict__ lse3, bf16_t* __restrict__ dqkv,
                                                             float* __restrict__ dlm, Geo g, int steps_per_wg) {
    __shared__ __attribute__((aligned(16))) bf16_t s_ql[NM * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_g[NM * NP];
    __shared__ __attribute__((aligned(16))) bf16_t s_p[NM * NP];       // P  [landmark][key 0..63]
    __shared__ __attribute__((aligned(16))) bf16_t s_ds[NM * NP];      // dS [landmark][key 0..63]
    __shared__ __attribute__((aligned(16))) bf16_t s_k[HT * NP];
    __shared__ __attribute__((aligned(16))) float s_mr[HT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hl = lane >> 5;
    const int bh = blockIdx.y, b = bh / g.h, hd = bh % g.h, D = g.D;
    const long LD = g.lm_ld;
    const int nsteps = g.n_p / HT;
    const int h0 = blockIdx.x * steps_per_wg, h1 = min(h0 + steps_per_wg, nsteps);
    if (h0 >= h1) return;
    constexpr bool masked = MASKED;
    const int lq = 32 * wave + c;                   // this lane's landmark (phase 1)
    const float ml = masked ? g.mlm[(long)b * NM + lq] : 1.f;
    const bf16_t* qlb = lm + (long)b * NM * LD + hd * ND;
    const bf16_t* gvb = dav + (long)bh * NM * ND;
#pragma unroll
    for (int i = 0; i < NM * 8 / NT8; i++) {
        const int cid = tid + i * NT8, r = cid >> 3, cc = cid & 7;
        *reinterpret_cast<u32x4*>(s_ql + r * NP + cc * 8) = *reinterpret_cast<const u32x4*>(qlb + (long)r * LD + cc * 8);
        *reinterpret_cast<u32x4*>(s_g + r * NP + cc * 8) = *reinterpret_cast<const u32x4*>(gvb + (long)r * ND + cc * 8);
    }
    bf16x8 qlf[4], gf[4];      // B fragments of this wave's 32 landmarks (phase 1)
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        qlf[ks] = frag_g(qlb + (long)lq * LD, 16 * ks, lane);
        gf[ks] = frag_g(gvb + (long)lq * ND, 16 * ks, lane);
    }
    const float delv = delta3[(long)bh * NM + lq] * g.scale, lsev = lse3[(long)bh * NM + lq] * LOG2E;
    const bf16_t* kb = qkv + (long)b * g.n_p * 3 * D + D + hd * ND;
    const bf16_t* vb = kb + D;
    f32x16 accq[2] = {zero16(), zero16()};   // dq_l[landmark (registers)][d (nb block, lanes)]
    // k tile: staged through registers one step ahead (one 16-byte piece per thread); v: A fragments (rows = keys) from global
    const int kr = tid >> 3, kc = tid & 7;
    u32x4 rk = *reinterpret_cast<const u32x4*>(kb + ((long)h0 * HT + kr) * 3 * D + kc * 8);
    bf16x8 vn[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) vn[i][ks] = frag_g(vb + ((long)h0 * HT + 32 * i + c) * 3 * D, 16 * ks, lane);
    *reinterpret_cast<u32x4*>(s_k + kr * NP + kc * 8) = rk;
    if (tid < HT) s_mr[tid] = masked ? g.mrow[(long)b * g.n_p + (long)h0 * HT + tid] : 1.f;
    __syncthreads();
    const int kb2 = wave & 1, dhalf = (wave >> 1) & 1, isk = wave >> 2;      // phase 2 roles: keys 32 kb2 .., channels 32 dhalf .., dv (0) / dk (1)
    const bf16_t* a_img = isk ? s_ql : s_g;
    const bf16_t* b_img = isk ? s_ds : s_p;
#pragma unroll 1
    for (int ht = h0; ht < h1; ht++) {
        bf16x8 vf[2][4];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) vf[i][ks] = vn[i][ks];
        const bool more = ht + 1 < h1;
        if (more) {
            rk = *reinterpret_cast<const u32x4*>(kb + ((long)(ht + 1) * HT + kr) * 3 * D + kc * 8);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int ks = 0; ks < 4; ks++) vn[i][ks] = frag_g(vb + ((long)(ht + 1) * HT + 32 * i + c) * 3 * D, 16 * ks, lane);
        }
        // ---- phase 1: this wave's 32 landmarks against the step's 64 keys
#pragma unroll
        for (int i = 0; i < 2; i++) {
            f32x16 sb = zero16(), dp = zero16();   // S3^T[key][landmark], dP3^T[key][landmark]
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                sb = MFMA(frag_kc(s_k, 32 * i, 16 * ks, lane), qlf[ks], sb);
                dp = MFMA(vf[i][ks], gf[ks], dp);
            }
            if (masked) {
                const f32x16 vr = rowvals16(s_mr + 32 * i, hl);
                sb = sb * g.scale2;
                mask_fill16(sb, vr, ml);
                sb = fma_splat(sb, 1.f, -lsev);
                exp2_16(sb);
                dp = sb * fma_splat(dp, g.scale, -delv);
                mask_zero16(dp, vr, ml);
            } else {
                sb = fma_splat(sb, g.scale2, -lsev);
                exp2_16(sb);
                dp = sb * fma_splat(dp, g.scale, -delv);
            }
            const bf16x8 p0 = pack8<0>(sb), p1 = pack8<1>(sb), d0 = pack8<0>(dp), d1 = pack8<1>(dp);
            // dS^T in the accumulator layout IS dS as an A operand (row = landmark = lane, k = keys)
#pragma unroll
            for (int nb = 0; nb < 2; nb++) {
                accq[nb] = MFMA(d0, frag_tr(s_k, 32 * nb, 32 * i, lane), accq[nb]);
                accq[nb] = MFMA(d1, frag_tr(s_k, 32 * nb, 32 * i + 16, lane), accq[nb]);
            }
            // the lane's own image row (its landmark), keys 32 i + 8 gq + 4 hl + {0..3}: packed element 4 gq + e of p0 | p1 is
            // accumulator register 4 gq + e = key 8 gq + 4 hl + e
            bf16_t* prow = s_p + lq * NP + 32 * i + 4 * hl;
            bf16_t* drow = s_ds + lq * NP + 32 * i + 4 * hl;
            const u32x4 pw0 = __builtin_bit_cast(u32x4, p0), pw1 = __builtin_bit_cast(u32x4, p1);
            const u32x4 dw0 = __builtin_bit_cast(u32x4, d0), dw1 = __builtin_bit_cast(u32x4, d1);
            *reinterpret_cast<u32x2*>(prow) = u32x2{pw0[0], pw0[1]};
            *reinterpret_cast<u32x2*>(prow + 8) = u32x2{pw0[2], pw0[3]};
            *reinterpret_cast<u32x2*>(prow + 16) = u32x2{pw1[0], pw1[1]};
            *reinterpret_cast<u32x2*>(prow + 24) = u32x2{pw1[2], pw1[3]};
            *reinterpret_cast<u32x2*>(drow) = u32x2{dw0[0], dw0[1]};
            *reinterpret_cast<u32x2*>(drow + 8) = u32x2{dw0[2], dw0[3]};
            *reinterpret_cast<u32x2*>(drow + 16) = u32x2{dw1[0], dw1[1]};
            *reinterpret_cast<u32x2*>(drow + 24) = u32x2{dw1[2], dw1[3]};
        }
        __syncthreads();      // the images are complete; nobody reads s_k / s_mr any more
        // ---- phase 2: one of dv^T / dk^T, keys 32 kb2 .., channels 32 dhalf .., over all 256 landmarks; the next k tile moves in beside it
        if (more) {
            *reinterpret_cast<u32x4*>(s_k + kr * NP + kc * 8) = rk;
            if (tid < HT) s_mr[tid] = masked ? g.mrow[(long)b * g.n_p + (long)(ht + 1) * HT + tid] : 1.f;
        }
        f32x16 a2 = zero16();      // dv^T or dk^T [d][key]
#pragma unroll
        for (int t = 0; t < NM / 16; t++) a2 = MFMA(frag_tr(a_img, 32 * dhalf, 16 * t, lane), frag_tr(b_img, 32 * kb2, 16 * t, lane), a2);
        store_row8(dqkv + ((long)b * g.n_p + (long)ht * HT + 32 * kb2 + c) * 3 * D + (isk ? D : 2 * D) + hd * ND + 32 * dhalf, a2, hl);
        __syncthreads();      // the images may be overwritten; the next k tile is in place
    }
    float* dqb = dlm + (long)b * NM * 2 * D + hd * ND;
#pragma unroll
    for (int nb = 0; nb < 2; nb++) atomic_tile(dqb + (long)(32 * wave) * 2 * D + 32 * nb, 2 * D, accq[nb], hl, c);
}

// workgroups per (b, h) of the sequence-walking N kernels: enough to put two workgroups on every CU (the landmark images
// take 72 KB of LDS each), never more than one 32-row block per wave
int pick_walkers(int BH, int n_p) {
    int w = 1;
    while (BH * w < 512 && 4 * w * 2 <= n_p / 32) w *= 2;
    return w;
}

// the mask-aware instantiation only when a mask is given: the unmasked kernels keep their register budget
#define NYS_LAUNCH(kern, grid, block, shm, stream, ...)                                              \
    do {                                                                                             \
        if (g.mrow) hipLaunchKernelGGL((kern<true>), grid, block, shm, stream, __VA_ARGS__);        \
        else hipLaunchKernelGGL((kern<false>), grid, block, shm, stream, __VA_ARGS__);              \
    } while (0)

int check_geo(const char* fn, int B, int h, int n_p, int m, int dh) {
    MH_REQUIRE(m == NM && dh == ND, "%s: built for m = %d landmarks and dh = %d (got m=%d dh=%d); other shapes use mh_gemm + mh_softmax",
               fn, NM, ND, m, dh);
    MH_REQUIRE(B >= 0 && h >= 1 && n_p >= NM && n_p % NM == 0, "%s: n_p=%d must be a positive multiple of %d", fn, n_p, NM);
    return MH_OK;
}

}  // namespace

extern "C" int mh_nys_attn1_fwd(const void* qkv, const void* lm, const void* w2, void* out, float* lse1, const float* mrow,
                                const float* mlm, int B, int h, int n_p, int m, int dh, float scale, int accumulate, int64_t lm_ld, void* o1,
                                mh_stream s) {
    if (int e = check_geo("mh_nys_attn1_fwd", B, h, n_p, m, dh)) return e;
    MH_REQUIRE(lm_ld == 0 || (lm_ld >= 2L * h * ND && lm_ld % 8 == 0), "mh_nys_attn1_fwd: lm_ld must be 0 or a multiple of 8 >= 2 D");
    if (B == 0) return MH_OK;
    MH_REQUIRE((mrow == nullptr) == (mlm == nullptr), "mh_nys_attn1_fwd: mrow and mlm go together");
    const Geo g{h, n_p, h * ND, scale, scale * LOG2E, mrow, mlm, accumulate, lm_ld > 0 ? lm_ld : 2L * h * ND, nullptr, nullptr};
    MH_REQUIRE((((uintptr_t)o1) & 15) == 0, "mh_nys_attn1_fwd: o1 must be 16-byte aligned");
    NYS_LAUNCH(nys_a1_fwd_kernel, dim3(pick_walkers(B * h, n_p), B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv, (const bf16_t*)lm,
                       (const bf16_t*)w2, (bf16_t*)out, lse1, g, Q8Out{}, (bf16_t*)o1);
    MH_LAUNCH_CHECK("mh_nys_attn1_fwd");
    return MH_OK;
}

extern "C" int mh_nys_attn1_fwd_q8(const void* qkv, const void* lm, const void* w2, void* out, float* lse1, int B, int h, int n_p, int m,
                                   int dh, float scale, int accumulate, void* q8, unsigned* ring, const float* tick, float margin,
                                   float* q8_scale, void* o1, mh_stream s) {
    if (int e = check_geo("mh_nys_attn1_fwd_q8", B, h, n_p, m, dh)) return e;
    if (B == 0) return MH_OK;
    MH_REQUIRE(q8 && ring && tick && q8_scale && margin >= 1.f && ((uintptr_t)q8 & 3) == 0, "mh_nys_attn1_fwd_q8: q8, ring, tick, scale and margin >= 1");
    const Geo g{h, n_p, h * ND, scale, scale * LOG2E, nullptr, nullptr, accumulate, 2L * h * ND, nullptr, nullptr};
    const Q8Out o{(unsigned char*)q8, ring, tick, margin, q8_scale};
    hipLaunchKernelGGL((nys_a1_fwd_kernel<false, true>), dim3(pick_walkers(B * h, n_p), B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv,
                       (const bf16_t*)lm, (const bf16_t*)w2, (bf16_t*)out, lse1, g, o, (bf16_t*)o1);
    MH_LAUNCH_CHECK("mh_nys_attn1_fwd_q8");
    return MH_OK;
}

int pick_splits(int BH, int ntiles, int which);

extern "C" int64_t mh_nys_attn3_ws_floats(int B, int h, int n_p) {
    if (B <= 0 || h <= 0 || n_p < TR) return 0;
    const int splits = pick_splits(B * h, n_p / TR, 0);
    return splits > 1 ? (int64_t)B * h * splits * A3_PART : 0;
}

extern "C" int64_t mh_nys_attn3_workspace_bytes(int B, int h, int n_p) { return 4 * mh_nys_attn3_ws_floats(B, h, n_p); }

extern "C" int mh_nys_attn3_fwd(const void* qkv, const void* lm, float* av, float* lse3, float* workspace, int64_t ws_floats,
                                const float* mrow, const float* mlm, int B, int h, int n_p, int m, int dh, float scale, int64_t lm_ld,
                                const float* rc_w, void* rc_out, mh_stream s) {
    if (int e = check_geo("mh_nys_attn3_fwd", B, h, n_p, m, dh)) return e;
    MH_REQUIRE(lm_ld == 0 || (lm_ld >= 2L * h * ND && lm_ld % 8 == 0), "mh_nys_attn3_fwd: lm_ld must be 0 or a multiple of 8 >= 2 D");
    MH_REQUIRE((rc_w == nullptr) == (rc_out == nullptr) && (((uintptr_t)rc_out) & 15) == 0, "mh_nys_attn3_fwd: rc_w and rc_out (16-byte aligned) go together");
    if (B == 0) return MH_OK;
    const Geo g{h, n_p, h * ND, scale, scale * LOG2E, mrow, mlm, 0, lm_ld > 0 ? lm_ld : 2L * h * ND, rc_w, (bf16_t*)rc_out};
    const int ntiles = n_p / TR;
    int splits = pick_splits(B * h, ntiles, 0);
    if (!workspace || ws_floats < (int64_t)B * h * splits * A3_PART) splits = 1;      // no room for partials: one workgroup per (b, h)
    const int tpw = (ntiles + splits - 1) / splits;
    splits = (ntiles + tpw - 1) / tpw;                                                 // no empty ranges
    if (rc_w) {
        if (g.mrow) hipLaunchKernelGGL((nys_a3_fwd_kernel<true, true>), dim3(splits, B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv,
                                       (const bf16_t*)lm, av, lse3, workspace, g, tpw);
        else hipLaunchKernelGGL((nys_a3_fwd_kernel<false, true>), dim3(splits, B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv,
                                (const bf16_t*)lm, av, lse3, workspace, g, tpw);
    } else {
        NYS_LAUNCH(nys_a3_fwd_kernel, dim3(splits, B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv, (const bf16_t*)lm, av,
                           lse3, workspace, g, tpw);
    }
    MH_LAUNCH_CHECK("mh_nys_attn3_fwd");
    if (splits > 1) {
        hipLaunchKernelGGL(nys_a3_combine_kernel, dim3(ND / 16, B * h), dim3(NM), 0, (hipStream_t)s, (const float*)workspace, av, lse3,
                           splits);
        MH_LAUNCH_CHECK("mh_nys_attn3_fwd(combine)");
    }
    return MH_OK;
}

// splits: workgroups per (b, h) for the landmark-owner kernels (partials in a workspace, or f32 atomics)
// which: 0 = attn3 forward (+ combine pass), 1 = attn1 backward dw2 / dk_l, 2 = attn3 backward dq_l (f32 atomics).
// One workgroup per CU in all (B h x splits ~ 256): measured in the step at B h = 128, 2 ranges per (b, h) beat 4 by 1.7 %
// (half the partial tiles / atomics, and these launches run beside the half-chip chain) and 1 by 1.2 %.
int pick_splits(int BH, int ntiles, int which) {
    int splits = 1;
    while (BH * splits < 256 && splits * 2 <= ntiles) splits *= 2;
    return splits;
}

extern "C" int mh_nys_attn1_bwd(const void* qkv, const void* lm, const void* w2, const void* dout, const float* lse1, const void* o1,
                                float* delta1, void* dqkv, float* dw2, float* dlm, const float* mrow, const float* mlm, int B, int h,
                                int n_p, int m, int dh, float scale, int64_t lm_ld, int which, mh_stream s) {
    if (int e = check_geo("mh_nys_attn1_bwd", B, h, n_p, m, dh)) return e;
    MH_REQUIRE(lm_ld == 0 || (lm_ld >= 2L * h * ND && lm_ld % 8 == 0), "mh_nys_attn1_bwd: lm_ld must be 0 or a multiple of 8 >= 2 D");
    MH_REQUIRE(which >= 1 && which <= 3, "mh_nys_attn1_bwd: which = 1 (dw2, dk_l, delta1), 2 (dq from delta1) or 3 (both, in that order)");
    MH_REQUIRE(delta1 && (!(which & 1) || (o1 && dw2 && dlm && (((uintptr_t)o1) & 15) == 0)) && (!(which & 2) || dqkv),
               "mh_nys_attn1_bwd: missing buffer for the requested part");
    if (B == 0) return MH_OK;
    const Geo g{h, n_p, h * ND, scale, scale * LOG2E, mrow, mlm, 0, lm_ld > 0 ? lm_ld : 2L * h * ND, nullptr, nullptr};
    if (which & 1) {
        const int ntiles = n_p / TR, splits = pick_splits(B * h, ntiles, 1), tpw = (ntiles + splits - 1) / splits;
        NYS_LAUNCH(nys_a1_bwd_dw_kernel, dim3(splits, B * h), dim3(NTW), 0, (hipStream_t)s, (const bf16_t*)qkv,
                           (const bf16_t*)lm, (const bf16_t*)w2, (const bf16_t*)dout, lse1, (const bf16_t*)o1, delta1, dw2, dlm, g, tpw);
        MH_LAUNCH_CHECK("mh_nys_attn1_bwd(dw)");
    }
    if (which & 2) {
        NYS_LAUNCH(nys_a1_bwd_dq_kernel, dim3(pick_walkers(B * h, n_p), B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv,
                           (const bf16_t*)lm, (const bf16_t*)w2, (const bf16_t*)dout, lse1, (const float*)delta1, (bf16_t*)dqkv, g);
        MH_LAUNCH_CHECK("mh_nys_attn1_bwd(dq)");
    }
    return MH_OK;
}

extern "C" int mh_nys_attn3_bwd(const void* qkv, const void* lm, const float* av, const void* dav, const float* lse3, float* delta3,
                                void* dqkv, float* dlm, const float* mrow, const float* mlm, int B, int h, int n_p, int m, int dh,
                                float scale, int64_t lm_ld, int one_pass, mh_stream s) {
    if (int e = check_geo("mh_nys_attn3_bwd", B, h, n_p, m, dh)) return e;
    MH_REQUIRE(lm_ld == 0 || (lm_ld >= 2L * h * ND && lm_ld % 8 == 0), "mh_nys_attn3_bwd: lm_ld must be 0 or a multiple of 8 >= 2 D");
    if (B == 0) return MH_OK;
    const Geo g{h, n_p, h * ND, scale, scale * LOG2E, mrow, mlm, 0, lm_ld > 0 ? lm_ld : 2L * h * ND, nullptr, nullptr};
    if (av) hipLaunchKernelGGL(nys_delta3_kernel, dim3(B * h), dim3(NM), 0, (hipStream_t)s, av, (const bf16_t*)dav, delta3);
    if (one_pass) {
        const int nsteps = n_p / HT;
        int splits = 1;
        while (B * h * splits < 256 && splits * 2 <= nsteps) splits *= 2;
        const int spw = (nsteps + splits - 1) / splits;
        NYS_LAUNCH(nys_a3_bwd_one_kernel, dim3((nsteps + spw - 1) / spw, B * h), dim3(NT8), 0, (hipStream_t)s, (const bf16_t*)qkv,
                           (const bf16_t*)lm, (const float*)delta3, (const bf16_t*)dav, lse3, (bf16_t*)dqkv, dlm, g, spw);
        MH_LAUNCH_CHECK("mh_nys_attn3_bwd(one pass)");
        return MH_OK;
    }
    NYS_LAUNCH(nys_a3_bwd_dkv_kernel, dim3(pick_walkers(B * h, n_p), B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv,
                       (const bf16_t*)lm, (const float*)delta3, (const bf16_t*)dav, lse3, (bf16_t*)dqkv, g);
    MH_LAUNCH_CHECK("mh_nys_attn3_bwd(dkv)");
    const int ntiles = n_p / TR, splits = pick_splits(B * h, ntiles, 2), tpw = (ntiles + splits - 1) / splits;
    NYS_LAUNCH(nys_a3_bwd_dql_kernel, dim3(splits, B * h), dim3(NT), 0, (hipStream_t)s, (const bf16_t*)qkv,
                       (const bf16_t*)lm, (const float*)delta3, (const bf16_t*)dav, lse3, dlm, g, tpw);
    MH_LAUNCH_CHECK("mh_nys_attn3_bwd(dql)");
    return MH_OK;
}
