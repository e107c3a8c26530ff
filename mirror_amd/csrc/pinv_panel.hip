// Moore-Penrose iteration ([3P] moore_penrose_iter_pinv, called at models/mirror.py:312) as ONE launch per pass,
// m = 256, bf16 MFMA with f32 accumulation — "column panel" formulation.
//
//   z <- 1/4 z (13I - P (15I - P (7I - P))),  P = X z        24 (forward) / 48 (backward) dependent 256^3 products
//
// One 256-thread workgroup (4 waves, one per SIMD, the whole 512-register budget each) owns one (batch, head).  Wave w
// owns COLUMNS [64w, 64w+64) of every product C = A B: it needs all of A and only its own 64 columns of B.
//   * A lives in LDS as one whole 128 KiB image, image[k][i] = A[i][k], read with ds_read_b64_tr_b16;
//   * the B panel lives in registers as 16 x 2 bf16x8 MFMA operands — and that is exactly what the accumulator layout
//     of v_mfma_f32_32x32x16_bf16 produces (column = lane & 31, rows in registers), so a product's result panel
//     is the next product's B operand without leaving the register file (rows inside a 16-row k-step appear in the
//     order 4hl+{0..3}, 8+4hl+{0..3}; the A fragments are read in the same order);
//   * a result that the next product needs as A is written from the panels into the LDS image (8-byte stores, one
//     image row per lane) — no global round trip on the critical path.
// Every matrix the kernels keep in HBM (saved iterates, backward work space, X, z_0, d z_iters) is stored "panel
// native" (PN): the 16 bytes a lane feeds to one k-step are contiguous and lanes are consecutive, so panel loads and
// stores are fully coalesced 16-byte accesses and an A image is the same two 8-byte LDS stores per item that a panel
// held in registers takes.  The backward pass is run on the transposed quantities
// (U = dz^T, V3 = dT3^T, V2 = dT2^T, W = dP^T) so that no product ever needs a transposed operand:
//     V3 = 1/4 U Z        V2 = -V3 P        W = V2 P - 7 V2 + P V2 - T2 V3        U' = W X + 1/4 T3 U
//     dX^T = sum_k Z_k W_k
// which also makes the column-major results the row-major dX / dz0 the caller wants (U = PN of dZ^T is packed by
// mh_pinv_chain_pack).
//
// LDS image swizzle (pitch = 512 B, no padding): the 8-byte chunk ch of image row k is stored at chunk ch ^ swz(k),
//   swz(k) = (k1 k2 k3 k0 k1) as bits 0..4.  Transposing reads (4 rows x 64 B per 32 lanes) then hit 64 distinct
//   banks, panel writes (16 lanes, one row each, same chunk) hit 16 distinct chunk slots, and 16-byte row copies
//   stay 16-byte (the two halves swap when k1 = 1).
#include "gemm_kernel.h"
#include <cstdlib>

#ifndef EXP
#define EXP 0
#endif

namespace {

constexpr int CM = 256;        // matrix size
constexpr int CT = 256;        // threads per workgroup
constexpr int NJ = 2;          // 32-column blocks per wave
constexpr int NCH = CM * CM / 8 / CT;   // 16-byte chunks per thread in a row copy
constexpr long MAT = (long)CM * CM;
constexpr int IMG = CM * CM * 2;
// (round 5, measured and removed: the first 8 / 16 items of the forward's image fills requested in front of the epilogue that precedes
//  them — 136-152 B of scratch, forward 224 -> 242 us)
// (Measured and removed in round 4, DESIGN.md section 6: claiming the CU's whole 160 KiB of LDS - neutral; the z_k-only kernels with
//  row-quarter accumulators and three register panels - forward 204 us, backward 708 us against 231 / 458: fewer bytes, more time;
//  image fills whose first half is requested before the barrier - 471 vs 466 us; the round-1 backward kernel without panel prefetch - 496.)
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int swz(int k) { return ((k >> 1) & 7) | ((k & 1) << 3) | (((k >> 1) & 1) << 4); }

__device__ __forceinline__ bf16x8 tr_pair(const char* img, unsigned lo, unsigned hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + hi));
    s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// Addresses of this lane's transposing reads.  Row block blk, k-step T reads image rows 16T + kl (+8) at chunk
// (8 blk + cl) ^ swz(row): the swizzle touches chunk bits 0..4 only, so the address splits into four lane-dependent
// bases (blk & 3, through the XOR) plus the compile-time offset 256 (blk >> 2) + 8192 (T & 7) that fits the ds_read
// immediate; k-steps 8..15 use a second set of bases 64 KiB further on (made opaque, or the compiler folds them back
// into base + constant, materialises all 128 sums and spills them).  16 address registers in all.
__device__ __forceinline__ void read_bases(unsigned (&blo)[2][4], unsigned (&bhi)[2][4], int lane) {
    const int g16 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int kl = 4 * (g16 >> 1) + q, cl = 4 * (g16 & 1) + p;
    const int slo = swz(kl), shi = swz(kl + 8);
#pragma unroll
    for (int b = 0; b < 4; b++) {
        blo[0][b] = kl * 512 + (((8 * b + cl) ^ slo) << 3);
        bhi[0][b] = (kl + 8) * 512 + (((8 * b + cl) ^ shi) << 3);
        blo[1][b] = blo[0][b] + 65536;
        bhi[1][b] = bhi[0][b] + 65536;
        asm volatile("" : "+v"(blo[0][b]), "+v"(bhi[0][b]), "+v"(blo[1][b]), "+v"(bhi[1][b]));
    }
}

// the 8 A fragments (one per 32-row block) of k-step T
template <int T>
__device__ __forceinline__ void load_frags(bf16x8 (&af)[8], const char* img, const unsigned (&blo)[2][4], const unsigned (&bhi)[2][4]) {
#pragma unroll
    for (int blk = 0; blk < 8; blk++) {
        constexpr int hs = T >> 3;
        const int off = (blk >> 2) * 256 + (T & 7) * 8192;
        s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + blo[hs][blk & 3] + off));
        s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + bhi[hs][blk & 3] + off));
        af[blk] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    }
}

// acc[blk][jb] (+)= A(rows 32 blk ..) . B panel (column block jb) over all 256 k.  The fragments of k-step T+1 are
// requested ahead of the MFMAs of k-step T (one wave per SIMD: nobody else hides the LDS latency).
// ZERO: the product starts a new sum (the first k-step takes a zero C instead of 256 accumulator writes).
template <int T, bool ZERO, bool PF = false>
__device__ __forceinline__ void gemm_step(f32x16 (&acc)[8][NJ], bf16x8 (&cur)[8], bf16x8 (&nxt)[8], const char* img,
                                          const unsigned (&blo)[2][4], const unsigned (&bhi)[2][4], bf16x8 (&pB)[16][NJ],
                                          const bf16_t* __restrict__ nextG = nullptr, int wave = 0, int lane = 0) {
    constexpr int TN = (T + 1) & 15, hs = TN >> 3;
#ifndef GEMM_GROUP
#define GEMM_GROUP 2      // row blocks whose next fragments are requested together (measured: 2 beats 1, 4 and 8)
#endif
#pragma unroll
    for (int b0 = 0; b0 < 8; b0 += GEMM_GROUP) {
        if constexpr (T + 1 < 16) {
#pragma unroll
            for (int blk = b0; blk < b0 + GEMM_GROUP; blk++) {
                const int off = (blk >> 2) * 256 + (TN & 7) * 8192;
                s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + blo[hs][blk & 3] + off));
                s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + bhi[hs][blk & 3] + off));
                nxt[blk] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int blk = b0; blk < b0 + GEMM_GROUP; blk++)
#pragma unroll
            for (int jb = 0; jb < NJ; jb++) {
                if (EXP == 1) { acc[blk][jb][0] += (float)cur[blk][0] + (float)pB[T][jb][0]; continue; }   // no MFMA, same data flow
                if constexpr (ZERO && T == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; r++) z[r] = 0.f;
                    acc[blk][jb] = MFMA(cur[blk], pB[T][jb], z);
                } else {
                    acc[blk][jb] = MFMA(cur[blk], pB[T][jb], acc[blk][jb]);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PF) {
        // k-step T has consumed pB[T][*]: the NEXT product's panel entries (panel native, HBM) are requested into them now and
        // land under the remaining k-steps of this product instead of in front of the next one
#pragma unroll
        for (int jb = 0; jb < NJ; jb++)
            pB[T][jb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(nextG + (((2 * wave + jb) * 16 + T) * 512) + (lane << 3)));
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <bool ZERO, bool PF = false>
__device__ __forceinline__ void panel_gemm(f32x16 (&acc)[8][NJ], const char* img, const unsigned (&blo)[2][4],
                                           const unsigned (&bhi)[2][4], bf16x8 (&pB)[16][NJ], const bf16_t* __restrict__ nextG = nullptr,
                                           int wave = 0, int lane = 0) {
    if constexpr (PF) asm volatile("" : "+v"(lane));   // as load_panel: no 32 hoisted 64-bit pointers
    bf16x8 f0[8], f1[8];
    load_frags<0>(f0, img, blo, bhi);
    __builtin_amdgcn_sched_barrier(0);
    gemm_step<0, ZERO, PF>(acc, f0, f1, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<1, ZERO, PF>(acc, f1, f0, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<2, ZERO, PF>(acc, f0, f1, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<3, ZERO, PF>(acc, f1, f0, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<4, ZERO, PF>(acc, f0, f1, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<5, ZERO, PF>(acc, f1, f0, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<6, ZERO, PF>(acc, f0, f1, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<7, ZERO, PF>(acc, f1, f0, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<8, ZERO, PF>(acc, f0, f1, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<9, ZERO, PF>(acc, f1, f0, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<10, ZERO, PF>(acc, f0, f1, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<11, ZERO, PF>(acc, f1, f0, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<12, ZERO, PF>(acc, f0, f1, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<13, ZERO, PF>(acc, f1, f0, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<14, ZERO, PF>(acc, f0, f1, img, blo, bhi, pB, nextG, wave, lane);
    gemm_step<15, ZERO, PF>(acc, f1, f0, img, blo, bhi, pB, nextG, wave, lane);
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[8][NJ]) {
    if (EXP == 5) return;
#pragma unroll
    for (int blk = 0; blk < 8; blk++)
#pragma unroll
        for (int jb = 0; jb < NJ; jb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[blk][jb][r] = 0.f;
}
__device__ __forceinline__ void scale_acc(f32x16 (&acc)[8][NJ], float a) {
#pragma unroll
    for (int blk = 0; blk < 8; blk++)
#pragma unroll
        for (int jb = 0; jb < NJ; jb++) acc[blk][jb] *= a;
}

// out = alpha acc + diag I + rcoef R as the packed bf16 panel (R = another packed panel, same positions).
// The diagonal of column block jb lies in row block 2 wave + jb, at accumulator register dreg of the lanes whose
// half matches (dreg = -1 elsewhere): one compare against a constant per element, nothing to precompute.
template <bool HASR>
__device__ __forceinline__ void finish(const f32x16 (&acc)[8][NJ], float alpha, float diag, const bf16x8 (&pR)[16][NJ],
                                       float rcoef, bf16x8 (&pO)[16][NJ], int wave, int dreg) {
    if (EXP == 2) {   // experiment: no epilogue arithmetic
#pragma unroll
        for (int blk = 0; blk < 8; blk++)
#pragma unroll
            for (int jb = 0; jb < NJ; jb++) {
                asm volatile("" ::"v"(acc[blk][jb][0]));
                pO[2 * blk][jb] = pR[2 * blk][jb];
                pO[2 * blk + 1][jb] = pR[2 * blk + 1][jb];
            }
        return;
    }
#pragma unroll
    for (int blk = 0; blk < 8; blk++)
#pragma unroll
        for (int jb = 0; jb < NJ; jb++) {
            const float dg = (blk == 2 * wave + jb) ? diag : 0.f;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                bf16x8 o;
                u32x4 rw = __builtin_bit_cast(u32x4, pR[2 * blk + t][jb]);
                // opaque: otherwise the f32 values are "remembered" from the epilogue that produced R and spilled
                if constexpr (HASR) asm volatile("" : "+v"(rw));
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int r = 8 * t + e;
                    float v = alpha * acc[blk][jb][r];
                    if (r == dreg) v += dg;
                    if constexpr (HASR) v += rcoef * __uint_as_float((e & 1) ? (rw[e >> 1] & 0xffff0000u) : (rw[e >> 1] << 16));
                    o[e] = (__bf16)v;
                }
                pO[2 * blk + t][jb] = o;
                __builtin_amdgcn_sched_barrier(0);   // one block at a time: hoisting all accumulator reads would spill
            }
        }
}

// finish() that sends the result where the NEXT products read it instead of into the panel registers: the packed bf16 entries go to
// HBM (panel native: the lane that stores an entry is the one that loads it again as a B operand later) and into the LDS image (as
// image_from_panel writes them), straight from the accumulators.  The panel registers therefore keep what they hold (a B operand that
// the next product uses again, or the next operand that panel_gemm<.., PF> requested behind the k sweep).  Call it between two
// barriers: every wave must have finished reading the image.
template <bool HASR, bool NTS = false, bool NEXT = false>      // NTS: the global copy is read again only much later (W_k: in the dX sum behind the loop)
__device__ __forceinline__ void finish_out(const f32x16 (&acc)[8][NJ], float alpha, float diag, bf16x8 (&pR)[16][NJ], float rcoef,
                                           bf16_t* __restrict__ G, char* img, int wave, int dreg, int j0, int hl, int lane,
                                           const bf16_t* __restrict__ nextG = nullptr) {
    // NEXT: an entry of the addend panel is dead once its block is finished: the NEXT product's B operand (panel native, HBM) is
    // requested into it right there, so that panel arrives under the rest of this epilogue instead of in front of its product
    asm volatile("" : "+v"(j0), "+v"(hl), "+v"(lane));
    const int s = swz(j0);
#pragma unroll
    for (int blk = 0; blk < 8; blk++)
#pragma unroll
        for (int jb = 0; jb < NJ; jb++) {
            const float dg = (blk == 2 * wave + jb) ? diag : 0.f;
            char* row = img + (j0 + 32 * jb) * 512;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int T = 2 * blk + t;
                bf16x8 o;
                u32x4 rw = __builtin_bit_cast(u32x4, pR[T][jb]);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int r = 8 * t + e;
                    float v = alpha * acc[blk][jb][r];
                    if (r == dreg) v += dg;
                    if constexpr (HASR) v += rcoef * __uint_as_float((e & 1) ? (rw[e >> 1] & 0xffff0000u) : (rw[e >> 1] << 16));
                    o[e] = (__bf16)v;
                }
                const u32x4 v = __builtin_bit_cast(u32x4, o);
                if constexpr (NTS) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(G + (((2 * wave + jb) * 16 + T) * 512) + (lane << 3)));
                else *reinterpret_cast<u32x4*>(G + (((2 * wave + jb) * 16 + T) * 512) + (lane << 3)) = v;
                *reinterpret_cast<u32x2*>(row + (((4 * T + hl) ^ s) << 3)) = u32x2{v[0], v[1]};
                *reinterpret_cast<u32x2*>(row + (((4 * T + 2 + hl) ^ s) << 3)) = u32x2{v[2], v[3]};
                if constexpr (NEXT)
                    pR[T][jb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(nextG + (((2 * wave + jb) * 16 + T) * 512) + (lane << 3)));
                __builtin_amdgcn_sched_barrier(0);   // one block at a time: hoisting all accumulator reads would spill
            }
        }
}

// HBM storage of chain-private matrices, "panel native": PN[jblk][T][lane][8] holds the bf16x8 that lane (c = lane & 31,
// hl = lane >> 5) of column block jblk feeds to k-step T, i.e. M[16T + 4hl + {0..3}, 16T + 8 + 4hl + {0..3}][32 jblk + c]:
// every panel load / store is one fully coalesced 16-byte access per lane.
// NT: non-temporal stores for matrices that only the BACKWARD launch reads again (P_k, T2_k, T3_k: 3 x 128 KiB per workgroup and
// iteration) — kept in L2 like ordinary lines they evict X and z_k, which this workgroup reads back as images one product later
template <bool NT = false>
__device__ __forceinline__ void store_panel(bf16_t* __restrict__ G, const bf16x8 (&p)[16][NJ], int wave, int lane) {
    if (EXP == 4) return;
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int jb = 0; jb < NJ; jb++)
#pragma unroll
        for (int T = 0; T < 16; T++) {
            u32x4* dst = reinterpret_cast<u32x4*>(G + (((2 * wave + jb) * 16 + T) * 512) + (lane << 3));
            if constexpr (NT) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, p[T][jb]), dst);
            else *dst = __builtin_bit_cast(u32x4, p[T][jb]);
        }
}
template <bool NT = false>      // NT: the LAST read of a matrix the forward launch saved (streamed through once): a non-temporal load
__device__ __forceinline__ void load_panel(bf16x8 (&p)[16][NJ], const bf16_t* __restrict__ G, int wave, int lane) {
    asm volatile("" : "+v"(lane));   // loop-invariant sources (X) would otherwise get 32 hoisted, spilled 64-bit pointers
#pragma unroll
    for (int jb = 0; jb < NJ; jb++)
#pragma unroll
        for (int T = 0; T < 16; T++) {
            const u32x4* src = reinterpret_cast<const u32x4*>(G + (((2 * wave + jb) * 16 + T) * 512) + (lane << 3));
            p[T][jb] = __builtin_bit_cast(bf16x8, NT ? __builtin_nontemporal_load(src) : *src);
        }
}
__device__ __forceinline__ void negate_panel(bf16x8 (&p)[16][NJ]) {
#pragma unroll
    for (int T = 0; T < 16; T++)
#pragma unroll
        for (int jb = 0; jb < NJ; jb++) {
            u32x4 v = __builtin_bit_cast(u32x4, p[T][jb]);
            v ^= 0x80008000u;
            p[T][jb] = __builtin_bit_cast(bf16x8, v);
        }
}
// f32 column-major store of alpha * acc (i.e. row-major of the transposed matrix)
__device__ __forceinline__ void store_f32(float* __restrict__ G, const f32x16 (&acc)[8][NJ], float alpha, int j0, int hl) {
    asm volatile("" : "+v"(j0), "+v"(hl));
#pragma unroll
    for (int jb = 0; jb < NJ; jb++) {
        float* row = G + (long)(j0 + 32 * jb) * CM + 4 * hl;
#pragma unroll
        for (int blk = 0; blk < 8; blk++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = alpha * acc[blk][jb][4 * g + e];
                *reinterpret_cast<f32x4*>(row + 32 * blk + 8 * g) = v;
                __builtin_amdgcn_sched_barrier(0);
            }
    }
}

// panels -> LDS image (image row j = this lane's column)
__device__ __forceinline__ void image_from_panel(char* img, const bf16x8 (&p)[16][NJ], int j0, int hl) {
    asm volatile("" : "+v"(j0), "+v"(hl));   // keep the swizzled addresses out of loop-invariant hoisting (they would be spilled)
    const int s = swz(j0);         // bits 0..3 of the column only: the same for both column blocks
#pragma unroll
    for (int jb = 0; jb < NJ; jb++) {
        char* row = img + (j0 + 32 * jb) * 512;
#pragma unroll
        for (int T = 0; T < 16; T++) {
            const u32x4 v = __builtin_bit_cast(u32x4, p[T][jb]);
            *reinterpret_cast<u32x2*>(row + (((4 * T + hl) ^ s) << 3)) = u32x2{v[0], v[1]};
            *reinterpret_cast<u32x2*>(row + (((4 * T + 2 + hl) ^ s) << 3)) = u32x2{v[2], v[3]};
        }
    }
}
// panel-native HBM matrix -> LDS image, BATCH 16-byte items in flight per thread (fewer while an accumulator tile is
// live).  Item it = tid + 256 n is lane (tid & 63) of k-step wave + 4 (n & 3) of column block n >> 2.
template <int BATCH, bool NT = false>
__device__ __forceinline__ void image_from_global(char* img, const bf16_t* __restrict__ G, int tid) {
    if (EXP == 3) return;
    asm volatile("" : "+v"(tid));   // recompute the addresses at every call instead of hoisting + spilling them
    const int lane = tid & 63, c = lane & 31, hl = lane >> 5, s = swz(c), tw = tid >> 6;
#pragma unroll
    for (int n0 = 0; n0 < 32; n0 += BATCH) {
        u32x4 r[BATCH];
#pragma unroll
        for (int n = 0; n < BATCH; n++) {
            const u32x4* src = reinterpret_cast<const u32x4*>(G + ((long)(tid + CT * (n0 + n)) << 3));
            r[n] = NT ? __builtin_nontemporal_load(src) : *src;
        }
#pragma unroll
        for (int n = 0; n < BATCH; n++) {
            const int jblk = (n0 + n) >> 2, T = tw + 4 * ((n0 + n) & 3);
            char* row = img + (32 * jblk + c) * 512;
            *reinterpret_cast<u32x2*>(row + (((4 * T + hl) ^ s) << 3)) = u32x2{r[n][0], r[n][1]};
            *reinterpret_cast<u32x2*>(row + (((4 * T + 2 + hl) ^ s) << 3)) = u32x2{r[n][2], r[n][3]};
        }
    }
}
// ... __syncthreads() on both sides included
template <bool NT = false>
__device__ __forceinline__ void image_swap(char* img, const bf16_t* __restrict__ G, int tid) {
    __syncthreads();
    image_from_global<16, NT>(img, G, tid);
    __syncthreads();
}
// LDS image -> column-major HBM matrix G[j][i] (row copy, coalesced): the form the caller's GEMMs read
__device__ __forceinline__ void image_to_global(const char* img, bf16_t* __restrict__ G, int tid) {
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int n = 0; n < NCH; n++) {
        const int cid = tid + CT * n, k = cid >> 5, c16 = cid & 31, s = swz(k);
        u32x4 v = *reinterpret_cast<const u32x4*>(img + k * 512 + ((c16 ^ (s >> 1)) << 4));
        if (s & 1) v = u32x4{v[2], v[3], v[0], v[1]};
        *reinterpret_cast<u32x4*>(G + (long)cid * 8) = v;
    }
}
// make this workgroup's global stores visible to its other waves, and wait until nobody reads the LDS image any more
__device__ __forceinline__ void publish() {
    // workgroup scope only (all waves share this CU's L1, stores are write-through, and every line is written before
    // its first read): an agent-scope fence writes back and invalidates the whole L2 every time
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---------------------------------------------------------------------------------------------------------- forward
// XP = PN(X); saved[k] = PN{z_k, P_k, T2_k, T3_k}, z_0 pre-filled; zfT[j][i] = z_iters[i][j] (column-major)
__global__ __launch_bounds__(CT) void pinv_panel_fwd_kernel(const bf16_t* __restrict__ XT, bf16_t* __restrict__ saved,
                                                            bf16_t* __restrict__ zfT, int BH, int iters,
                                                            const float* __restrict__ z0f, const unsigned long long* __restrict__ st, int z0_rm) {
    __shared__ __attribute__((aligned(16))) char img[IMG];
    const int tid = threadIdx.x, lane = tid & 63, hl = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: panel addresses become SGPR base + one VGPR
    const int j = 64 * wave + (lane & 31);
    const int bh = blockIdx.x;
    const int dreg = (hl == ((lane >> 2) & 1)) ? ((((lane & 31) >> 3) << 2) | (lane & 3)) : -1;
    unsigned rlo[2][4], rhi[2][4];
    read_bases(rlo, rhi, lane);
    const bf16_t* Xb = XT + bh * MAT;
    f32x16 acc[8][NJ];
    bf16x8 p[16][NJ];       // THE panel: B operand of the running product, then (in place) its result
    if (z0f) {
        // z_0 = attn2^T / (c r) from mh_nys_sim2's unscaled f32 panel-native attn2^T and the tensor-wide maxima (complete only now,
        // after that launch); rounded to bf16 once, and left in saved[0] for the backward
        const float inv = 1.f / (__uint_as_float((unsigned)(st[0] >> 32)) * __uint_as_float((unsigned)(st[1] >> 32)));
        const float* zb = z0f + bh * MAT;
#pragma unroll
        for (int jb = 0; jb < NJ; jb++)
#pragma unroll
            for (int T = 0; T < 16; T++) {
                // z0_rm: z0f is x itself, row-major (mh_nys_sim2's attn2): column j of z_0 = x^T / (c r) is ROW j of x, and a panel entry is
                // two runs of four consecutive elements of that row (the lane reads its own row: 1 KiB apart between lanes, L2-hot) —
                // no transposed f32 copy of attn2 is ever written.  Otherwise: the unscaled panel-native f32 x^T
                const float* src = z0_rm ? zb + (long)(64 * wave + 32 * jb + (lane & 31)) * CM + 16 * T + 4 * hl
                                         : zb + (((2 * wave + jb) * 16 + T) * 512) + (lane << 3);
                const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + (z0_rm ? 8 : 4));
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; e++) { o[e] = (__bf16)(a[e] * inv); o[4 + e] = (__bf16)(b[e] * inv); }
                p[T][jb] = o;
            }
        store_panel(saved + bh * MAT, p, wave, lane);
    } else {
        load_panel(p, saved + bh * MAT, wave, lane);                       // z_0
    }
    image_from_global<16>(img, Xb, tid);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < iters; k++) {
        bf16_t* base = saved + ((long)k * 4 * BH + bh) * MAT;              // [iters][4][BH][m][m]
        bf16_t* P = base + (long)BH * MAT;
        bf16_t* T2 = base + 2L * BH * MAT;
        bf16_t* T3 = base + 3L * BH * MAT;
        panel_gemm<true>(acc, img, rlo, rhi, p);                                // P = X z
        finish<false>(acc, 1.f, 0.f, p, 0.f, p, wave, dreg);
        store_panel<true>(P, p, wave, lane);
        __syncthreads();
        image_from_panel(img, p, j, hl);
        __syncthreads();
        panel_gemm<true>(acc, img, rlo, rhi, p);                                // T2 = 15I - 7P + P P
        finish<true>(acc, 1.f, 15.f, p, -7.f, p, wave, dreg);
        store_panel<true>(T2, p, wave, lane);
        panel_gemm<true>(acc, img, rlo, rhi, p);                                // T3 = 13I - P T2
        finish<false>(acc, -1.f, 13.f, p, 0.f, p, wave, dreg);
        store_panel<true>(T3, p, wave, lane);
        publish();
        image_from_global<16>(img, base, tid);                            // z_k (written by this workgroup one step ago)
        __syncthreads();
        panel_gemm<true>(acc, img, rlo, rhi, p);                                // z' = 1/4 z T3
        finish<false>(acc, 0.25f, 0.f, p, 0.f, p, wave, dreg);
        if (k + 1 < iters) {
            store_panel(saved + ((long)(k + 1) * 4 * BH + bh) * MAT, p, wave, lane);
            publish();
            image_from_global<16>(img, Xb, tid);
        }
        __syncthreads();
    }
    image_from_panel(img, p, j, hl);                                      // z_iters leaves column-major, via the image
    __syncthreads();
    image_to_global(img, zfT + bh * MAT, tid);
}


// --------------------------------------------------------------------------------- backward, B panels requested a product ahead
// Same algebra, same saved / work layout and the same 48 + 6 products as the round-1 backward kernel.  What changes is where results go
// and when operands arrive: a product's result leaves the accumulators for HBM + the LDS image directly (finish_out), so the panel
// registers are free to receive the NEXT product's B operand entry by entry behind this product's k sweep (panel_gemm<.., PF>), and a
// B operand that two consecutive products share (P in V2 = -V3 P and V2 P) is loaded once.  Of the 8 panel loads per iteration that
// stood exposed in front of their product, none is left in round 5's order (V2 stays in the panel for the W epilogue, X is requested
// inside that epilogue, the next Z behind the last product).
__global__ __launch_bounds__(CT) void pinv_panel_bwd2_kernel(const bf16_t* __restrict__ XT, const bf16_t* __restrict__ saved,
                                                             const bf16_t* __restrict__ dzf, bf16_t* __restrict__ work,
                                                             float* __restrict__ dX, float* __restrict__ dz0, int BH, int iters) {
    __shared__ __attribute__((aligned(16))) char img[IMG];
    const int tid = threadIdx.x, lane = tid & 63, hl = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = 64 * wave + (lane & 31);
    const int bh = blockIdx.x;
    const int dreg = (hl == ((lane >> 2) & 1)) ? ((((lane & 31) >> 3) << 2) | (lane & 3)) : -1;
    unsigned rlo[2][4], rhi[2][4];
    read_bases(rlo, rhi, lane);
    const bf16_t* Xb = XT + bh * MAT;
    f32x16 acc[8][NJ];
    bf16x8 p[16][NJ];
    const bf16_t* U = dzf + bh * MAT;
    image_from_global<16>(img, U, tid);
    load_panel(p, saved + ((long)(iters - 1) * 4 * BH + bh) * MAT, wave, lane);            // Z of the first (last) iteration
    __syncthreads();
#pragma unroll 1
    for (int k = iters - 1; k >= 0; k--) {
        const bf16_t* sb = saved + ((long)k * 4 * BH + bh) * MAT;
        const bf16_t* P = sb + (long)BH * MAT;
        const bf16_t* T2 = sb + 2L * BH * MAT;
        const bf16_t* T3 = sb + 3L * BH * MAT;
        const bf16_t* Zn = saved + ((long)(k > 0 ? k - 1 : 0) * 4 * BH + bh) * MAT;         // next iteration's Z (k == 0: a harmless reload)
        // work = [iters + 4][BH] matrices: W_k (kept for dX below) in slot k, then V3, V2 (temporaries of one iteration) and two slots for
        // U' (written in iteration k, read as U in iteration k - 1)
        bf16_t* W = work + ((long)k * BH + bh) * MAT;
        bf16_t* V3 = work + ((long)iters * BH + bh) * MAT;
        bf16_t* V2 = work + ((long)(iters + 1) * BH + bh) * MAT;
        bf16_t* Un = work + ((long)(iters + 2 + (k & 1)) * BH + bh) * MAT;
        // V3 = 1/4 U Z                                  (image U, panel Z; P arrives behind the sweep)
        panel_gemm<true, true>(acc, img, rlo, rhi, p, P, wave, lane);
        __syncthreads();
        finish_out<false>(acc, 0.25f, 0.f, p, 0.f, V3, img, wave, dreg, j, hl, lane);
        __syncthreads();
        // V2 = -V3 P                                    (image V3, panel P: stays for the next product)
        panel_gemm<true, false>(acc, img, rlo, rhi, p);
        __syncthreads();
        finish_out<false>(acc, -1.f, 0.f, p, 0.f, V2, img, wave, dreg, j, hl, lane);
        __syncthreads();
        // W = V2 P - 7 V2 + P V2 - T2 V3.  Order (round 5): V2 P, then - T2 V3, then P V2 LAST — its panel IS V2, the epilogue's addend, so
        // nothing has to be re-read for the - 7 V2 term (round 3 had P V2 in the middle and requested V2 a third time behind the last
        // product: entry by entry it cost 28 spilled registers and waits inside the MFMA loop, as one exposed panel load 1.8 us per iteration)
        panel_gemm<true, true>(acc, img, rlo, rhi, p, V3, wave, lane);                    // V2 P   (image V2, panel P; V3 arrives: own stores)
        negate_panel(p);
        image_swap<true>(img, T2, tid);
        panel_gemm<false, true>(acc, img, rlo, rhi, p, V2, wave, lane);                   // - T2 V3 (image T2, panel -V3; V2 arrives: own stores)
        image_swap<true>(img, P, tid);
        panel_gemm<false, false>(acc, img, rlo, rhi, p);                                  // + P V2 (image P, panel V2: stays for the epilogue)
        __syncthreads();
        // 4 W (exact in bf16); X, the next product's panel, is requested entry by entry behind the addend entries as they are consumed
        finish_out<true, true, true>(acc, 4.f, 0.f, p, -28.f, W, img, wave, dreg, j, hl, lane, Xb);
        __syncthreads();
        // U' = 1/4 (4W X + T3 U)
        panel_gemm<true, true>(acc, img, rlo, rhi, p, U, wave, lane);                     // 4W X   (image 4W, panel X; U arrives)
        image_swap<true>(img, T3, tid);
        panel_gemm<false, false>(acc, img, rlo, rhi, p);                                  // + T3 U (image T3, panel U)
        load_panel(p, Zn, wave, lane);            // the next iteration's Z: the panel is free from here on, the loads fly under the epilogue
        __syncthreads();
        finish_out<false>(acc, 0.25f, 0.f, p, 0.f, Un, img, wave, dreg, j, hl, lane);
        if (k == 0) store_f32(dz0 + bh * MAT, acc, 0.25f, j, hl);
        __syncthreads();
        U = Un;
    }
    // dX^T = sum_k Z_k W_k
    zero_acc(acc);
    load_panel(p, work + ((long)bh) * MAT, wave, lane);                                     // W_0 (own stores)
#pragma unroll 1
    for (int k = 0; k < iters; k++) {
        const bf16_t* Z = saved + ((long)k * 4 * BH + bh) * MAT;
        const bf16_t* Wn = work + ((long)(k + 1 < iters ? k + 1 : k) * BH + bh) * MAT;
        image_swap<true>(img, Z, tid);
        panel_gemm<false, true>(acc, img, rlo, rhi, p, Wn, wave, lane);
    }
    store_f32(dX + bh * MAT, acc, 0.25f, j, hl);      // the panels hold 4 W
}


// One thread per panel-native item (bh, jblk, T, lane): i_e = 16T + 4hl + (e & 3) + 8 (e >> 2), j = 32 jblk + c.
__device__ __forceinline__ void pn_item(int it, int& j, int& i0) {
    const int lane = it & 63, T = (it >> 6) & 15, jblk = it >> 10;
    j = 32 * jblk + (lane & 31);
    i0 = 16 * T + 4 * (lane >> 5);
}
__device__ __forceinline__ u32x4 pack_bf16x8(const float (&v)[8]) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; w++) o[w] = pack_bf2(v[2 * w], v[2 * w + 1]);
    return o;
}

// x = attn2 (f32 row-major) -> xp = PN(x), z0p = PN(z0), z0 = x^T / (c r) in f32 row-major (for mh_pinv_z0_bwd)
__global__ __launch_bounds__(256) void pinv_chain_prep_kernel(const float* __restrict__ x, const unsigned long long* __restrict__ st,
                                                              float* __restrict__ z0, bf16_t* __restrict__ xp,
                                                              bf16_t* __restrict__ z0p) {
    const float c = __uint_as_float((unsigned)(st[0] >> 32)), r = __uint_as_float((unsigned)(st[1] >> 32));
    const float inv = 1.f / (c * r);
    const int it = blockIdx.x * 256 + threadIdx.x;          // 8192 items per matrix
    const long base = (long)blockIdx.y * MAT;
    int j, i0;
    pn_item(it, j, i0);
    float xv[8], zv[8];
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + base + (long)j * CM + i0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(x + base + (long)j * CM + i0 + 8);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int i = i0 + (e & 3) + 8 * (e >> 2);
        xv[e] = x[base + (long)i * CM + j];                   // X[i][j]
        zv[e] = (e < 4 ? a[e] : b[e - 4]) * inv;              // z0[i][j] = x[j][i] / (c r)
        z0[base + (long)i * CM + j] = zv[e];
    }
    *reinterpret_cast<u32x4*>(xp + base + (long)it * 8) = pack_bf16x8(xv);
    *reinterpret_cast<u32x4*>(z0p + base + (long)it * 8) = pack_bf16x8(zv);
}

// dz (f32 row-major, d z_iters) -> PN(U), U = dz^T
__global__ __launch_bounds__(256) void pinv_chain_pack_kernel(const float* __restrict__ dz, bf16_t* __restrict__ up) {
    const int it = blockIdx.x * 256 + threadIdx.x;
    const long base = (long)blockIdx.y * MAT;
    int j, i0;
    pn_item(it, j, i0);
    const f32x4 a = *reinterpret_cast<const f32x4*>(dz + base + (long)j * CM + i0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(dz + base + (long)j * CM + i0 + 8);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};   // U[i][j] = dz[j][i]
    *reinterpret_cast<u32x4*>(up + base + (long)it * 8) = pack_bf16x8(v);
}

}  // namespace

extern "C" int mh_pinv_chain_prep(const float* x, const uint64_t* stats64, float* z0, void* xp, void* z0p, int BH, int m,
                                  mh_stream s) {
    MH_REQUIRE(m == CM, "mh_pinv_chain_prep: m=%d unsupported (the chain kernels are built for m = %d; other sizes use mh_gemm)", m, CM);
    if (BH <= 0) return MH_OK;
    hipLaunchKernelGGL(pinv_chain_prep_kernel, dim3(MAT / 8 / 256, BH), dim3(256), 0, (hipStream_t)s, x,
                       (const unsigned long long*)stats64, z0, (bf16_t*)xp, (bf16_t*)z0p);
    MH_LAUNCH_CHECK("mh_pinv_chain_prep");
    return MH_OK;
}

extern "C" int mh_pinv_chain_pack(const float* dz, void* up, int BH, int m, mh_stream s) {
    MH_REQUIRE(m == CM, "mh_pinv_chain_pack: m=%d unsupported (built for m = %d)", m, CM);
    if (BH <= 0) return MH_OK;
    hipLaunchKernelGGL(pinv_chain_pack_kernel, dim3(MAT / 8 / 256, BH), dim3(256), 0, (hipStream_t)s, dz, (bf16_t*)up);
    MH_LAUNCH_CHECK("mh_pinv_chain_pack");
    return MH_OK;
}

extern "C" int mh_pinv_chain_fwd(const void* XT, void* saved, void* zfT, int BH, int m, int iters, const float* z0f, const uint64_t* stats64,
                                 int z0_rowmajor, mh_stream s) {
    MH_REQUIRE(m == CM, "mh_pinv_chain_fwd: m=%d unsupported (built for m = %d; other sizes use mh_gemm)", m, CM);
    MH_REQUIRE(iters >= 1 && BH >= 0, "mh_pinv_chain_fwd: bad arguments");
    if (BH == 0) return MH_OK;
#ifdef MH_EXP       // timing-experiment builds only (make EXP=1): what the step costs without the chain; results are garbage
    if (getenv("MH_EXP_CHAIN_SKIP")) return MH_OK;
    if (const char* e = getenv("MH_EXP_CHAIN_FWD_ITERS")) iters = atoi(e);     // what a faster chain would buy: fewer iterations, same launch
#endif
    MH_REQUIRE(!z0f || (stats64 && ((uintptr_t)z0f & 15) == 0), "mh_pinv_chain_fwd: z0f needs the maxima and 16-byte alignment");
    hipLaunchKernelGGL(pinv_panel_fwd_kernel, dim3(BH), dim3(CT), 0, (hipStream_t)s, (const bf16_t*)XT, (bf16_t*)saved,
                       (bf16_t*)zfT, BH, iters, z0f, (const unsigned long long*)stats64, z0_rowmajor);
    MH_LAUNCH_CHECK("mh_pinv_chain_fwd");
    return MH_OK;
}

extern "C" int mh_pinv_chain_bwd(const void* XT, const void* saved, const void* dzf, void* work, float* dX, float* dz0, int BH,
                                 int m, int iters, mh_stream s) {
    MH_REQUIRE(m == CM, "mh_pinv_chain_bwd: m=%d unsupported (built for m = %d; other sizes use mh_gemm)", m, CM);
    MH_REQUIRE(iters >= 1 && BH >= 0, "mh_pinv_chain_bwd: bad arguments");
    if (BH == 0) return MH_OK;
#ifdef MH_EXP
    if (getenv("MH_EXP_CHAIN_SKIP")) return MH_OK;
    if (const char* e = getenv("MH_EXP_CHAIN_BWD_ITERS")) iters = atoi(e);
#endif
    hipLaunchKernelGGL(pinv_panel_bwd2_kernel, dim3(BH), dim3(CT), 0, (hipStream_t)s, (const bf16_t*)XT, (const bf16_t*)saved,
                       (const bf16_t*)dzf, (bf16_t*)work, dX, dz0, BH, iters);
    MH_LAUNCH_CHECK("mh_pinv_chain_bwd");
    return MH_OK;
}

extern "C" int64_t mh_pinv_chain_workspace_bytes(int BH, int m, int iters, int which) {
    if (BH <= 0 || m != CM || iters < 1 || which < 0 || which > 1) return 0;
    return (which == 0 ? (int64_t)iters * 4 : (int64_t)iters + 4) * BH * MAT * 2;
}
