// MFMA GEMM for gfx950: one templated kernel behind mh_gemm() (instantiated in gemm_f32.hip / gemm_bf16.hip /
// gemm_mixed.hip so the three families compile in parallel).
//
//   block  = 256 threads = 4 waves (2 x 2), wave tile = (32*WM) x (32*WN), block tile (64*WM) x (64*WN)
//   MMA    = bf16: v_mfma_f32_32x32x16_bf16, BK = 64   |   f32: v_mfma_f32_32x32x2_f32 (exact), BK = 16
//   staging: global -> registers (16-B loads, prefetched one K-tile ahead) -> LDS (double buffered),
//            f32 operands are rounded to bf16 on the way into LDS when MMA = bf16
//   LDS images (bank maths from MI355X_MICROARCH.md §LDS):
//     K-contiguous operand  : [rows][BK]  pitch BK+8 bf16 / 20 f32 -> ds_read_b128 fragments, conflict free
//     K-strided operand bf16: [BK][rows]  pitch rows+32            -> ds_read_b64_tr_b16 (hardware transpose)
//     K-strided operand f32 : [BK][rows]  pitch rows+4             -> ds_read_b32, lanes consecutive
//   f32 MMA k-order trick: lane half h supplies k = 8h + s at MFMA step s for BOTH operands, so each
//   lane reads 8 consecutive k with two ds_read_b128 (any k permutation shared by A and B is legal).
//   FULL = true: M, N, K are tile multiples and both operands are 16-B aligned -> no bounds logic at all in the
//   main loop or the epilogue (the guarded variant costs branches + early waits around every load/store).
//   Workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of tiles.
#pragma once
#include "common.h"
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The K % 64 remainder of a weight gradient (one row per slide: B x 4097 rows) as a rank-(batch x KT) update: when the main launch
// reduces through partial tiles + a fold pass, the fold adds it (C[m][n] += alpha sum_{z, k < KT} A_z[k][m] B_z[k][n], both operands
// with the contraction index as their row) instead of a launch of its own (gemm.hip: try_rank_update).
struct GemmTail {
    const void* A; const void* B;
    long lda, sA, ldb, sB;
    int KT, batch, a_f32, b_f32;
    float alpha;
};
void gemm_note_variant(const char* fmt, ...);      // errors.cpp: names the instance a launch site picked (mh_gemm_variant_name)
template <typename T> inline const char* gemm_tn() { return sizeof(T) == 4 ? "float" : "bf16"; }
inline const char* gemm_tf(bool b) { return b ? "true" : "false"; }
GemmTail* gemm_pending_tail();       // the tail mh_gemm offers to the next fold launch (KT == 0: none); the fold clears it when it takes it

struct GemmArgs {
    const void* A; const void* B; void* C; const float* bias;
    int M, N, K;
    long lda, ldb, ldc;
    long sA1, sA2, sB1, sB2, sC1, sC2;
    int batch2;
    float alpha, diag;
    int act, accumulate, split_k, k_per_split;
    int vecA, vecB;
    int atomic;  // f32 atomicAdd into C: split-K, or a batch that broadcasts into one C
    int tiles_m, tiles_n;
    int vecC;    // C rows are 16-B aligned (LDS-staged wide-store epilogue allowed)
    const void* R; float rcoef;   // optional addend rcoef * R, R laid out exactly like C (same dtype and strides)
    float* ws; long ws_floats;    // split-K partial tiles go here instead of f32 atomics (gemm_big.hip), then a fold pass
    const float* scale_a; const float* scale_b;   // fp8 operands (gemm_big.hip FP8 instance): alpha *= scale_a[0] * scale_b[0]
    int row_softmax;              // gemm_tile.hip, N == 384, bf16 C: C = softmax over each row of alpha * A B (the whole row is in one tile)
    int kseg; long sAk, sBk;      // C = sum over kseg operand pairs (A + s sAk, B + s sBk), K each (gemm_tile.hip only; 0 / 1: one pair)
    mh_gemm_epi epi;              // fused epilogue (gemm_big.hip only; kind 0: none)
    int a_rpb, a_skip;            // row-window remap of A (gemm_big.hip, K-contiguous A): flat row r -> r + (r / a_rpb) * a_skip
    int shared_chip;              // mh_gemm_desc.shared_chip: no persistent kernel
    int c_rpb, c_skip;            // row-window remap of C (gemm_big.hip, bf16 C): flat row r -> r + (r / c_rpb) * c_skip
    int w_last;                   // index of the last row window (mh_gemm_desc.window_batches - 1; 1 << 30: unlimited): rows past it follow it
};

template <int MMA, bool KC, int ROWS>
struct TileGeom {
    static constexpr int BK = MMA ? 64 : 16;
    static constexpr int ESZ = MMA ? 2 : 4;
    static constexpr int LROWS = KC ? ROWS : BK;
    static constexpr int PITCH = KC ? (MMA ? BK + 8 : 20) : (MMA ? ROWS + 32 : ROWS + 4);
    static constexpr int BYTES = LROWS * PITCH * ESZ;
};

// ------------------------------------------------------------------ global -> regs -> LDS
// The staging registers are a plain local array of the kernel (passed by reference): as a struct member the
// compiler kept them in scratch memory, which put an s_waitcnt right behind every global load.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// FULL: 0 = guarded loads, 1 = whole tiles everywhere, 2 = whole tiles in M / N with a ragged last K-tile (K % VEC == 0),
//       3 = this operand's free dimension ends inside the tile (dim % VEC == 0): clamped address + select, K whole
template <int MMA, typename TG, bool KC, int ROWS, int FULL, int NT = 256>
struct Stager {
    using G = TileGeom<MMA, KC, ROWS>;
    static constexpr int VEC = 16 / (int)sizeof(TG);
    static constexpr int CONTIG = KC ? G::BK : ROWS;
    static constexpr int CPR = CONTIG / VEC;
    static constexpr int NCH = G::LROWS * CPR / NT;
    static_assert(G::LROWS * CPR % NT == 0, "tile must split evenly over the block's threads");

    static __device__ __forceinline__ void load(u32x4 (&regs)[NCH], const TG* __restrict__ base, long ld, int tile0,
                                                int dim, int k0, int kend, bool vec_ok, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int cid = tid + i * NT;
            const int r = cid / CPR, c = cid % CPR;
            int gm, gk;
            long off;
            // FULL + K-contiguous rows: a ragged last row-tile re-reads row dim-1 (its results are never stored)
            if (KC) { gm = FULL ? min(tile0 + r, dim - 1) : tile0 + r; gk = k0 + c * VEC; off = (long)gm * ld + gk; }
            else    { gk = k0 + r; gm = tile0 + c * VEC; off = (long)gk * ld + gm; }
            if constexpr (FULL == 1) {
                regs[i] = *reinterpret_cast<const u32x4*>(base + off);
            } else if constexpr (FULL == 2) {
                // ragged K only (the 96-wide heads of the template geometry: K = 96 = 1.5 K-tiles): a 16-byte chunk is inside or
                // outside K as a whole -> clamped address + select, no branches (the guarded path costs a divergent branch and
                // an early wait per chunk: 2.3x on those products)
                const bool in = KC ? (gk + VEC <= kend) : (gk < kend);
                const long o2 = KC ? (long)gm * ld + min(gk, kend - VEC) : (long)min(gk, kend - 1) * ld + gm;
                u32x4 v = *reinterpret_cast<const u32x4*>(base + o2);
                if (!in) v = u32x4{0u, 0u, 0u, 0u};
                regs[i] = v;
            } else if constexpr (FULL == 3) {
                // ragged free dimension only (N = 96 of the template's attention products): rows past the end re-read the last
                // row (K-contiguous form, gm is clamped above: those results are never stored), chunks past the end read as zeros
                if (KC) {
                    regs[i] = *reinterpret_cast<const u32x4*>(base + off);
                } else {
                    const bool in = gm + VEC <= dim;
                    u32x4 v = *reinterpret_cast<const u32x4*>(base + (long)gk * ld + min(gm, dim - VEC));
                    if (!in) v = u32x4{0u, 0u, 0u, 0u};
                    regs[i] = v;
                }
            } else {
                const bool row_ok = KC ? (gm < dim) : (gk < kend);
                const int cstart = KC ? gk : gm;
                const int climit = KC ? kend : dim;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (row_ok) {
                    if (vec_ok && cstart + VEC <= climit) {
                        v = *reinterpret_cast<const u32x4*>(base + off);
                    } else if constexpr (sizeof(TG) == 4) {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (cstart + e < climit) v[e] = __float_as_uint(reinterpret_cast<const float*>(base)[off + e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; e++)
                            if (cstart + e < climit) v[e >> 1] |= (unsigned)reinterpret_cast<const bf16_t*>(base)[off + e] << (16 * (e & 1));
                    }
                }
                regs[i] = v;
            }
        }
    }

    static __device__ __forceinline__ void store(const u32x4 (&regs)[NCH], char* tile, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int cid = tid + i * NT;
            const int r = cid / CPR, c = cid % CPR;
            const u32x4 v = regs[i];
            if constexpr ((int)sizeof(TG) == G::ESZ) {
                *reinterpret_cast<u32x4*>(tile + (r * G::PITCH + c * VEC) * G::ESZ) = v;
            } else {  // f32 in HBM -> bf16 in LDS
                u32x2 p;
                p[0] = pack_bf2(__uint_as_float(v[0]), __uint_as_float(v[1]));
                p[1] = pack_bf2(__uint_as_float(v[2]), __uint_as_float(v[3]));
                *reinterpret_cast<u32x2*>(tile + (r * G::PITCH + c * 4) * 2) = p;
            }
        }
    }
};

// ------------------------------------------------------------------ LDS -> MFMA fragments
// bf16: 8 consecutive k (k0 + 8*(lane>>5) + j) of tile row (row0 + (lane&31))
template <bool KC, int ROWS>
__device__ __forceinline__ bf16x8 frag_bf16(const char* tile, int row0, int k0, int lane) {
    using G = TileGeom<1, KC, ROWS>;
    if constexpr (KC) {
        const int r = lane & 31, hh = lane >> 5;
        return *reinterpret_cast<const bf16x8*>(tile + ((row0 + r) * G::PITCH + k0 + 8 * hh) * 2);
    } else {
        // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(row) block; lane 4q+p supplies the address of
        // block row q, columns 4p..4p+3; lane i receives column i of the 4 rows (cdna_hip_programming T10).
        const int g16 = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const int mb = row0 + 16 * (g16 & 1);
        const int kb = k0 + 8 * (g16 >> 1);
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const char* a0 = tile + ((kb + q) * G::PITCH + mb + 4 * p) * 2;
        const char* a1 = a0 + 4 * G::PITCH * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    }
}

// f32: element s of the result is k = k0 + 8*(lane>>5) + s of tile row (row0 + (lane&31))
template <bool KC, int ROWS>
__device__ __forceinline__ void frag_f32(const char* tile, int row0, int lane, float (&out)[8]) {
    using G = TileGeom<0, KC, ROWS>;
    const int r = lane & 31, hh = lane >> 5;
    const float* t = reinterpret_cast<const float*>(tile);
    if constexpr (KC) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(t + (row0 + r) * G::PITCH + 8 * hh);
        const f32x4 b = *reinterpret_cast<const f32x4*>(t + (row0 + r) * G::PITCH + 8 * hh + 4);
        out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; out[3] = a[3];
        out[4] = b[0]; out[5] = b[1]; out[6] = b[2]; out[7] = b[3];
    } else {
#pragma unroll
        for (int s = 0; s < 8; s++) out[s] = t[(8 * hh + s) * G::PITCH + row0 + r];
    }
}

// MODE: 0 plain store, 1 read-modify-write accumulate, 2 f32 atomicAdd
template <typename TC, int MODE> __device__ __forceinline__ void c_store(TC* p, float v) {
    if constexpr (MODE == 2) {
        if constexpr (sizeof(TC) == 4) atomicAdd(reinterpret_cast<float*>(p), v);
    } else if constexpr (MODE == 1) {
        stf(p, ldf(p) + v);
    } else {
        stf(p, v);
    }
}

template <typename TC, int WM, int WN, bool FULL, int MODE>
__device__ __forceinline__ void epilogue(const GemmArgs& g, TC* C, const TC* R, f32x16 (&acc)[WM][WN], int row_base,
                                         int col_base, int lane, bool lead) {
    // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int r = lane & 31, hh = lane >> 5;
    const float diag = lead ? g.diag : 0.f;
#pragma unroll
    for (int j = 0; j < WN; j++) {
        const int col = col_base + j * 32 + r;
        const bool col_ok = FULL || col < g.N;
        const float bias = (g.bias && lead && col_ok) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; i++) {
            const int rbase = row_base + i * 32 + 4 * hh;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = rbase + (reg & 3) + 8 * (reg >> 2);
                float v = g.alpha * acc[i][j][reg] + bias;
                if (row == col) v += diag;
                if (FULL || (col_ok && row < g.M)) {
                    if (R) v += g.rcoef * ldf(R + (long)row * g.ldc + col);
                    if (g.act == MH_ACT_RELU) v = fmaxf(v, 0.f);
                    c_store<TC, MODE>(C + (long)row * g.ldc + col, v);
                }
            }
        }
    }
}

// FULL tiles, plain or read-modify-write output: the accumulators go through LDS (f32 [BM][BN+4]) and leave as
// 16-B row-contiguous stores (4 rows x 256/512 B per wave-instruction).  The direct path issues 64 narrow stores
// per lane (2-byte ones for bf16 C), which made the epilogue longer than the K=512 main loop.
template <typename TC, int WM, int WN, int MODE>
__device__ __forceinline__ void epilogue_lds(const GemmArgs& g, TC* C, const TC* R, f32x16 (&acc)[WM][WN], char* smem, int tile_row0,
                                             int tile_col0, int wm, int wn, int lane, int tid, bool lead) {
    constexpr int BM = 64 * WM, BN = 64 * WN, PITCH = BN + 4;
    float* t = reinterpret_cast<float*>(smem);
    const int r = lane & 31, hh = lane >> 5;
    const float diag = lead ? g.diag : 0.f;
#pragma unroll
    for (int j = 0; j < WN; j++) {
        const int lc = wn * WN * 32 + j * 32 + r;
        const int col = tile_col0 + lc;
        const float bias = (g.bias && lead) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; i++) {
            const int lr0 = wm * WM * 32 + i * 32 + 4 * hh;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int lr = lr0 + (reg & 3) + 8 * (reg >> 2);
                float v = g.alpha * acc[i][j][reg] + bias;
                if (tile_row0 + lr == col) v += diag;
                if (R && tile_row0 + lr < g.M) v += g.rcoef * ldf(R + (long)(tile_row0 + lr) * g.ldc + col);
                if (g.act == MH_ACT_RELU) v = fmaxf(v, 0.f);
                t[lr * PITCH + lc] = v;
            }
        }
    }
    __syncthreads();
    constexpr int EPC = 16 / (int)sizeof(TC);          // elements per 16-B chunk: 4 (f32) or 8 (bf16)
    constexpr int CPR = BN / EPC;                      // chunks per tile row
    constexpr int NCH = BM * CPR / 256;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int cid = tid + i * 256;
        const int lr = cid / CPR, c = cid % CPR;
        const float* src = t + lr * PITCH + c * EPC;
        if (tile_row0 + lr >= g.M) continue;           // ragged last row-tile (K-contiguous A only)
        TC* dst = C + (long)(tile_row0 + lr) * g.ldc + tile_col0 + c * EPC;
        f32x4 x0 = *reinterpret_cast<const f32x4*>(src);
        u32x4 o;
        if constexpr (sizeof(TC) == 4) {
            if constexpr (MODE == 1) x0 += *reinterpret_cast<const f32x4*>(dst);
            o[0] = __float_as_uint(x0[0]); o[1] = __float_as_uint(x0[1]);
            o[2] = __float_as_uint(x0[2]); o[3] = __float_as_uint(x0[3]);
        } else {
            f32x4 x1 = *reinterpret_cast<const f32x4*>(src + 4);
            if constexpr (MODE == 1) {
                const u32x4 old = *reinterpret_cast<const u32x4*>(dst);
                x0[0] += __uint_as_float(old[0] << 16); x0[1] += __uint_as_float(old[0] & 0xffff0000u);
                x0[2] += __uint_as_float(old[1] << 16); x0[3] += __uint_as_float(old[1] & 0xffff0000u);
                x1[0] += __uint_as_float(old[2] << 16); x1[1] += __uint_as_float(old[2] & 0xffff0000u);
                x1[2] += __uint_as_float(old[3] << 16); x1[3] += __uint_as_float(old[3] & 0xffff0000u);
            }
            o[0] = pack_bf2(x0[0], x0[1]);
            o[1] = pack_bf2(x0[2], x0[3]);
            o[2] = pack_bf2(x1[0], x1[1]);
            o[3] = pack_bf2(x1[2], x1[3]);
        }
        *reinterpret_cast<u32x4*>(dst) = o;
    }
}

template <int MMA, typename TA, typename TB, typename TC, bool AKC, bool BKC, int WM, int WN, int FULL>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    using GA = TileGeom<MMA, AKC, BM>;
    using GB = TileGeom<MMA, BKC, BN>;
    constexpr int BK = GA::BK;
    __shared__ __attribute__((aligned(16))) char smem[2 * (GA::BYTES + GB::BYTES)];
    constexpr int STAGE = GA::BYTES + GB::BYTES;  // stage s: A at smem + s*STAGE, B right behind it

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware remap (bijective form, cdna_hip_programming §5): blocks b and b+8 share an XCD, so give each
    // of the 8 groups a contiguous run of tile ids -> neighbouring tiles (same A row-panel) hit one L2.
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tile_m = wgid / g.tiles_n, tile_n = wgid % g.tiles_n;
    const int z = blockIdx.z;
    const int b1 = z / g.batch2, b2 = z % g.batch2;
    const TA* A = reinterpret_cast<const TA*>(g.A) + b1 * g.sA1 + b2 * g.sA2;
    const TB* B = reinterpret_cast<const TB*>(g.B) + b1 * g.sB1 + b2 * g.sB2;
    TC* C = reinterpret_cast<TC*>(g.C) + b1 * g.sC1 + b2 * g.sC2;
    const TC* R = g.R ? reinterpret_cast<const TC*>(g.R) + b1 * g.sC1 + b2 * g.sC2 : nullptr;
    const int split = blockIdx.y;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nt = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // FULL == 3: whole tiles in M and K, N ends inside the (single) column tile: only B's loads and the stores know about it
    constexpr bool WHOLE = FULL == 1 || FULL == 2;
    using SA = Stager<MMA, TA, AKC, BM, FULL == 3 ? 1 : FULL>;
    using SB = Stager<MMA, TB, BKC, BN, FULL>;
    u32x4 ra[SA::NCH], rb[SB::NCH];
    // One register set: K-tile t+1 is written to LDS right AFTER the barrier that frees its stage and tile t+2 is
    // requested at once, so the ds_write pass drains under the MFMAs of tile t (not between the last MFMA and the barrier).
    if (nt > 0) {
        SA::load(ra, A, g.lda, tile_m * BM, g.M, kbeg, kend, g.vecA, tid);
        SB::load(rb, B, g.ldb, tile_n * BN, g.N, kbeg, kend, g.vecB, tid);
        SA::store(ra, smem, tid);
        SB::store(rb, smem + GA::BYTES, tid);
        if (nt > 1) {
            SA::load(ra, A, g.lda, tile_m * BM, g.M, kbeg + BK, kend, g.vecA, tid);
            SB::load(rb, B, g.ldb, tile_n * BN, g.N, kbeg + BK, kend, g.vecB, tid);
        }
    }
    __syncthreads();

    for (int t = 0; t < nt; t++) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            SA::store(ra, smem + (cur ^ 1) * STAGE, tid);
            SB::store(rb, smem + (cur ^ 1) * STAGE + GA::BYTES, tid);
            if (t + 2 < nt) {
                const int k0 = kbeg + (t + 2) * BK;
                SA::load(ra, A, g.lda, tile_m * BM, g.M, k0, kend, g.vecA, tid);
                SB::load(rb, B, g.ldb, tile_n * BN, g.N, k0, kend, g.vecB, tid);
            }
        }
        const char* at = smem + cur * STAGE;
        const char* bt = at + GA::BYTES;
        if constexpr (MMA) {
#pragma unroll
            for (int ks = 0; ks < BK; ks += 16) {
                bf16x8 af[WM], bfr[WN];
#pragma unroll
                for (int i = 0; i < WM; i++) af[i] = frag_bf16<AKC, BM>(at, wm * WM * 32 + i * 32, ks, lane);
#pragma unroll
                for (int j = 0; j < WN; j++) bfr[j] = frag_bf16<BKC, BN>(bt, wn * WN * 32 + j * 32, ks, lane);
#pragma unroll
                for (int i = 0; i < WM; i++)
#pragma unroll
                    for (int j = 0; j < WN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        } else {
            float af[WM][8], bfr[WN][8];
#pragma unroll
            for (int i = 0; i < WM; i++) frag_f32<AKC, BM>(at, wm * WM * 32 + i * 32, lane, af[i]);
#pragma unroll
            for (int j = 0; j < WN; j++) frag_f32<BKC, BN>(bt, wn * WN * 32 + j * 32, lane, bfr[j]);
#pragma unroll
            for (int s = 0; s < 8; s++)
#pragma unroll
                for (int i = 0; i < WM; i++)
#pragma unroll
                    for (int j = 0; j < WN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bfr[j][s], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    const int row_base = tile_m * BM + wm * WM * 32, col_base = tile_n * BN + wn * WN * 32;
    const bool lead = (split == 0);
    if constexpr (WHOLE && MMA == 1) {
        static_assert(BM * (BN + 4) * 4 <= 2 * STAGE, "epilogue tile must fit the staging LDS");
        if (g.vecC && !g.atomic) {
            if (g.accumulate) epilogue_lds<TC, WM, WN, 1>(g, C, R, acc, smem, tile_m * BM, tile_n * BN, wm, wn, lane, tid, lead);
            else epilogue_lds<TC, WM, WN, 0>(g, C, R, acc, smem, tile_m * BM, tile_n * BN, wm, wn, lane, tid, lead);
            return;
        }
    }
    if (WHOLE && tile_m * BM + BM <= g.M) {   // interior tile: unguarded stores
        if (g.atomic) epilogue<TC, WM, WN, true, 2>(g, C, R, acc, row_base, col_base, lane, lead);
        else if (g.accumulate) epilogue<TC, WM, WN, true, 1>(g, C, R, acc, row_base, col_base, lane, lead);
        else epilogue<TC, WM, WN, true, 0>(g, C, R, acc, row_base, col_base, lane, lead);
    } else {
        if (g.atomic) epilogue<TC, WM, WN, false, 2>(g, C, R, acc, row_base, col_base, lane, lead);
        else if (g.accumulate) epilogue<TC, WM, WN, false, 1>(g, C, R, acc, row_base, col_base, lane, lead);
        else epilogue<TC, WM, WN, false, 0>(g, C, R, acc, row_base, col_base, lane, lead);
    }
}

// ------------------------------------------------------------------ host dispatch (per family)
template <int MMA, typename TA, typename TB, typename TC, bool AKC, bool BKC>
static void launch_w(GemmArgs& a, int batch, hipStream_t s) {
    constexpr int BK = MMA ? 64 : 16;
    // 128 x 64 tiles for N <= 64 -- and for launches that would leave more than a third of the CUs without a 128 x 128 tile
    // (the row remainders of the big-tile split: 4096 x 512 is 128 tiles, 256 half tiles finish in ~60 % of the time)
    constexpr bool fill = true;
    const long wide_wgs = (long)mh_cdiv(a.M, 128) * mh_cdiv(a.N, 128) * a.split_k * batch;
    const bool narrow = a.N <= 64 || (fill && MMA == 1 && wide_wgs <= 160 && a.N % 64 == 0 && a.M >= 1024);
    const int BN = narrow ? 64 : 128;
    a.tiles_m = mh_cdiv(a.M, 128);
    a.tiles_n = mh_cdiv(a.N, BN);
    // a ragged M is fine when A's rows are K-contiguous (loads clamp to the last row, stores are guarded)
    const bool mn_ok = a.vecA && a.vecB && (AKC || a.M % 128 == 0) && a.N % BN == 0;
    const bool full = mn_ok && a.K % BK == 0 && a.k_per_split % BK == 0 && a.K % a.k_per_split == 0;
    // whole tiles in M and N, one ragged K-tile at the end (no split-K): the FULL == 2 instances (bf16 MMA only: the batched
    // dh = 96 products of the template geometry)
    constexpr int VMAX = (sizeof(TA) == 2 || sizeof(TB) == 2) ? 8 : 4;
    const bool ktail = MMA == 1 && mn_ok && !full && a.split_k == 1 && a.K % VMAX == 0 && a.K > BK && a.M % 128 == 0;
    // N ends inside the one column tile (64 < N < 128, N % 8 == 0), everything else whole: FULL == 3
    const bool ntail = MMA == 1 && !narrow && a.vecA && a.vecB && a.M % 128 == 0 && a.N % BN != 0 && a.N < BN && a.N % VMAX == 0 &&
                       a.K % BK == 0 && a.k_per_split % BK == 0 && a.K % a.k_per_split == 0;
    dim3 grid(a.tiles_m * a.tiles_n, a.split_k, batch);
    gemm_note_variant("gemm_kernel<%d,%s,%s,%s,%s,%s,2,%d,%d>", MMA, gemm_tn<TA>(), gemm_tn<TB>(), MMA ? gemm_tn<TC>() : "float", gemm_tf(AKC), gemm_tf(BKC),
                      narrow ? 1 : 2, full ? 1 : ((!narrow && MMA == 1 && sizeof(TA) == 2 && sizeof(TB) == 2) ? (ktail ? 2 : (ntail ? 3 : 0)) : 0));
    if (narrow) {
        if (full) hipLaunchKernelGGL((gemm_kernel<MMA, TA, TB, TC, AKC, BKC, 2, 1, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gemm_kernel<MMA, TA, TB, TC, AKC, BKC, 2, 1, 0>), grid, dim3(256), 0, s, a);
    } else {
        if (full) hipLaunchKernelGGL((gemm_kernel<MMA, TA, TB, TC, AKC, BKC, 2, 2, 1>), grid, dim3(256), 0, s, a);
        else if constexpr (MMA == 1 && sizeof(TA) == 2 && sizeof(TB) == 2) {
            if (ktail) hipLaunchKernelGGL((gemm_kernel<MMA, TA, TB, TC, AKC, BKC, 2, 2, 2>), grid, dim3(256), 0, s, a);
            else if (ntail) hipLaunchKernelGGL((gemm_kernel<MMA, TA, TB, TC, AKC, BKC, 2, 2, 3>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gemm_kernel<MMA, TA, TB, TC, AKC, BKC, 2, 2, 0>), grid, dim3(256), 0, s, a);
        } else hipLaunchKernelGGL((gemm_kernel<MMA, TA, TB, TC, AKC, BKC, 2, 2, 0>), grid, dim3(256), 0, s, a);
    }
}

template <int MMA, typename TA, typename TB, typename TC>
static void launch_l(GemmArgs& a, int akc, int bkc, int batch, hipStream_t s) {
    if (akc && bkc) launch_w<MMA, TA, TB, TC, true, true>(a, batch, s);
    else if (akc && !bkc) launch_w<MMA, TA, TB, TC, true, false>(a, batch, s);
    else if (!akc && bkc) launch_w<MMA, TA, TB, TC, false, true>(a, batch, s);
    else launch_w<MMA, TA, TB, TC, false, false>(a, batch, s);
}

// family entry points (one translation unit each)
void gemm_launch_f32(GemmArgs& a, int akc, int bkc, int batch, hipStream_t s);                 // MMA f32
void gemm_launch_bf16(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s);       // bf16 operands
void gemm_launch_mixed_ff(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s);   // f32 x f32 operands, bf16 MMA
void gemm_launch_mixed_fb(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s);   // f32 A, bf16 B
void gemm_launch_mixed_bf(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s);   // bf16 A, f32 B
