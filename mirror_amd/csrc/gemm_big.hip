// Large-tile variant of the bf16 MFMA GEMM: 256 x 256 x 64 block tile, 512 threads = 8 waves (2 x 4), wave tile
// 128 x 64 (4 x 2 MFMA blocks, 128 accumulator registers), two waves per SIMD.
//
// Why: the projection / weight-gradient GEMMs of the step are long and thin (M = 69632, K = 512..1536).  With 128 x 128
// tiles every MFMA flop pulls (BM + BN) / (BM BN) = 1/64 byte through L2 -> LDS, i.e. ~39 TB/s at the 2.5 PFLOP/s
// peak — more than the fabric delivers; 256 x 256 tiles halve that, and halve the LDS fragment reads per MFMA as well
// (6 fragment reads feed 8 MFMAs instead of 4 feeding 4).  Same staging scheme as gemm_kernel (global -> registers
// one K-tile ahead -> double-buffered LDS, ds_read_b64_tr_b16 for K-strided operands), 144 KiB of LDS.
// Used by gemm_launch_bf16 when N is a multiple of 256, K of 64, M a multiple of 256 (or ragged with K-contiguous A rows:
// loads clamp to the last row, stores are guarded) and the operands are 16-byte aligned.
#include "gemm_kernel.h"
#ifndef GEMM_EXP
#define GEMM_EXP 0
#endif

// the tail mh_gemm offers to the next fold launch (gemm_kernel.h: GemmTail); host-side state of the calling thread's launch sequence
static thread_local GemmTail g_tail = {nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0.f};
GemmTail* gemm_pending_tail() { return &g_tail; }

namespace {

constexpr int BIG = 256, NTB = 512, BWM = 4, BWN = 2;

// Fused epilogues (mh_gemm_epi, include/mirror_hip.h) on one quad of a row: x = the Linear's result for columns gcol .. gcol + 3
// of flat row grow.  The result is rounded to bf16 first: that is what the composed path hands the elementwise op.
__device__ __forceinline__ f32x4 round_bf16_4(f32x4 x) {
    return f32x4{bf2f(f2bf(x[0])), bf2f(f2bf(x[1])), bf2f(f2bf(x[2])), bf2f(f2bf(x[3]))};
}
// (b0, t0) = batch and row-in-batch of the TILE's first row, computed once per tile: a 256-row tile crosses at most one batch edge
// (rows_per_batch > 256 is checked by the host), so a quad's (b, t) is an add and a compare — as a 32-bit division per quad (no integer
// divide instruction: ~25 VALU each, 32 quads per thread and tile) the retention_embed launch ran 84 us against 57 for the plain product
struct TileRow { int b0, t0, row0; };
template <int EPI>
__device__ __forceinline__ f32x4 epi_quad(const GemmArgs& g, f32x4 x, int grow, int gcol, const TileRow& tr) {
    if constexpr (EPI == MH_EPI_MASKPOS) {
        // random_masking's token select + `+ retention_gene_embed` (models/mirror.py:636-643, :691-693)
        x = round_bf16_4(x);
        const int rpb = g.epi.rows_per_batch, first = g.epi.first;
        int t = tr.t0 + (grow - tr.row0), b = tr.b0;
        if (t >= rpb) { t -= rpb; b += 1; }
        if (t >= first && g.epi.mask[(long)b * (rpb - first) + (t - first)] != 0.f) x = *reinterpret_cast<const f32x4*>(g.epi.token + gcol);
        return x + *reinterpret_cast<const f32x4*>(g.epi.pos + (long)t * g.N + gcol);
    } else {
        return x;
    }
}

// accumulators -> f32 LDS tile [128][260] (one half of the rows at a time) -> 16-byte row-contiguous stores
// physical row of C for flat result row r (GemmArgs.c_rpb / c_skip: row windows of larger batches, see mh_gemm_desc.c_rows_per_batch)
__device__ __forceinline__ long c_phys_row(const GemmArgs& g, int r) { return g.c_rpb > 0 ? (long)r + (long)min(r / g.c_rpb, g.w_last) * g.c_skip : (long)r; }

template <typename TC, int MODE, int EPI = 0>
__device__ __forceinline__ void epilogue_big(const GemmArgs& g, TC* C, f32x16 (&acc)[BWM][BWN], char* smem, int tile_row0,
                                             int tile_col0, int wm, int wn, int lane, int tid, bool lead, long ldc, float alpha) {
    uint64_t drop_blk0 = 0;
    uint32_t thr = 0;
    float dscale = 1.f;
    if constexpr (EPI == MH_EPI_DROPADD) {
        uint64_t off = g.epi.offset;
        if (g.epi.dev_base) off += *g.epi.dev_base & ~7ull;
        drop_blk0 = off >> 3;
        thr = drop16_thr(g.epi.p);
        dscale = drop16_scale(thr);
    }
    constexpr int PITCH = BIG + 4, HALF = BIG / 2;
    float* t = reinterpret_cast<float*>(smem);
    const int r = lane & 31, hh = lane >> 5;
    TileRow trow{0, 0, tile_row0};
    if constexpr (EPI == MH_EPI_MASKPOS) { trow.b0 = tile_row0 / g.epi.rows_per_batch; trow.t0 = tile_row0 - trow.b0 * g.epi.rows_per_batch; }
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if (wm == half) {
#pragma unroll
            for (int j = 0; j < BWN; j++) {
                const int lc = wn * BWN * 32 + j * 32 + r;
                const float bias = (g.bias && lead) ? g.bias[tile_col0 + lc] : 0.f;
#pragma unroll
                for (int i = 0; i < BWM; i++) {
                    const int lr0 = i * 32 + 4 * hh;
#pragma unroll
                    for (int reg = 0; reg < 16; reg++) {
                        const int lr = lr0 + (reg & 3) + 8 * (reg >> 2);
                        float v = alpha * acc[i][j][reg] + bias;
                        if (g.act == MH_ACT_RELU) v = fmaxf(v, 0.f);
                        t[lr * PITCH + lc] = v;
                    }
                }
            }
        }
        __syncthreads();
        constexpr int EPC = 16 / (int)sizeof(TC);
        constexpr int CPR = BIG / EPC;
        constexpr int NCH = HALF * CPR / NTB;
        if constexpr (EPI == MH_EPI_DROPADD) {
            // [3P] to_out[1] = Dropout, then TransLayer's residual add (models/mirror.py:312-313): 8 columns per thread = one block
            // of the lite dropout stream (common.h), the mask mh_dropout_lite draws for the same (seed, offset, element)
            constexpr int CPR8 = BIG / 8, NCH8 = HALF * CPR8 / NTB;
#pragma unroll 2
            for (int i = 0; i < NCH8; i++) {
                const int cid = tid + i * NTB;
                const int lr = cid / CPR8, c = cid % CPR8;
                const int grow = tile_row0 + half * HALF + lr, gcol = tile_col0 + c * 8;
                if (grow >= g.M) continue;
                const float* src = t + lr * PITCH + c * 8;
                const float* rp = g.epi.resid + (long)grow * g.N + gcol;
                f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
                const f32x4 x0 = round_bf16_4(*reinterpret_cast<const f32x4*>(src)), x1 = round_bf16_4(*reinterpret_cast<const f32x4*>(src + 4));
                const uint32_t keep = drop16_keep8(drop_blk0 + (((uint64_t)grow * (uint64_t)g.N + (uint64_t)gcol) >> 3), g.epi.seed, thr);
#pragma unroll
                for (int e = 0; e < 4; e++) {      // two roundings (scale, then add), as mh_dropout_lite: no fused multiply-add
                    r0[e] = __fadd_rn(r0[e], (keep & (1u << e)) ? __fmul_rn(x0[e], dscale) : 0.f);
                    r1[e] = __fadd_rn(r1[e], (keep & (16u << e)) ? __fmul_rn(x1[e], dscale) : 0.f);
                }
                float* dst = reinterpret_cast<float*>(C) + (long)grow * ldc + gcol;
                *reinterpret_cast<f32x4*>(dst) = r0;
                *reinterpret_cast<f32x4*>(dst + 4) = r1;
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int cid = tid + i * NTB;
            const int lr = cid / CPR, c = cid % CPR;
            const float* src = t + lr * PITCH + c * EPC;
            if (tile_row0 + half * HALF + lr >= g.M) continue;      // ragged last row tile (K-contiguous A only)
            TC* dst = C + (long)(tile_row0 + half * HALF + lr) * ldc + tile_col0 + c * EPC;
            f32x4 x0 = *reinterpret_cast<const f32x4*>(src);
            u32x4 o;
            if constexpr (sizeof(TC) == 4) {
                if constexpr (EPI != 0) x0 = epi_quad<EPI>(g, x0, tile_row0 + half * HALF + lr, tile_col0 + c * EPC, trow);
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
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst));      // pq_store_note
        }
        __syncthreads();
    }
}

// split-K / batch-broadcast: f32 atomicAdd straight from the accumulators (lanes = 32 consecutive columns: coalesced);
// rows past M (ragged last tile) are skipped
__device__ __forceinline__ void epilogue_atomic_big(const GemmArgs& g, float* C, f32x16 (&acc)[BWM][BWN], int row0, int col0, int lane) {
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < BWM; i++) {
        const int rb = row0 + 32 * i + 4 * hh;
        float* base = C + (long)rb * g.ldc + col0 + r;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int dr = (reg & 3) + 8 * (reg >> 2);
            if (rb + dr >= g.M) continue;
            float* p = base + (long)dr * g.ldc;
#pragma unroll
            for (int j = 0; j < BWN; j++) atomicAdd(p + 32 * j, g.alpha * acc[i][j][reg]);
        }
    }
}

// bf16 output: the accumulators hold C^T (the MFMA takes the B fragment as its A operand), so a lane owns 4 consecutive
// columns of one row: the whole 256 x 256 tile goes to LDS as bf16 with 8-byte stores (one pass, half the bytes of the
// f32 image and no 4-byte scatter) and leaves as 16-byte row-contiguous stores.  Pitch 260: 16 lanes (rows) x 2 dwords
// cover the 32 banks exactly.
template <int MODE, int EPI = 0>
__device__ __forceinline__ void epilogue_big_t(const GemmArgs& g, bf16_t* C, f32x16 (&acc)[BWM][BWN], char* smem, int tile_row0,
                                               int tile_col0, int wm, int wn, int lane, int tid, bool lead, float alpha) {
    constexpr int PITCH = BIG + 4;
    float sq_sum = 0.f, sq_cnt = 0.f;          // MH_EPI_SQERR partials of this thread
    static_assert(BIG * PITCH * 2 <= 2 * (TileGeom<1, true, BIG>::BYTES + TileGeom<1, true, BIG>::BYTES), "bf16 tile must fit");
    bf16_t* t = reinterpret_cast<bf16_t*>(smem);
    const int r = lane & 31, hh = lane >> 5;
    const bool has_bias = g.bias && lead;
#pragma unroll
    for (int j = 0; j < BWN; j++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            const int lc = wn * BWN * 32 + 32 * j + 8 * gq + 4 * hh;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (has_bias) bv = *reinterpret_cast<const f32x4*>(g.bias + tile_col0 + lc);
#pragma unroll
            for (int i = 0; i < BWM; i++) {
                const int lr = wm * BWM * 32 + 32 * i + r;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] = alpha * acc[i][j][4 * gq + e] + bv[e];
                    if (g.act == MH_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                }
                u32x2 o;
                o[0] = pack_bf2(v[0], v[1]);
                o[1] = pack_bf2(v[2], v[3]);
                *reinterpret_cast<u32x2*>(t + lr * PITCH + lc) = o;
            }
        }
    __syncthreads();
    constexpr int CPR = BIG / 8;                 // 16-byte chunks per tile row
    constexpr int NCH = BIG * CPR / NTB;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int cid = tid + i * NTB;
        const int lr = cid / CPR, c = cid % CPR;
        if (tile_row0 + lr >= g.M) continue;      // ragged last row tile (K-contiguous A only)
        bf16_t* dst = C + c_phys_row(g, tile_row0 + lr) * g.ldc + tile_col0 + c * 8;
        u32x2 lo = *reinterpret_cast<const u32x2*>(t + lr * PITCH + c * 8);
        u32x2 hi = *reinterpret_cast<const u32x2*>(t + lr * PITCH + c * 8 + 4);
        u32x4 o = {lo[0], lo[1], hi[0], hi[1]};
        if constexpr (MODE == 1) {
            const u32x4 old = *reinterpret_cast<const u32x4*>(dst);
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const float a0 = __uint_as_float(o[w] << 16) + __uint_as_float(old[w] << 16);
                const float a1 = __uint_as_float(o[w] & 0xffff0000u) + __uint_as_float(old[w] & 0xffff0000u);
                o[w] = pack_bf2(a0, a1);
            }
        }
        if (GEMM_EXP == 5) { if (o[0] == 0x12345678u) *reinterpret_cast<u32x4*>(dst) = o; continue; }
        __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst));      // pq_store_note
        if constexpr (EPI == MH_EPI_SQERR) {
            // masked squared error against the f32 target rows (losses/mirror_loss.py:98-103) on the bf16-rounded prediction
            const int rpb = g.epi.rows_per_batch;             // % 256 == 0: a tile lies inside one batch
            const long b = tile_row0 / rpb;
            const int t = tile_row0 - (int)(b * rpb) + lr;
            if (g.epi.mask[b * rpb + t] != 0.f) {
                const float* tg = g.epi.tgt + b * g.epi.tgt_bs + (long)t * g.N + tile_col0 + c * 8;
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(tg), t1 = *reinterpret_cast<const f32x4*>(tg + 4);
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const float d0 = __uint_as_float(o[w] << 16) - (w < 2 ? t0[2 * w] : t1[2 * w - 4]);
                    const float d1 = __uint_as_float(o[w] & 0xffff0000u) - (w < 2 ? t0[2 * w + 1] : t1[2 * w - 3]);
                    sq_sum += d0 * d0 + d1 * d1;
                }
                sq_cnt += 8.f;
            }
        }
    }
    if constexpr (EPI == MH_EPI_SQERR) {
        float* red = reinterpret_cast<float*>(smem + BIG * PITCH * 2);     // behind the bf16 tile image (133120 of 147456 bytes)
        sq_sum = wave_sum(sq_sum);
        sq_cnt = wave_sum(sq_cnt);
        if (lane == 0) { red[2 * (tid >> 6)] = sq_sum; red[2 * (tid >> 6) + 1] = sq_cnt; }
        __syncthreads();
        if (tid == 0) {
            float a = 0.f, n = 0.f;
#pragma unroll
            for (int w = 0; w < NTB / 64; w++) { a += red[2 * w]; n += red[2 * w + 1]; }
            const float inv = 1.f / (float)g.N;          // mh_mse_masked_fwd's convention: row means over D, rows counted once
            atomicAdd(g.epi.sq, a * inv);
            atomicAdd(g.epi.sq + 1, n * inv);
        }
    }
}

// FP8 instance (BASELINE config 5): the operands are e4m3 bytes with K contiguous, described to the staging code as bf16 rows
// of half the length (same 128-byte K-tile rows, same LDS image); a lane then owns 32 consecutive bytes of a 64-byte k-step
// and the product is v_mfma_scale_f32_32x32x64_f8f6f4 (unit scales) — 16 instead of 32 MFMAs per K-tile at twice the K each.
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x8 f8_pair(bf16x8 lo, bf16x8 hi) {
    const i32x4f a = __builtin_bit_cast(i32x4f, lo), b = __builtin_bit_cast(i32x4f, hi);
    return i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// A tile load for the fused-epilogue instances: K-contiguous A whose M rows are row windows of larger batches (GemmArgs.a_rpb /
// a_skip: `to_out(out)[:, -n:]`, `retention_head(x)[:, 1:]`): a 256-row tile crosses at most one batch boundary (a_rpb >= 256),
// at local row `bnd`; rows past M re-read row M - 1 (their results are never stored).
template <int NCH>
__device__ __forceinline__ void load_a_window(u32x4 (&regs)[NCH], const bf16_t* __restrict__ base, long ld, int tile0, int dim, int k0,
                                              int tid, int adj0, int bnd, int skip) {
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int cid = tid + i * NTB;
        const int rl = min(tile0 + (cid >> 3), dim - 1) - tile0, c = cid & 7;
        const int row = tile0 + rl + adj0 + (rl >= bnd ? skip : 0);
        regs[i] = *reinterpret_cast<const u32x4*>(base + (long)row * ld + k0 + c * 8);
    }
}

template <typename TC, bool AKC, bool BKC, bool FP8 = false, bool PART = false, int EPI = 0>
__global__ __launch_bounds__(NTB) void gemm_big_kernel(GemmArgs g) {
    static_assert(!FP8 || (AKC && BKC), "fp8 operands are K-contiguous");
    static_assert(EPI == 0 || (AKC && !FP8 && !PART), "fused epilogues: K-contiguous A, plain bf16 operands, no split-K");
    using GA = TileGeom<1, AKC, BIG>;
    using GB = TileGeom<1, BKC, BIG>;
    constexpr int BK = 64;
    constexpr int STAGE = GA::BYTES + GB::BYTES;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    static_assert((BIG / 2) * (BIG + 4) * 4 <= 2 * STAGE, "epilogue half tile must fit the staging LDS");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    // XCD-aware order over the WHOLE grid (tiles x K-slices x batches), not just over the tiles of one slice: workgroups are dealt
    // round-robin over the 8 XCDs (private L2s), so linear id L runs on XCD L % 8; unit u = (L % 8) * (total / 8) + L / 8 gives
    // each XCD a contiguous range of units, and unit -> (batch, K-slice, tile) with the tile fastest puts all M x N tiles of one
    // K-slice on ONE XCD at about the same time: a weight-gradient slice's A and B panels are fetched from HBM once per XCD
    // instead of once per tile (513 -> 260 MB per launch on the step's nine weight gradients).
    const int tiles = gridDim.x, nwg = tiles * gridDim.y * gridDim.z;
    const int lin = blockIdx.x + tiles * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = lin & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    const int wgid = unit % tiles, slice = unit / tiles;
    const int tile_m = wgid / g.tiles_n, tile_n = wgid % g.tiles_n;
    const int z = slice / (int)gridDim.y;
    const int b1 = z / g.batch2, b2 = z % g.batch2;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + b1 * g.sA1 + b2 * g.sA2;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B) + b1 * g.sB1 + b2 * g.sB2;
    TC* C = reinterpret_cast<TC*>(g.C) + b1 * g.sC1 + b2 * g.sC2;
    const int split = slice % (int)gridDim.y;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nt = (kend - kbeg) / BK;

    f32x16 acc[BWM][BWN];
#pragma unroll
    for (int i = 0; i < BWM; i++)
#pragma unroll
        for (int j = 0; j < BWN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    using SA = Stager<1, bf16_t, AKC, BIG, true, NTB>;
    using SB = Stager<1, bf16_t, BKC, BIG, true, NTB>;
    u32x4 ra[SA::NCH], rb[SB::NCH];
    // One register set, written to LDS right AFTER the barrier that frees the other stage and re-issued at once: the
    // ds_write pass (~80 B/clk per CU, ~800 cycles per K-tile) then drains under the MFMAs of the current tile instead
    // of sitting between the last MFMA and the barrier (8192^3: 732 -> 1094 TFLOP/s).
    int adj0 = 0, bnd = 1 << 30;
    if constexpr (AKC) {
        if (g.a_rpb > 0) {
            const int b0 = min((tile_m * BIG) / g.a_rpb, g.w_last);
            adj0 = b0 * g.a_skip;
            bnd = b0 < g.w_last ? (b0 + 1) * g.a_rpb - tile_m * BIG : (1 << 30);
        }
    }
#define LOAD_A_(K0_)                                                                                         \
    do {                                                                                                     \
        if constexpr (EPI != 0) load_a_window<SA::NCH>(ra, A, g.lda, tile_m * BIG, g.M, K0_, tid, adj0, bnd, g.a_skip); \
        else SA::load(ra, A, g.lda, tile_m * BIG, g.M, K0_, kend, true, tid);                                \
    } while (0)
    if (nt > 0) {
        LOAD_A_(kbeg);
        SB::load(rb, B, g.ldb, tile_n * BIG, g.N, kbeg, kend, true, tid);
        SA::store(ra, smem, tid);
        SB::store(rb, smem + GA::BYTES, tid);
        if (nt > 1) {
            LOAD_A_(kbeg + BK);
            SB::load(rb, B, g.ldb, tile_n * BIG, g.N, kbeg + BK, kend, true, tid);
        }
    }
    __syncthreads();
    for (int t = 0; t < nt; t++) {
        const int cur = t & 1;
#ifndef GEMM_EXP
#define GEMM_EXP 0
#endif
        if (t + 1 < nt) {
            if (GEMM_EXP != 1 && GEMM_EXP != 3) {     // timing experiments (tools/exp/gemm_exp.cpp): 1 = no loads / stores, 3 = no LDS stores
                SA::store(ra, smem + (cur ^ 1) * STAGE, tid);
                SB::store(rb, smem + (cur ^ 1) * STAGE + GA::BYTES, tid);
            }
            if (t + 2 < nt && GEMM_EXP != 1) {
                const int k0 = GEMM_EXP == 2 ? kbeg : kbeg + (t + 2) * BK;    // 2 = always the same K-tile (cache resident)
                LOAD_A_(k0);
                SB::load(rb, B, g.ldb, tile_n * BIG, g.N, k0, kend, true, tid);
            }
        }
        const char* at = smem + cur * STAGE;
        const char* bt = at + GA::BYTES;
        if constexpr (FP8) {
#pragma unroll
            for (int s8 = 0; s8 < 2; s8++) {      // two 64-byte k-steps per 128-byte tile row; lane half hl owns bytes [32 hl, 32 hl + 32)
                const int ks = 32 * s8 + 8 * (lane >> 5);      // in 2-byte units; frag_bf16 adds another 8 (lane >> 5)
                i32x8 af[BWM], bfr[BWN];
#pragma unroll
                for (int i = 0; i < BWM; i++)
                    af[i] = f8_pair(frag_bf16<true, BIG>(at, wm * BWM * 32 + i * 32, ks, lane), frag_bf16<true, BIG>(at, wm * BWM * 32 + i * 32, ks + 8, lane));
#pragma unroll
                for (int j = 0; j < BWN; j++)
                    bfr[j] = f8_pair(frag_bf16<true, BIG>(bt, wn * BWN * 32 + j * 32, ks, lane), frag_bf16<true, BIG>(bt, wn * BWN * 32 + j * 32, ks + 8, lane));
#pragma unroll
                for (int i = 0; i < BWM; i++)
#pragma unroll
                    for (int j = 0; j < BWN; j++) {
                        if constexpr (sizeof(TC) == 2) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, 127, 0, 127);   // C^T
                        else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bfr[j], acc[i][j], 0, 0, 0, 127, 0, 127);
                    }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < BK; ks += 16) {
            bf16x8 af[BWM], bfr[BWN];
#pragma unroll
            for (int i = 0; i < BWM; i++) af[i] = frag_bf16<AKC, BIG>(at, wm * BWM * 32 + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < BWN; j++) bfr[j] = frag_bf16<BKC, BIG>(bt, wn * BWN * 32 + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < BWM; i++)
#pragma unroll
                for (int j = 0; j < BWN; j++) {
                    if constexpr (sizeof(TC) == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
        }
        }
        __syncthreads();
    }
#undef LOAD_A_
    if constexpr (EPI != 0) {
        if constexpr (sizeof(TC) == 2) epilogue_big_t<0, EPI>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, true, g.alpha);
        else epilogue_big<TC, 0, EPI>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, true, g.ldc, g.alpha);
        return;
    }
    if constexpr (FP8) {      // per-tensor dequantisation factors live on the device: fold them into alpha
        const float a8 = g.alpha * g.scale_a[0] * g.scale_b[0];
        if constexpr (sizeof(TC) == 2) epilogue_big_t<0>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, true, a8);
        else epilogue_big<TC, 0>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, true, g.ldc, a8);
        return;
    }
    if (GEMM_EXP == 4) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < BWM; i++)
#pragma unroll
            for (int j = 0; j < BWN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) ss += acc[i][j][r];
        if (ss == 123.456f) C[0] = (TC)0;
        return;
    }
    const bool lead = (split == 0);
    if constexpr (sizeof(TC) == 2) {      // bf16 C is never an atomic target (mh_gemm requires f32 for split-K)
        if (g.accumulate) epilogue_big_t<1>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, lead, g.alpha);
        else epilogue_big_t<0>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, lead, g.alpha);
        return;
    }
    if constexpr (PART) {
        // partial tile -> workspace [part][M][N] with plain 16-byte stores (part = z * splits + split); 16.7 M same-matrix
        // f32 atomics of a 64-way split cost more than the whole K loop, a fold pass over the partials does not.  (Its own
        // instance: with three epilogues inlined into one kernel the compiler spilled in the f32 instances.)
        if constexpr (sizeof(TC) == 4) {      // partial tiles are stored as bf16 (see fold_partials_kernel)
            bf16_t* P = reinterpret_cast<bf16_t*>(g.ws) + ((long)z * gridDim.y + split) * (long)g.M * g.N;
            epilogue_big<bf16_t, 0>(g, P, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, false, (long)g.N, g.alpha);
        }
        return;
    }
    if (g.atomic) {
        if constexpr (sizeof(TC) == 4)
            epilogue_atomic_big(g, C, acc, tile_m * BIG + wm * BWM * 32, tile_n * BIG + wn * BWN * 32, lane);
    } else if (g.accumulate) {
        epilogue_big<TC, 1>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, lead, g.ldc, g.alpha);
    } else {
        epilogue_big<TC, 0>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, lead, g.ldc, g.alpha);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_pp_kernel: the same 256 x 256 x 64 tile and wave grid, a different main loop (cdna_hip_programming.md §5: direct-to-LDS
// staging, counted waits, raw barriers, the two wave rows in ping-pong).
//   * staging: global_load_lds_dwordx4 (1 KiB per wave instruction, no staging registers, no ds_write pass).  The LDS images are
//     lane-linear, the bank swizzle sits in the per-lane SOURCE address and in the fragment reads (rule 21):
//       K-contiguous operand: [256 rows][128 B], 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7)  -> ds_read_b128 conflict free
//       K-strided operand   : [64 k][512 B], byte offset o of row k stored at o ^ ((k & 3) << 6)             -> ds_read_b64_tr_b16 conflict free
//   * two LDS stages; tile t + 1 is requested during k-steps 0 and 1 of tile t (4 pieces per wave each) and waited for with
//     vmcnt(0) at the end of k-step 3, when nothing younger is in flight: >= 2 k-steps (~1000 cycles) of MFMA time cover the latency.
//   * a k-step = LOAD segment (6 fragment reads, the stage requests) | barrier | COMPUTE segment (8 MFMAs) | barrier; waves 4-7 run
//     one barrier behind waves 0-3, so on every SIMD one wave computes while its partner loads (the matrix pipe is never left idle
//     by the fragment reads, and the two groups' LDS bursts do not collide).
// Epilogues, XCD order, split-K partials and the fused epilogues are gemm_big_kernel's.
constexpr int PP_OP = BIG * 64 * 2;          // bytes of one operand tile image
constexpr int PP_STAGE = 2 * PP_OP;
constexpr int PP_LDS = 136 * 1024;           // 2 stages (128 KiB); the epilogue images (133120 B + reduction scratch) reuse them
typedef __attribute__((address_space(3))) void pp_lds_t;
typedef __attribute__((address_space(1))) const void pp_glb_t;

#define PP_GLDS(gp, lp, off) __builtin_amdgcn_global_load_lds((pp_glb_t*)(gp), (pp_lds_t*)(lp), 16, (off), 0)

// one operand tile = 32 pieces of 1 KiB; wave w requests pieces 4 w .. 4 w + 3
template <bool KC>
struct PPStage {
    const bf16_t* p0;      // this lane's source address for piece 0 of its wave at k0 = kbeg (even-piece swizzle)
    const bf16_t* p1;      // ... with the odd-piece swizzle
    long step;             // elements between two pieces of this wave
    long kadv;             // elements per K-tile
    // rows: tile rows for KC (clamped to dim - 1, optional batch-window remap), k rows for KS
    __device__ __forceinline__ void init(const bf16_t* base, long ld, int tile0, int dim, int kbeg, int wave, int lane, int adj0, int bnd, int skip) {
        if constexpr (KC) {
            // piece i of wave w = rows 32 w + 8 i + (lane >> 3); LDS chunk lane & 7 holds source chunk (lane & 7) ^ ((row >> 1) & 7),
            // (row >> 1) & 7 = (4 i + (lane >> 4)) & 7: two variants (i even / odd)
            const int r = 32 * wave + (lane >> 3);
            const int c0 = (lane & 7) ^ ((lane >> 4) & 7), c1 = c0 ^ 4;
            // rows are clamped / remapped per piece below: keep the row-independent part here
            p0 = base + kbeg + c0 * 8;
            p1 = base + kbeg + c1 * 8;
            step = ld;
            kadv = 64;
            row0_ = r; tile0_ = tile0; dim_ = dim; adj0_ = adj0; bnd_ = bnd; skip_ = skip;
        } else {
            // piece i of wave w = k rows 8 w + 2 i + (lane >> 5); LDS 16-byte chunk lane & 31 holds source chunk (lane & 31) ^ ((k & 3) << 2),
            // k & 3 = (2 i + (lane >> 5)) & 3: two variants (i even / odd)
            const int k = 8 * wave + (lane >> 5);
            const int c0 = (lane & 31) ^ (((lane >> 5) & 3) << 2), c1 = (lane & 31) ^ (((2 + (lane >> 5)) & 3) << 2);
            p0 = base + (long)(kbeg + k) * ld + tile0 + c0 * 8;
            p1 = base + (long)(kbeg + k) * ld + tile0 + c1 * 8;
            step = 2 * ld;
            kadv = 64 * ld;
            row0_ = 0; tile0_ = 0; dim_ = 0; adj0_ = 0; bnd_ = 0; skip_ = 0;
        }
    }
    int row0_, tile0_, dim_, adj0_, bnd_, skip_;
    // request K-tile t of this operand into the LDS image `img` (this wave's four pieces)
    __device__ __forceinline__ void issue(char* img, int wave, int t) const {
        char* dst = img + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bf16_t* src;
            if constexpr (KC) {
                const int rl = min(tile0_ + row0_ + 8 * i, dim_ - 1) - tile0_;
                const long row = tile0_ + rl + adj0_ + (rl >= bnd_ ? skip_ : 0);
                src = ((i & 1) ? p1 : p0) + row * step + (long)t * kadv;
            } else {
                src = ((i & 1) ? p1 : p0) + (long)i * step + (long)t * kadv;
            }
            PP_GLDS(src, dst + i * 1024, 0);
        }
    }
};

// fragment of row block `rb` (32 rows) at k-step s of a K-contiguous image: lane offset table off[s], block stride 4096 B
__device__ __forceinline__ bf16x8 pp_frag_kc(const char* img, const unsigned (&off)[4], int rb, int s) {
    return *reinterpret_cast<const bf16x8*>(img + off[s] + rb * 4096);
}
// ... of a K-strided image: `off` = the lane's offset for this 32-column block (pp_ks_off), k-step stride 8192 B, second 4-row group + 2048 B
__device__ __forceinline__ bf16x8 pp_frag_ks(const char* img, unsigned off, int s) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const char* a0 = img + off + s * 8192;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 2048));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// ds_read_b64_tr_b16 address of this lane for 32-column block cb (0..7) of a K-strided image at k-step 0: lane 4 q + p of a 16-lane
// group supplies k row q (+ 8 for the upper lane half), columns 4 p .. 4 p + 3 of the group's 16 columns; the 64-byte column group
// index is XORed with the k row (k & 3 = q)
__device__ __forceinline__ unsigned pp_ks_off(int cb, int lane) {
    const int g16 = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    return (8 * (g16 >> 1) + q) * 512 + (cb >> 2) * 256 + (((cb & 3) ^ q) << 6) + 32 * (g16 & 1) + 8 * p4;
}

template <typename TC, bool AKC, bool BKC, bool PART = false, int EPI = 0>
__global__ __launch_bounds__(NTB) void gemm_pp_kernel(GemmArgs g) {
    static_assert(EPI == 0 || (AKC && !PART), "fused epilogues: K-contiguous A, no split-K");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles = gridDim.x, nwg = tiles * gridDim.y * gridDim.z;
    const int lin = blockIdx.x + tiles * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = lin & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    const int wgid = unit % tiles, slice = unit / tiles;
    const int tile_m = wgid / g.tiles_n, tile_n = wgid % g.tiles_n;
    const int z = slice / (int)gridDim.y;
    const int b1 = z / g.batch2, b2 = z % g.batch2;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + b1 * g.sA1 + b2 * g.sA2;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B) + b1 * g.sB1 + b2 * g.sB2;
    TC* C = reinterpret_cast<TC*>(g.C) + b1 * g.sC1 + b2 * g.sC2;
    const int split = slice % (int)gridDim.y;
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nt = (kend - kbeg) / 64;

    int adj0 = 0, bnd = 1 << 30;
    if constexpr (AKC) {
        if (g.a_rpb > 0) {
            const int bq = min((tile_m * BIG) / g.a_rpb, g.w_last);
            adj0 = bq * g.a_skip;
            bnd = bq < g.w_last ? (bq + 1) * g.a_rpb - tile_m * BIG : (1 << 30);
        }
    }
    PPStage<AKC> sa;
    PPStage<BKC> sb;
    sa.init(A, g.lda, tile_m * BIG, g.M, kbeg, wave, lane, adj0, bnd, g.a_skip);
    sb.init(B, g.ldb, tile_n * BIG, g.N, kbeg, wave, lane, 0, 1 << 30, 0);

    // fragment read offsets (lane-dependent part; see the image layouts above): K-contiguous: one per k-step, the row block is an
    // immediate; K-strided: one per 32-column block of this wave, the k-step is an immediate
    unsigned offa[4], offb[4];
    if constexpr (AKC) {
        const int r = lane & 31, h = lane >> 5, f = (r >> 1) & 7;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) offa[s4] = wm * 4 * 4096 + r * 128 + (((2 * s4 + h) ^ f) << 4);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) offa[i] = pp_ks_off(wm * 4 + i, lane);
    }
    if constexpr (BKC) {
        const int r = lane & 31, h = lane >> 5, f = (r >> 1) & 7;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) offb[s4] = wn * 2 * 4096 + r * 128 + (((2 * s4 + h) ^ f) << 4);
    } else {
#pragma unroll
        for (int j = 0; j < 2; j++) offb[j] = pp_ks_off(wn * 2 + j, lane);
        offb[2] = offb[3] = 0;
    }

    f32x16 acc[BWM][BWN];
#pragma unroll
    for (int i = 0; i < BWM; i++)
#pragma unroll
        for (int j = 0; j < BWN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    if (nt > 0) {
        sa.issue(smem, wave, 0);
        sb.issue(smem + PP_OP, wave, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // waves 4-7 run one barrier behind waves 0-3
#pragma unroll 1
    for (int t = 0; t < nt; t++) {
        const char* at = smem + (t & 1) * PP_STAGE;
        const char* bt = at + PP_OP;
        char* an = smem + ((t + 1) & 1) * PP_STAGE;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            // ---- LOAD segment
            bf16x8 af[BWM], bfr[BWN];
#pragma unroll
            for (int j = 0; j < BWN; j++) {
                if constexpr (BKC) bfr[j] = pp_frag_kc(bt, offb, j, s);
                else bfr[j] = pp_frag_ks(bt, offb[j], s);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < BWM; i++) {
                if constexpr (AKC) af[i] = pp_frag_kc(at, offa, i, s);
                else af[i] = pp_frag_ks(at, offa[i], s);
            }
            if (t + 1 < nt) {
                if (s == 0) sa.issue(an, wave, t + 1);
                if (s == 1) sb.issue(an + PP_OP, wave, t + 1);
            }
            if (s == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile t + 1 (requested two k-steps ago) has landed
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- COMPUTE segment
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < BWM; i++)
#pragma unroll
                for (int j = 0; j < BWN; j++) {
                    if constexpr (sizeof(TC) == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (EPI != 0) {
        if constexpr (sizeof(TC) == 2) epilogue_big_t<0, EPI>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, true, g.alpha);
        else epilogue_big<TC, 0, EPI>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, true, g.ldc, g.alpha);
        return;
    }
    const bool lead = (split == 0);
    if constexpr (sizeof(TC) == 2) {
        if (g.accumulate) epilogue_big_t<1>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, lead, g.alpha);
        else epilogue_big_t<0>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, lead, g.alpha);
        return;
    }
    if constexpr (PART) {
        if constexpr (sizeof(TC) == 4) {      // partial tiles are stored as bf16 (see fold_partials_kernel)
            bf16_t* P = reinterpret_cast<bf16_t*>(g.ws) + ((long)z * gridDim.y + split) * (long)g.M * g.N;
            epilogue_big<bf16_t, 0>(g, P, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, false, (long)g.N, g.alpha);
        }
        return;
    }
    if (g.atomic) {
        if constexpr (sizeof(TC) == 4)
            epilogue_atomic_big(g, C, acc, tile_m * BIG + wm * BWM * 32, tile_n * BIG + wn * BWN * 32, lane);
    } else if (g.accumulate) {
        epilogue_big<TC, 1>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, lead, g.ldc, g.alpha);
    } else {
        epilogue_big<TC, 0>(g, C, acc, smem, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, lead, g.ldc, g.alpha);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_pq_kernel: gemm_pp_kernel made PERSISTENT — one workgroup per CU walks its share of the (tile, K-slice) units.
// The projection / gradient GEMMs of the step have K = 512 ... 1536: 8 ... 24 K-tiles per output tile, so with one launch-scheduled
// workgroup per tile a third of the time went to per-tile fixed costs with the matrix pipe idle — the first K-tile's load latency,
// and an epilogue whose 128 KiB of C stores had to drain before the CU could take its next tile, with all 256 CUs in that phase
// at the same moment (an HBM write burst, then an HBM-idle K loop).  Here
//   * the stage pipeline runs ACROSS units: the next unit's first K-tile is requested during the last K-tile of the current one;
//   * the epilogue stages C through the LDS stage it has just consumed (half a tile at a time for bf16, a quarter for f32: 66.5 KiB
//     in a 68 KiB slot) while the other stage already holds the next unit's first K-tile;
//   * the C stores are only ISSUED in the epilogue: they drain underneath the next unit's K loop.
// vmcnt counts stores and loads together, in order: the first wait of a unit's K loop (end of K-tile 0) also waits for the
// stores issued ~1 us earlier; what is left of their drain time is the only exposed part of the epilogue's memory traffic.
// Cycle stamps (round 4, tools/exp/pq_stamps_patch.py) of a K loop segment, per wave row: fragment reads + requests + wait ~880,
// barrier ~140, 16 MFMAs ~700, barrier ~300: ~2000 cycles for 2 x 512 cycles of matrix-pipe work per SIMD.  Built and measured on one
// box against this loop, none faster: a four-segment ring with the request for segment s + 3 issued between the MFMAs (the load
// phase drops to ~350 cycles, the MFMA phase grows to ~800: same 2000), 1 / 2 / 3 of its 4 pieces in the load phase (equal; the
// long-K gradient products 3-11 % slower), no requests at all (1750: the floor of the two-barrier structure).
// The unit decode (integer divisions, per-lane addresses: ~1.2k cycles in front of a unit's first segment) moved into the epilogue's
// waiting time (a third unit record, decoded by the wave row that is not writing): no spills, products alone 0-3 % faster, step +0.28 % +- 0.11.
// ---- stage layout of gemm_pq_kernel: every operand tile is TWO k-half sub-images of 16 KiB (k in [0, 32) and [32, 64) of the K-tile),
// so that a segment of two k-steps reads one sub-image per operand and the other half can be in flight:
//   K-contiguous: [256 rows][64 B], 16-byte chunk c of row r stored at chunk c ^ F[(r >> 2) & 3], F = {0, 2, 3, 1}
//   K-strided   : [32 k][512 B], byte offset o of row k stored at o ^ ((k & 3) << 6) ^ (((k >> 3) & 1) << 5)
// Both swizzles are made for the fragments of v_mfma_f32_16x16x32_bf16 (a lane = row l & 15, k = 8 (l >> 4) .. + 7: one ds_read_b128
// of a K-contiguous row's chunk l >> 4, or two ds_read_b64_tr_b16 of k rows 8 (l >> 4) + q (+ 4) of a K-strided image): each lane
// group of a wave instruction meets 64 distinct banks (F: the four chunk slots a 16-lane group of a b128 read takes from rows
// r, r + 4, r + 8, r + 12 differ; K-strided: the eight (k >> 3 & 1, k & 3) combinations of a 32-lane half take eight different 32-byte
// slots of a 256-byte span).  The 16 x 16 x 32 shape does the same flops per cycle as 32 x 32 x 16, but the chip holds a higher clock
// under it and its shorter MFMAs interleave better with the partner wave's loads: the products of the step ran 4-8 % faster with it.
// slot = A_k0 | A_k1 | B_k0 | B_k1 (16 KiB each) | 4 KiB that only the C staging uses.  A sub-image = 16 pieces of 1 KiB: 2 per wave.
constexpr int P2_SUB = 16 * 1024;
__device__ __forceinline__ int p2_f4(int q) { return (0x78 >> (2 * q)) & 3; }       // F = {0, 2, 3, 1}
template <bool KC>
struct P2Stage {
    const bf16_t* p0;      // this lane's source address of its wave's piece 0 of sub-image k0 at K-tile 0 (KS: the even-piece swizzle)
    const bf16_t* p1;      // KS: the odd-piece swizzle; KC: piece 1 (row clamp / window remap applied per piece)
    long kadv, hadv, step; // elements per K-tile, per k-half, (KS) per piece
    __device__ __forceinline__ void init(const bf16_t* base, long ld, int tile0, int dim, int kbeg, int wave, int lane, int adj0, int bnd, int skip) {
        if constexpr (KC) {
            // piece i of wave w = rows 32 w + 16 i + (lane >> 2); LDS chunk lane & 3 holds source chunk (lane & 3) ^ F[(row >> 2) & 3],
            // (row >> 2) & 3 = (lane >> 4) & 3 for every piece
            const int c = (lane & 3) ^ p2_f4((lane >> 4) & 3);
            const bf16_t* b0 = base + kbeg + c * 8;
            const int r0 = 32 * wave + (lane >> 2), r1 = r0 + 16;
            const int l0 = min(tile0 + r0, dim - 1) - tile0, l1 = min(tile0 + r1, dim - 1) - tile0;
            p0 = b0 + (long)(tile0 + l0 + adj0 + (l0 >= bnd ? skip : 0)) * ld;
            p1 = b0 + (long)(tile0 + l1 + adj0 + (l1 >= bnd ? skip : 0)) * ld;
            kadv = 64; hadv = 32; step = 0;
        } else {
            // piece i of wave w = k rows 4 w + 2 i + (lane >> 5) of the sub-image; LDS 16-byte chunk lane & 31 holds source chunk
            // (lane & 31) ^ ((k & 3) << 2) ^ (((k >> 3) & 1) << 1), k & 3 = (2 i + (lane >> 5)) & 3, k >> 3 = wave >> 1
            const int k = 4 * wave + (lane >> 5);
            const int h8 = ((wave >> 1) & 1) << 1;
            const int c0 = (lane & 31) ^ (((lane >> 5) & 3) << 2) ^ h8, c1 = (lane & 31) ^ (((2 + (lane >> 5)) & 3) << 2) ^ h8;
            p0 = base + (long)(kbeg + k) * ld + tile0 + c0 * 8;
            p1 = base + (long)(kbeg + k + 2) * ld + tile0 + c1 * 8;
            kadv = 64 * ld; hadv = 32 * ld; step = 0;
        }
    }
    // request k-half kh of K-tile t into the sub-image at `sub` (this wave's two pieces)
    __device__ __forceinline__ void issue(char* sub, int wave, int t, int kh) const {
        char* dst = sub + wave * 2048;
        const long o = (long)t * kadv + (long)kh * hadv;
        PP_GLDS(p0 + o, dst, 0);
        PP_GLDS(p1 + o, dst + 1024, 0);
    }
};
// fragments of v_mfma_f32_16x16x32_bf16 for 16 rows (A) / 16 columns (B) over the 32 k of a sub-image:
// K-contiguous: ONE ds_read_b128 at this lane's offset (row l & 15, chunk (l >> 4) ^ F) + 1024 B per 16-row block
__device__ __forceinline__ unsigned q_kc_off(int row0, int lane) {
    const int r = lane & 15, c = lane >> 4;
    return (unsigned)((row0 + r) * 64 + ((c ^ p2_f4((r >> 2) & 3)) << 4));
}
__device__ __forceinline__ bf16x8 q_frag_kc(const char* sub, unsigned off, int blk) {
    return *reinterpret_cast<const bf16x8*>(sub + off + blk * 1024);
}
// K-strided: two ds_read_b64_tr_b16; lane 4 q + p of a 16-lane group g supplies k row 8 g + q (+ 4), columns 4 p .. 4 p + 3 of the
// block's 16; the block's 32-byte slot index is XORed with (q << 1) ^ (g & 1) (the image's swizzle), so every block has its own offset
__device__ __forceinline__ unsigned q_ks_off(int jb16, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    return (unsigned)((8 * g + q) * 512 + ((jb16 ^ ((q << 1) ^ (g & 1))) << 5) + 8 * p4);
}
__device__ __forceinline__ bf16x8 q_frag_ks(const char* sub, unsigned off) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sub + off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sub + off + 2048));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
constexpr int QM = 8, QN = 4;                  // 16 x 16 accumulator blocks of a wave's 128 x 64 part of the tile

constexpr int PQ_SLOT = 68 * 1024;            // LDS stage slot: A image 32 KiB | B image 32 KiB | 4 KiB that only the C staging uses
constexpr int PQ_LDS = 2 * PQ_SLOT;

// pq_store_note: C leaves with NON-TEMPORAL stores (global_store_dwordx4 ... nt).  A C tile is written once and read by another kernel
// tens of MB later; kept in L2 like ordinary lines it evicted the weight panel and the A tiles its own launch re-reads (every weight
// tile is read by all 256 workgroups, an A tile by tiles_n of them): [65536 x 1536] x [1536 x 512] -> bf16 115.4 -> 103.6 us alone on
// the chip, the K = 512 bf16 products -4 %, f32 C at K = 1024 -4 %, f32 C at K = 512 +4 % (choosing the form per launch by K through a
// run-time flag made the non-temporal path itself 6 % slower and the step +0.39 % +- 0.08: one form everywhere); step -0.62 % +- 0.09.
// ---- epilogues of gemm_pq_kernel: C through the LDS stage the unit has just consumed.  Measured with in-kernel cycle stamps
// (tools/exp/pq_stamps_patch.py; K = 512: a third of a unit's 48k cycles was epilogue): a write phase took ~4000 cycles because every
// bias quad was a dependent load behind an s_waitcnt vmcnt(0) (emitted with or without a bias) and every element cost a multiply-add,
// a max, a select and its own conversion.  Hence: activation and the plain case are template parameters (with a run-time flag in the
// element loop the compiler computes both forms of all 128 values before it selects, and spills them), the bias quads of a column
// block are requested together, two elements share one v_cvt_pk_bf16_f32, and the staging reads of a phase all land in registers
// before the barrier that frees the image, so the stores are issued while the other wave row already writes the next phase.
//
// bf16 C (or the bf16 split-K partial tiles: C = the slice's partial matrix, ldc = N, remap = false).  The accumulators are in C^T
// form: block (i, j) of lane (c16 = l & 15, g4 = l >> 4) holds columns 16 j + 4 g4 + {0..3} of row 16 i + c16 of the wave's
// 128 x 64 part; rows [128 half, +128) of the tile go through `t`, the wave row `half` writing, everybody storing 16-byte
// row-contiguous chunks.  KIND 0: alpha == 1, no bias, no activation: both wave rows pack their accumulators first (one
// v_cvt_pk_bf16_f32 per two elements, in place), so a write phase is 32 ds_write_b64 and nothing else; 1: alpha, bias (the lane's four
// bias quads requested together); 2: + ReLU.
template <int MODE, int EPI, int KIND>
__device__ __forceinline__ void pq_epilogue_bf16(const GemmArgs& g, bf16_t* C, long ldc, bool remap, f32x4 (&acc)[QM][QN], char* smem_c,
                                                 int tile_row0, int tile_col0, int wm, int wn, int lane, int tid, bool has_bias, float alpha) {
    constexpr int PITCH = BIG + 4, HALF = BIG / 2;
    // the staging addresses are recomputed per unit: hoisted out of the unit loop they would live (spilled) across the K loop
    asm volatile("" : "+v"(lane), "+v"(tid));
    bf16_t* t = reinterpret_cast<bf16_t*>(smem_c);
    const int c16 = lane & 15, g4 = lane >> 4;
    unsigned pk[QM][QN][2];
    if constexpr (KIND == 0) {
#pragma unroll
        for (int i = 0; i < QM; i++)
#pragma unroll
            for (int j = 0; j < QN; j++) {
                pk[i][j][0] = pack_bf2(acc[i][j][0], acc[i][j][1]);
                pk[i][j][1] = pack_bf2(acc[i][j][2], acc[i][j][3]);
            }
    }
    // physical rows of this tile (row windows, GemmArgs.c_rpb): a tile touches at most two windows when c_rpb >= 256
    const bool win = remap && g.c_rpb > 0, win2 = win && g.c_rpb >= BIG;
    long c_adj = 0;
    int c_bnd = 1 << 30;
    if (win2) {
        const int bq = min(tile_row0 / g.c_rpb, g.w_last);
        c_adj = (long)bq * g.c_skip;
        if (bq < g.w_last) c_bnd = (bq + 1) * g.c_rpb;
    }
    float sq_sum = 0.f, sq_cnt = 0.f;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if (wm == half) {
            f32x4 bv[QN];
            if constexpr (KIND != 0) {
#pragma unroll
                for (int j = 0; j < QN; j++) bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (has_bias) {
#pragma unroll
                    for (int j = 0; j < QN; j++) bv[j] = *reinterpret_cast<const f32x4*>(g.bias + tile_col0 + wn * QN * 16 + 16 * j + 4 * g4);
                }
            }
#pragma unroll
            for (int i = 0; i < QM; i++)
#pragma unroll
                for (int j = 0; j < QN; j++) {
                    u32x2 o2;
                    if constexpr (KIND == 0) {
                        o2 = u32x2{pk[i][j][0], pk[i][j][1]};
                    } else {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            v[e] = alpha * acc[i][j][e] + bv[j][e];
                            if constexpr (KIND == 2) v[e] = fmaxf(v[e], 0.f);
                        }
                        o2 = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                    }
                    *reinterpret_cast<u32x2*>(t + (16 * i + c16) * PITCH + wn * QN * 16 + 16 * j + 4 * g4) = o2;
                }
        }
        __syncthreads();
        constexpr int CPR = BIG / 8;                 // 16-byte chunks per tile row
        constexpr int NCH = HALF * CPR / NTB;
        u32x4 o[NCH];
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int cid = tid + i * NTB;
            const int lr = cid / CPR, c = cid % CPR;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(t + lr * PITCH + c * 8);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(t + lr * PITCH + c * 8 + 4);
            o[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int cid = tid + i * NTB;
            const int lr = cid / CPR, c = cid % CPR;
            const int grow = tile_row0 + half * HALF + lr;
            if (grow >= g.M) continue;               // ragged last row tile (K-contiguous A only)
            long prow = grow;
            if (win2) prow = grow + c_adj + (grow >= c_bnd ? g.c_skip : 0);
            else if (win) prow = c_phys_row(g, grow);
            bf16_t* dst = C + prow * ldc + tile_col0 + c * 8;
            if constexpr (MODE == 1) {
                const u32x4 old = *reinterpret_cast<const u32x4*>(dst);
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const float a0 = __uint_as_float(o[i][w] << 16) + __uint_as_float(old[w] << 16);
                    const float a1 = __uint_as_float(o[i][w] & 0xffff0000u) + __uint_as_float(old[w] & 0xffff0000u);
                    o[i][w] = pack_bf2(a0, a1);
                }
            }
            if (remap) __builtin_nontemporal_store(o[i], reinterpret_cast<u32x4*>(dst));      // see pq_store_note
            else *reinterpret_cast<u32x4*>(dst) = o[i];                                        // split-K partial tiles: the fold launch reads them next
            if constexpr (EPI == MH_EPI_SQERR) {
                const int rpb = g.epi.rows_per_batch;             // % 256 == 0: a tile lies inside one batch
                const long b = tile_row0 / rpb;
                const int tt = grow - (int)(b * rpb);
                if (g.epi.mask[b * rpb + tt] != 0.f) {
                    const float* tg = g.epi.tgt + b * g.epi.tgt_bs + (long)tt * g.N + tile_col0 + c * 8;
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(tg), t1 = *reinterpret_cast<const f32x4*>(tg + 4);
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const float d0 = __uint_as_float(o[i][w] << 16) - (w < 2 ? t0[2 * w] : t1[2 * w - 4]);
                        const float d1 = __uint_as_float(o[i][w] & 0xffff0000u) - (w < 2 ? t0[2 * w + 1] : t1[2 * w - 3]);
                        sq_sum += d0 * d0 + d1 * d1;
                    }
                    sq_cnt += 8.f;
                }
            }
        }
    }
    if constexpr (EPI == MH_EPI_SQERR) {
        float* red = reinterpret_cast<float*>(smem_c + HALF * PITCH * 2);     // behind the half-tile image (66560 of 69632 bytes): not a stage image
        sq_sum = wave_sum(sq_sum);
        sq_cnt = wave_sum(sq_cnt);
        if (lane == 0) { red[2 * (tid >> 6)] = sq_sum; red[2 * (tid >> 6) + 1] = sq_cnt; }
        __syncthreads();
        if (tid == 0) {
            float a = 0.f, n = 0.f;
#pragma unroll
            for (int w = 0; w < NTB / 64; w++) { a += red[2 * w]; n += red[2 * w + 1]; }
            const float inv = 1.f / (float)g.N;
            atomicAdd(g.epi.sq, a * inv);
            atomicAdd(g.epi.sq + 1, n * inv);
        }
        __syncthreads();
    }
}
// f32 C (C^T accumulators: block (i, j) of lane (c16, g4) holds columns 16 j + 4 g4 + {0..3} of row 16 i + c16): rows [64 q, +64) of the
// tile through `t`, q = 0 .. 3
template <int MODE, int EPI, bool RELU>
__device__ __forceinline__ void pq_epilogue_f32(const GemmArgs& g, float* C, f32x4 (&acc)[QM][QN], char* smem_c, int tile_row0, int tile_col0,
                                                int wm, int wn, int lane, int tid, bool lead, long ldc, float alpha) {
    constexpr int PITCH = BIG + 4, QR = BIG / 4;
    asm volatile("" : "+v"(lane), "+v"(tid));        // see pq_epilogue_bf16
    float* t = reinterpret_cast<float*>(smem_c);
    const int c16 = lane & 15, g4 = lane >> 4;
    TileRow trow{0, 0, tile_row0};
    if constexpr (EPI == MH_EPI_MASKPOS) { trow.b0 = tile_row0 / g.epi.rows_per_batch; trow.t0 = tile_row0 - trow.b0 * g.epi.rows_per_batch; }
    uint64_t drop_blk0 = 0;
    uint32_t thr = 0;
    float dscale = 1.f;
    if constexpr (EPI == MH_EPI_DROPADD) {
        uint64_t off = g.epi.offset;
        if (g.epi.dev_base) off += *g.epi.dev_base & ~7ull;
        drop_blk0 = off >> 3;
        thr = drop16_thr(g.epi.p);
        dscale = drop16_scale(thr);
    }
    f32x4 bias[QN];
#pragma unroll
    for (int j = 0; j < QN; j++)
        bias[j] = (g.bias && lead) ? *reinterpret_cast<const f32x4*>(g.bias + tile_col0 + wn * QN * 16 + 16 * j + 4 * g4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (wm == (q >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 4; ii++)
#pragma unroll
                for (int j = 0; j < QN; j++) {
                    f32x4 v = acc[4 * (q & 1) + ii][j] * alpha + bias[j];
                    if constexpr (RELU) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
                    }
                    *reinterpret_cast<f32x4*>(t + (16 * ii + c16) * PITCH + wn * QN * 16 + 16 * j + 4 * g4) = v;
                }
        }
        __syncthreads();
        if constexpr (EPI == MH_EPI_DROPADD) {
            constexpr int CPR8 = BIG / 8, NCH8 = QR * CPR8 / NTB;
            f32x4 x0[NCH8], x1[NCH8];
#pragma unroll
            for (int i = 0; i < NCH8; i++) {
                const int cid = tid + i * NTB;
                const float* src = t + (cid / CPR8) * PITCH + (cid % CPR8) * 8;
                x0[i] = *reinterpret_cast<const f32x4*>(src);
                x1[i] = *reinterpret_cast<const f32x4*>(src + 4);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NCH8; i++) {
                const int cid = tid + i * NTB;
                const int lr = cid / CPR8, c = cid % CPR8;
                const int grow = tile_row0 + q * QR + lr, gcol = tile_col0 + c * 8;
                if (grow >= g.M) continue;
                const float* rp = g.epi.resid + (long)grow * g.N + gcol;
                f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
                const f32x4 y0 = round_bf16_4(x0[i]), y1 = round_bf16_4(x1[i]);
                const uint32_t keep = drop16_keep8(drop_blk0 + (((uint64_t)grow * (uint64_t)g.N + (uint64_t)gcol) >> 3), g.epi.seed, thr);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    r0[e] = __fadd_rn(r0[e], (keep & (1u << e)) ? __fmul_rn(y0[e], dscale) : 0.f);
                    r1[e] = __fadd_rn(r1[e], (keep & (16u << e)) ? __fmul_rn(y1[e], dscale) : 0.f);
                }
                float* dst = C + (long)grow * ldc + gcol;
                __builtin_nontemporal_store(r0, reinterpret_cast<f32x4*>(dst));
                __builtin_nontemporal_store(r1, reinterpret_cast<f32x4*>(dst + 4));
            }
        } else if constexpr (EPI == MH_EPI_MASKPOS) {
            // mask token select + positional add with every operand REQUESTED beside the LDS reads, in front of the barrier (as quads
            // looked up one by one inside the store loop — mask, then token, then pos, each behind the other — the launch ran 84 us
            // against 56 for the plain f32 product): a thread's column quad is the same in all of its rows (NTB % CPR == 0), so the token
            // quad is one load; the mask values and the positional quads of its QR * CPR / NTB rows are independent loads
            constexpr int CPR = BIG / 4, NCH = QR * CPR / NTB;
            static_assert(NTB % CPR == 0, "a thread keeps its column quad across rows");
            const int c4 = (tid % CPR) * 4, gcol = tile_col0 + c4;
            const int rpb = g.epi.rows_per_batch, first = g.epi.first;
            f32x4 x[NCH], pq[NCH];
            float mk[NCH];
            const f32x4 tokq = *reinterpret_cast<const f32x4*>(g.epi.token + gcol);
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int lr = (tid + i * NTB) / CPR;
                x[i] = *reinterpret_cast<const f32x4*>(t + lr * PITCH + c4);
                const int grow = min(tile_row0 + q * QR + lr, g.M - 1);
                int tt = trow.t0 + (grow - trow.row0), b = trow.b0;
                if (tt >= rpb) { tt -= rpb; b += 1; }
                pq[i] = *reinterpret_cast<const f32x4*>(g.epi.pos + (long)tt * g.N + gcol);
                mk[i] = tt >= first ? g.epi.mask[(long)b * (rpb - first) + (tt - first)] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int lr = (tid + i * NTB) / CPR;
                const int grow = tile_row0 + q * QR + lr;
                if (grow >= g.M) continue;
                const f32x4 v = (mk[i] != 0.f ? tokq : round_bf16_4(x[i])) + pq[i];
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(C + (long)grow * ldc + gcol));
            }
        } else {
            constexpr int CPR = BIG / 4, NCH = QR * CPR / NTB;
            f32x4 x[NCH];
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int cid = tid + i * NTB;
                x[i] = *reinterpret_cast<const f32x4*>(t + (cid / CPR) * PITCH + (cid % CPR) * 4);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int cid = tid + i * NTB;
                const int lr = cid / CPR, c = cid % CPR;
                const int grow = tile_row0 + q * QR + lr;
                if (grow >= g.M) continue;
                float* dst = C + (long)grow * ldc + tile_col0 + c * 4;
                f32x4 x0 = x[i];
                if constexpr (EPI != 0) x0 = epi_quad<EPI>(g, x0, grow, tile_col0 + c * 4, trow);
                if constexpr (MODE == 1) x0 += *reinterpret_cast<const f32x4*>(dst);
                __builtin_nontemporal_store(x0, reinterpret_cast<f32x4*>(dst));
            }
        }
    }
}

// VAR = the epilogue variant, a template parameter so that each instance carries ONE epilogue (with several inlined side by side the
// register allocator spilled loop invariants across the K loop and reloaded them in front of the stores, each reload a full drain of
// the wave's memory queue): bf16 tiles (C or split-K partials): 0 = alpha == 1, no bias, no activation, 1 = alpha / bias, 2 = + ReLU;
// f32 C: 0 = store, 1 = ReLU, 2 = accumulate.  f32 atomics stay on gemm_pp_kernel (pq_variant()); accumulating bf16 C is not taken by
// the 256 x 256 kernels at all (gemm_try_big_bf16).
template <typename TC, bool AKC, bool BKC, bool PART = false, int EPI = 0, int VAR = 0>
__global__ __launch_bounds__(NTB) void gemm_pq_kernel(GemmArgs g, int units, int tiles, int splits) {
    static_assert(EPI == 0 || (AKC && !PART), "fused epilogues: K-contiguous A, no split-K");
    // C^T accumulators (a lane owns 4 consecutive columns of a row) for every result: bf16 C / split-K partial tiles since round 4, f32 C
    // since round 5 (its staging writes were one 4-byte ds_write per ELEMENT: 16-byte writes now)
    constexpr bool CT = true;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int q8 = units >> 3, r8 = units & 7;

    // fragment read offsets within a sub-image (q_frag_kc / q_frag_ks): K-contiguous: one (the 16-row block is an immediate);
    // K-strided: one per 16-column block of this wave's part
    unsigned offa[AKC ? 1 : QM], offb[BKC ? 1 : QN];
    if constexpr (AKC) offa[0] = q_kc_off(wm * QM * 16, lane);
    else {
#pragma unroll
        for (int i = 0; i < QM; i++) offa[i] = q_ks_off(wm * QM + i, lane);
    }
    if constexpr (BKC) offb[0] = q_kc_off(wn * QN * 16, lane);
    else {
#pragma unroll
        for (int j = 0; j < QN; j++) offb[j] = q_ks_off(wn * QN + j, lane);
    }

    // unit v of this workgroup -> (tile_m, tile_n, batch z, K-slice): XCD-aware order over all units (see gemm_big_kernel); the
    // workgroups of one launch run on XCD blockIdx.x % 8, and v = blockIdx.x + k gridDim.x keeps that residue when gridDim.x % 8 == 0
    P2Stage<AKC> sa, sa_n;
    P2Stage<BKC> sb, sb_n;
    int tile_m = 0, tile_n = 0, z = 0, split = 0, nt = 0;
    int n_tile_m = 0, n_tile_n = 0, n_z = 0, n_split = 0, n_nt = 0;
    auto decode = [&](int v, int& tm, int& tn, int& zz, int& sp, int& ntl, P2Stage<AKC>& pa, P2Stage<BKC>& pb) {
        const int xcd = v & 7;
        const int unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
        const int wgid = unit % tiles, slice = unit / tiles;
        tm = wgid / g.tiles_n; tn = wgid % g.tiles_n;
        zz = slice / splits; sp = slice % splits;
        const int b1 = zz / g.batch2, b2 = zz % g.batch2;
        const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + b1 * g.sA1 + b2 * g.sA2;
        const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B) + b1 * g.sB1 + b2 * g.sB2;
        const int kbeg = sp * g.k_per_split;
        ntl = (min(g.K, kbeg + g.k_per_split) - kbeg) / 64;
        int adj0 = 0, bnd = 1 << 30;
        if constexpr (AKC) {
            if (g.a_rpb > 0) {
                const int bq = min((tm * BIG) / g.a_rpb, g.w_last);
                adj0 = bq * g.a_skip;
                bnd = bq < g.w_last ? (bq + 1) * g.a_rpb - tm * BIG : (1 << 30);
            }
        }
        pa.init(A, g.lda, tm * BIG, g.M, kbeg, wave, lane, adj0, bnd, g.a_skip);
        pb.init(B, g.ldb, tn * BIG, g.N, kbeg, wave, lane, 0, 1 << 30, 0);
    };

    int v = blockIdx.x;
    if (v >= units) return;
    decode(v, tile_m, tile_n, z, split, nt, sa, sb);
    int ctr = 0;                                   // K-tiles consumed so far: stage = ctr & 1
    if (nt > 0) {
        sa.issue(smem, wave, 0, 0);
        sb.issue(smem + 2 * P2_SUB, wave, 0, 0);
        sa.issue(smem + P2_SUB, wave, 0, 1);
        sb.issue(smem + 3 * P2_SUB, wave, 0, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (; v < units; v += gridDim.x) {
        const int vn = v + gridDim.x;
        const bool has_next = vn < units;
        if (has_next) decode(vn, n_tile_m, n_tile_n, n_z, n_split, n_nt, sa_n, sb_n);
        f32x4 acc[QM][QN];
#pragma unroll
        for (int i = 0; i < QM; i++)
#pragma unroll
            for (int j = 0; j < QN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wm == 1) __builtin_amdgcn_s_barrier();          // waves 4-7 run one barrier behind waves 0-3 inside the K loop
#pragma unroll 1
        for (int t = 0; t < nt; t++, ctr++) {
            const char* cur = smem + (ctr & 1) * PQ_SLOT;
            char* nxt = smem + ((ctr + 1) & 1) * PQ_SLOT;
            const bool last = t + 1 == nt;
            const bool feed = !last || (has_next && n_nt > 0);       // something to request during this K-tile
#pragma unroll
            for (int kh = 0; kh < 2; kh++) {
                // ---- LOAD segment: the fragments of both k-steps of this k-half, then the next K-tile's (or unit's) same k-half
                const char* asub = cur + kh * P2_SUB;
                const char* bsub = cur + (2 + kh) * P2_SUB;
                bf16x8 af[QM], bfr[QN];
#pragma unroll
                for (int j = 0; j < QN; j++) {
                    if constexpr (BKC) bfr[j] = q_frag_kc(bsub, offb[0], j);
                    else bfr[j] = q_frag_ks(bsub, offb[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < QM; i++) {
                    if constexpr (AKC) af[i] = q_frag_kc(asub, offa[0], i);
                    else af[i] = q_frag_ks(asub, offa[i]);
                }
                if (feed) {
                    if (!last) {
                        sa.issue(nxt + kh * P2_SUB, wave, t + 1, kh);
                        sb.issue(nxt + (2 + kh) * P2_SUB, wave, t + 1, kh);
                    } else {                                   // the next unit's first K-tile: the pipeline runs across units
                        sa_n.issue(nxt + kh * P2_SUB, wave, 0, kh);
                        sb_n.issue(nxt + (2 + kh) * P2_SUB, wave, 0, kh);
                    }
                    // the sub-images the NEXT segment reads (requested one segment ago) have landed; the four just requested stay in flight
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- COMPUTE segment: 32 MFMAs (16 x 16 x 32: the whole k-half each)
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < QM; i++)
#pragma unroll
                    for (int j = 0; j < QN; j++) {
                        if constexpr (CT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();          // both wave rows level again
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue through the stage just consumed; the other one holds (or is receiving) the next unit's first K-tile
        char* cst = smem + ((ctr + 1) & 1) * PQ_SLOT;
        const int b1 = z / g.batch2, b2 = z % g.batch2;
        TC* C = reinterpret_cast<TC*>(g.C) + b1 * g.sC1 + b2 * g.sC2;
        const bool lead = (split == 0);
        if constexpr (PART) {
            bf16_t* P = reinterpret_cast<bf16_t*>(g.ws) + ((long)z * splits + split) * (long)g.M * g.N;      // bf16 partial tiles
            pq_epilogue_bf16<0, 0, VAR>(g, P, (long)g.N, false, acc, cst, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, false, g.alpha);
        } else if constexpr (sizeof(TC) == 2) {
            pq_epilogue_bf16<0, EPI, VAR>(g, C, g.ldc, true, acc, cst, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, g.bias && lead, g.alpha);
        } else {
            pq_epilogue_f32<VAR == 2 ? 1 : 0, EPI, VAR == 1>(g, C, acc, cst, tile_m * BIG, tile_n * BIG, wm, wn, lane, tid, lead, g.ldc, g.alpha);
        }
        __syncthreads();
        tile_m = n_tile_m; tile_n = n_tile_n; z = n_z; split = n_split; nt = n_nt;
        sa = sa_n; sb = sb_n;
    }
}

// C[r][c] += sum_p P[p][r][c]: quads, 8 partials in flight; TAIL: + the rank-(batch x KT) remainder of the contraction (GemmTail).
// The partial tiles are bf16 (round 4): every K-slice's f32 accumulator is rounded ONCE on its way out — the same 2^-9 relative step
// the bf16 operands already carry — and the sum over the slices runs in f32 here; half the bytes of the f32 form in both directions
// (64 partials of a 512 x 512 gradient: 2 x 64 MB -> 2 x 32 MB per launch pair).
__device__ __forceinline__ f32x4 ld4_bf16(const bf16_t* p) {
    const u32x2 w = *reinterpret_cast<const u32x2*>(p);
    return f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
}
// PS threads share a quad (PS = 4: small outputs WITH a tail): a 512 x 512 gradient is 64 K quads = 256 workgroups of one quad per
// thread; its 64 partials stream at HBM rate (9 us for 33 MB), but the tail's rank-one terms behind them are two more dependent
// rounds of scattered loads per thread (19 us).  With the partials and the tail's terms of a quad dealt to four threads and summed
// through LDS the pass has four times the loads in flight (13 us).
template <bool TAIL, int PS>
__global__ __launch_bounds__(256) void fold_partials_kernel(const bf16_t* __restrict__ P, int parts, long mn, float* __restrict__ C, long ldc,
                                                            int N, GemmTail t) {
    constexpr int QB = 256 / PS;                      // quads per workgroup
    __shared__ f32x4 red[PS > 1 ? 256 : 1];
    const int ql = threadIdx.x % QB, pg = threadIdx.x / QB;
    const long quads = mn / 4;
    for (long qb = blockIdx.x; qb * QB < quads; qb += gridDim.x) {
        const long q = qb * QB + ql;
        const bool live = q < quads;
        const long i = q * 4, r = live ? i / N : 0, c = live ? i % N : 0;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            // this thread's share of the partials: p = pg, pg + PS, ...
            int p = pg;
            for (; p + 7 * PS < parts; p += 8 * PS) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = ld4_bf16(P + (long)(p + u * PS) * mn + i);
#pragma unroll
                for (int u = 0; u < 8; u++) s += v[u];
            }
            for (; p < parts; p += PS) s += ld4_bf16(P + (long)p * mn + i);
            if constexpr (TAIL) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int total = t.batch * t.KT;
                auto ld = [&](int k, float& a, f32x4& b) {
                    const int z = k / t.KT, kk = k - z * t.KT;
                    const long ia = z * t.sA + (long)kk * t.lda + r, ib = z * t.sB + (long)kk * t.ldb + c;
                    a = t.a_f32 ? reinterpret_cast<const float*>(t.A)[ia] : bf2f(reinterpret_cast<const bf16_t*>(t.A)[ia]);
                    if (t.b_f32) b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(t.B) + ib);
                    else {
                        const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(t.B) + ib);
                        b = f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
                    }
                };
                int k = pg;
                for (; k + 7 * PS < total; k += 8 * PS) {          // eight independent pairs in flight
                    float a[8];
                    f32x4 b[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) ld(k + u * PS, a[u], b[u]);
#pragma unroll
                    for (int u = 0; u < 8; u++) acc += b[u] * a[u];
                }
                for (; k < total; k += PS) {
                    float a;
                    f32x4 b;
                    ld(k, a, b);
                    acc += b * a;
                }
                s += acc * t.alpha;
            }
        }
        if constexpr (PS > 1) {
            __syncthreads();                          // (the previous round's readers are done with red)
            red[threadIdx.x] = s;
            __syncthreads();
            if (pg == 0 && live) {
#pragma unroll
                for (int u = 1; u < PS; u++) s += red[u * QB + ql];
            }
        }
        if (pg == 0 && live) {
            f32x4* cp = reinterpret_cast<f32x4*>(C + r * ldc + c);
            *cp = *cp + s;
        }
    }
}
static void launch_fold(const float* ws_, int parts, long mn, float* C, long ldc, int N, hipStream_t s) {
#ifdef MH_EXP
    if (getenv("MH_EXP_SKIP_FOLD")) { g_tail.KT = 0; return; }      // timing experiment: what the fold passes cost inside the step
#endif
    const bf16_t* ws = reinterpret_cast<const bf16_t*>(ws_);
    const long blocks1 = mh_cdiv(mn / 4, 256);
    // the tail's B rows are read as aligned quads of the output's columns: 4-element alignment of its rows and base
    const bool tail = g_tail.KT > 0 && (g_tail.ldb % 4) == 0 && (g_tail.sB % 4) == 0 && ((uintptr_t)g_tail.B & (g_tail.b_f32 ? 15 : 7)) == 0;
    // four threads per quad where the tail's dependent loads ride along on a small output (measured: 19.3 -> 13.2 us; the plain
    // fold of the same 33 MB is at HBM rate with one thread per quad, 9.1 us, and 1.3 us slower shared)
    const bool share = tail && blocks1 < 1024;
    const dim3 grid((unsigned)min(share ? mh_cdiv(mn / 4, 64) : blocks1, share ? 4096L : 2048L));
#define FOLD_(TAIL, PS) hipLaunchKernelGGL((fold_partials_kernel<TAIL, PS>), grid, dim3(256), 0, s, ws, parts, mn, C, ldc, N, g_tail)
    if (tail) {
        if (share) FOLD_(true, 4); else FOLD_(true, 1);
        g_tail.KT = 0;
    } else {
        FOLD_(false, 1);
    }
#undef FOLD_
}


// ---------------------------------------------------------------------------------------------------------------------------
// gemm_w4_kernel (round 5 experiment, verdict item 4 "GEMM structure"): the same 256 x 256 x 64 tile on FOUR waves, one per SIMD, each
// owning a 128 x 128 part (8 x 8 blocks of 16 x 16: 256 accumulator registers).  Per k-half a wave reads 16 fragments for 64 MFMAs (the
// 8-wave kernels: 12 for 32): a K-tile is 32 ds_read_b128 and then 128 MFMAs back to back on AGPR accumulators, one barrier per K-tile.
// K-contiguous A and B, bf16 C, alpha = 1, optional bias; whole tiles only; one workgroup per tile, two stages.
// MEASURED (profiles/r05_l_gemm_four_wave_experiment.txt): bit-equal to mh_gemm, 0.78-0.81 x the persistent 8-wave kernel's rate (633-801
// vs 780-1031 TF/s on the step's forward shapes).  What it lacks is everything around the K loop that the persistent kernel has (the
// pipeline across tiles, stores draining under the next tile's loop) plus a prefetch deeper than one K-tile — its K-tile lasts half
// as long, so one tile ahead no longer covers the L2 latency (without steady-state requests it runs at 786 TF/s at K = 512) — and the
// fragment reads interleaved with the MFMAs: the first form, with the next k-half's reads in front of the current MFMAs, made the
// compiler wait for lgkmcnt(0) anyway and shuffle 900 v_accvgpr copies through the loop (549-672 TF/s).  Not on the product path.
constexpr int W4_NT = 256, W4_QM = 8, W4_QN = 8;
constexpr int W4_STAGE = 4 * P2_SUB;                       // A_k0 | A_k1 | B_k0 | B_k1
constexpr int W4_CPITCH = 128 + 8;                         // bf16 elements per staged C row of a wave's 128 x 128 part
constexpr int W4_LDS = 4 * 128 * W4_CPITCH * 2;            // 139264 B: the C staging (4 waves) is larger than the two stages (131072)
__global__ __launch_bounds__(W4_NT) void gemm_w4_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: workgroups of one launch run on XCD blockIdx.x % 8; consecutive tiles of an XCD share their A rows
    const int tiles = g.tiles_m * g.tiles_n;
    const int q8 = tiles >> 3, r8 = tiles & 7, xcd = blockIdx.x & 7;
    const int unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tile_m = unit / g.tiles_n, tile_n = unit % g.tiles_n;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
    // each wave requests the pieces the 8-wave layout gives waves w and w + 4
    P2Stage<true> sa0, sa1, sb0, sb1;
    sa0.init(A, g.lda, tile_m * BIG, g.M, 0, wave, lane, 0, 1 << 30, 0);
    sa1.init(A, g.lda, tile_m * BIG, g.M, 0, wave + 4, lane, 0, 1 << 30, 0);
    sb0.init(B, g.ldb, tile_n * BIG, g.N, 0, wave, lane, 0, 1 << 30, 0);
    sb1.init(B, g.ldb, tile_n * BIG, g.N, 0, wave + 4, lane, 0, 1 << 30, 0);
    auto request = [&](char* st, int t) {
#pragma unroll
        for (int kh = 0; kh < 2; kh++) {
            sa0.issue(st + kh * P2_SUB, wave, t, kh);
            sa1.issue(st + kh * P2_SUB, wave + 4, t, kh);
            sb0.issue(st + (2 + kh) * P2_SUB, wave, t, kh);
            sb1.issue(st + (2 + kh) * P2_SUB, wave + 4, t, kh);
        }
    };
    const unsigned offa = q_kc_off(wm * 128, lane), offb = q_kc_off(wn * 128, lane);
    const int nt = g.K / 64;
    request(smem, 0);
    f32x4 acc[W4_QM][W4_QN];
#pragma unroll
    for (int i = 0; i < W4_QM; i++)
#pragma unroll
        for (int j = 0; j < W4_QN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int t = 0; t < nt; t++) {
        const char* cur = smem + (t & 1) * W4_STAGE;
        char* nxt = smem + ((t + 1) & 1) * W4_STAGE;
        // stage t has landed for everybody, and everybody has read stage t - 1 (the other buffer) into registers: refill it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < nt) request(nxt, t + 1);
        bf16x8 fa[2][W4_QM], fb[2][W4_QN];
#pragma unroll
        for (int kh = 0; kh < 2; kh++) {
#pragma unroll
            for (int i = 0; i < W4_QM; i++) fa[kh][i] = q_frag_kc(cur + kh * P2_SUB, offa, i);
#pragma unroll
            for (int j = 0; j < W4_QN; j++) fb[kh][j] = q_frag_kc(cur + (2 + kh) * P2_SUB, offb, j);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; kh++)
#pragma unroll
            for (int i = 0; i < W4_QM; i++)
#pragma unroll
                for (int j = 0; j < W4_QN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kh][j], fa[kh][i], acc[i][j], 0, 0, 0);   // C^T
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: C^T accumulators (lane: row 16 i + (l & 15), columns 16 j + 4 (l >> 4) .. + 3) -> the wave's own LDS part -> 16-byte
    // row-contiguous stores
    __syncthreads();
    bf16_t* cw = reinterpret_cast<bf16_t*>(smem) + wave * 128 * W4_CPITCH;
    const int c16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int j = 0; j < W4_QN; j++) {
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bq = *reinterpret_cast<const f32x4*>(g.bias + tile_n * BIG + wn * 128 + 16 * j + 4 * g4);
#pragma unroll
        for (int i = 0; i < W4_QM; i++) {
            const f32x4 v = acc[i][j] + bq;
            u32x2 w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(cw + (16 * i + c16) * W4_CPITCH + 16 * j + 4 * g4) = w;
        }
    }
    // (a wave reads back only what it wrote: no barrier)
    bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
    const int rr = lane >> 4, cc = lane & 15;             // 4 rows x 16 chunks of 16 bytes per wave-instruction
#pragma unroll 4
    for (int r = 0; r < 128; r += 4) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(cw + (r + rr) * W4_CPITCH + 8 * cc);
        const long row = (long)tile_m * BIG + wm * 128 + r + rr;
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(C + row * g.ldc + tile_n * BIG + wn * 128 + 8 * cc));
    }
}

// MH_GEMM_PP=0 keeps every launch on gemm_big_kernel (A/B switch; default: the ping-pong kernel)
// which main loop the 256 x 256-tile launches use: 0 = gemm_big_kernel (register staging), 1 = gemm_pp_kernel (direct-to-LDS,
// ping-pong), 2 = gemm_pq_kernel (the same, persistent: default).  env MH_GEMM_PP = 0 / 1 / 2, or mh_gemm_select_pp().
static int g_pp = -1;
static int pp_mode() {
    if (g_pp < 0) { const char* e = getenv("MH_GEMM_PP"); g_pp = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2; }
    return g_pp;
}
static bool pp_enabled() { return pp_mode() != 0; }
static int pq_grid(long units) {
    static const int cus = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
    return (int)(units < cus ? units : cus);
}
// the epilogue variant of gemm_pq_kernel for this launch (see its VAR), -1: not one of its cases
static int pq_variant(const GemmArgs& a, bool bf16_tiles, bool part, int epi) {
    const bool relu = a.act == MH_ACT_RELU;
    if (part) return a.alpha == 1.f ? 0 : 1;
    if (bf16_tiles) {
        if (a.accumulate || (epi != 0 && relu)) return -1;
        return relu ? 2 : (a.alpha == 1.f && !a.bias ? 0 : 1);
    }
    if (a.atomic || (epi != 0 && (relu || a.accumulate))) return -1;
    return a.accumulate ? (relu ? -1 : 2) : (relu ? 1 : 0);
}
template <typename K>
static void pp_attr(K kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
}
#define PQ_LAUNCH_VAR_(TC, AKC, BKC, PART, EPI, VAR, s, a, units_, grid)                                  \
    do {                                                                                             \
        static const bool attrq_ = (pp_attr(gemm_pq_kernel<TC, AKC, BKC, PART, EPI, VAR>), true);    \
        (void)attrq_;                                                                                \
        hipLaunchKernelGGL((gemm_pq_kernel<TC, AKC, BKC, PART, EPI, VAR>), dim3(pq_grid(units_)), dim3(NTB), PQ_LDS, s, a, (int)units_, \
                           (int)(grid).x, (int)(grid).y);                                            \
    } while (0)
#define PP_LAUNCH_(TC, AKC, BKC, PART, EPI, grid, s, a)                                              \
    do {                                                                                             \
        const int var_ = pq_variant(a, sizeof(TC) == 2, PART, EPI);                                  \
        const bool pq_ = pp_mode() == 2 && !(a).shared_chip && var_ >= 0;                            \
        gemm_note_variant("%s<%s,%s,%s%s%s>", pq_ ? "gemm_pq_kernel" : "gemm_pp_kernel", gemm_tn<TC>(), \
                          gemm_tf(AKC), gemm_tf(BKC), (PART) ? ",part" : "", (EPI) == 0 ? "" : ((EPI) == 1 ? ",epi1" : ((EPI) == 2 ? ",epi2" : ",epi3"))); \
        if (pq_) {                                                                                   \
            const long units_ = (long)(grid).x * (grid).y * (grid).z;                                \
            if (var_ == 0) PQ_LAUNCH_VAR_(TC, AKC, BKC, PART, EPI, 0, s, a, units_, grid);           \
            else if (var_ == 1) PQ_LAUNCH_VAR_(TC, AKC, BKC, PART, EPI, 1, s, a, units_, grid);      \
            else if constexpr ((EPI) == 0 && !(PART)) PQ_LAUNCH_VAR_(TC, AKC, BKC, PART, EPI, 2, s, a, units_, grid); \
        } else {                                                                                     \
            static const bool attr_ = (pp_attr(gemm_pp_kernel<TC, AKC, BKC, PART, EPI>), true);      \
            (void)attr_;                                                                             \
            hipLaunchKernelGGL((gemm_pp_kernel<TC, AKC, BKC, PART, EPI>), grid, dim3(NTB), PP_LDS, s, a); \
        }                                                                                            \
    } while (0)

template <typename TC>
void launch_big(GemmArgs& a, int akc, int bkc, int batch, hipStream_t s) {
    a.tiles_m = (a.M + BIG - 1) / BIG;     // a ragged last row tile is allowed when A rows are K-contiguous
    a.tiles_n = a.N / BIG;
    dim3 grid(a.tiles_m * a.tiles_n, a.split_k, batch);
    if (pp_enabled()) {
        const long parts_ = (long)a.split_k * batch, mn_ = (long)a.M * a.N;
        const bool partial_ = a.atomic && a.ws && a.M % BIG == 0 && a.ws_floats * 2 >= parts_ * mn_ && ((uintptr_t)a.ws & 15) == 0 && a.vecC &&
                              a.sC1 == 0 && a.sC2 == 0 && parts_ >= 8;
        if (!partial_) a.ws = nullptr;
#define PP_DISPATCH_(PART)                                                                 \
        do {                                                                               \
            if (akc && bkc) PP_LAUNCH_(TC, true, true, PART, 0, grid, s, a);               \
            else if (akc) PP_LAUNCH_(TC, true, false, PART, 0, grid, s, a);                \
            else if (bkc) PP_LAUNCH_(TC, false, true, PART, 0, grid, s, a);                \
            else PP_LAUNCH_(TC, false, false, PART, 0, grid, s, a);                        \
        } while (0)
        if constexpr (sizeof(TC) == 4) {
            if (partial_) {
                PP_DISPATCH_(true);
                launch_fold((const float*)a.ws, (int)parts_, mn_, (float*)a.C, (long)a.ldc, a.N, s);
                return;
            }
        }
        PP_DISPATCH_(false);
#undef PP_DISPATCH_
        return;
    }
    // reduction into one C: partial tiles in the caller's workspace when it is large enough, f32 atomics otherwise
    gemm_note_variant("gemm_big_kernel<%s,%s,%s>", gemm_tn<TC>(), gemm_tf(akc), gemm_tf(bkc));
    const long parts = (long)a.split_k * batch, mn = (long)a.M * a.N;
    const bool partial = a.atomic && a.ws && a.M % BIG == 0 && a.ws_floats * 2 >= parts * mn && ((uintptr_t)a.ws & 15) == 0 && a.vecC &&
                         a.sC1 == 0 && a.sC2 == 0 && parts >= 8;
    if (!partial) a.ws = nullptr;
    if constexpr (sizeof(TC) == 4) {
        if (partial) {
            if (akc && bkc) hipLaunchKernelGGL((gemm_big_kernel<TC, true, true, false, true>), grid, dim3(NTB), 0, s, a);
            else if (akc) hipLaunchKernelGGL((gemm_big_kernel<TC, true, false, false, true>), grid, dim3(NTB), 0, s, a);
            else if (bkc) hipLaunchKernelGGL((gemm_big_kernel<TC, false, true, false, true>), grid, dim3(NTB), 0, s, a);
            else hipLaunchKernelGGL((gemm_big_kernel<TC, false, false, false, true>), grid, dim3(NTB), 0, s, a);
            launch_fold((const float*)a.ws, (int)parts, mn, (float*)a.C, (long)a.ldc, a.N, s);
            return;
        }
    }
    if (akc && bkc) hipLaunchKernelGGL((gemm_big_kernel<TC, true, true>), grid, dim3(NTB), 0, s, a);
    else if (akc) hipLaunchKernelGGL((gemm_big_kernel<TC, true, false>), grid, dim3(NTB), 0, s, a);
    else if (bkc) hipLaunchKernelGGL((gemm_big_kernel<TC, false, true>), grid, dim3(NTB), 0, s, a);
    else hipLaunchKernelGGL((gemm_big_kernel<TC, false, false>), grid, dim3(NTB), 0, s, a);
}

}  // namespace


// experiment entry (round 5): C bf16 [M, N] = A [M, K] B^T ([N, K] weights) + bias on the 4-wave kernel; M, N % 256 == 0, K % 64 == 0
extern "C" int mh_gemm_w4(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                          mh_stream s) {
    MH_REQUIRE(M > 0 && M % BIG == 0 && N % BIG == 0 && K % 64 == 0 && K >= 64 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 &&
                   (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0),
               "mh_gemm_w4: whole 256 x 256 x 64 tiles and 16-byte aligned rows (M=%d N=%d K=%d)", M, N, K);
    GemmArgs a{};
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.tiles_m = M / BIG; a.tiles_n = N / BIG; a.alpha = 1.f;
    static const bool attr = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS), true);
    (void)attr;
    hipLaunchKernelGGL(gemm_w4_kernel, dim3(a.tiles_m * a.tiles_n), dim3(W4_NT), W4_LDS, (hipStream_t)s, a);
    MH_LAUNCH_CHECK("mh_gemm_w4");
    return MH_OK;
}

// tuning switch (tools/bench_gemm_pp.py A/B in one process): 2 = persistent ping-pong kernel (default), 1 = ping-pong kernel,
// 0 = gemm_big_kernel; returns the old value
extern "C" int mh_gemm_select_pp(int mode) {
    const int old = pp_mode();
    if (mode >= 0) g_pp = mode > 2 ? 2 : mode;          // mode < 0: query only
    return old;
}

// e4m3 operands (K contiguous, described in 2-byte units: K, lda, ldb, strides are HALF the byte counts): true when taken
bool gemm_try_big_fp8(GemmArgs& a, int dtC, int batch, hipStream_t s) {
    const bool ok = a.M > BIG && a.N % BIG == 0 && a.K % 64 == 0 && a.k_per_split == a.K && a.split_k == 1 && !a.atomic && !a.accumulate &&
                    a.vecA && a.vecB && a.vecC && !a.R && a.diag == 0.f && a.scale_a && a.scale_b;
    if (!ok || (long)((a.M + BIG - 1) / BIG) * (a.N / BIG) * batch < 128) return false;
    a.tiles_m = (a.M + BIG - 1) / BIG;
    a.tiles_n = a.N / BIG;
    dim3 grid(a.tiles_m * a.tiles_n, 1, batch);
    gemm_note_variant("gemm_big_kernel<%s,true,true,fp8>", dtC == MH_BF16 ? "bf16" : "float");
    if (dtC == MH_BF16) hipLaunchKernelGGL((gemm_big_kernel<bf16_t, true, true, true>), grid, dim3(NTB), 0, s, a);
    else hipLaunchKernelGGL((gemm_big_kernel<float, true, true, true>), grid, dim3(NTB), 0, s, a);
    return true;
}

// fused-epilogue launch (mh_gemm_desc.epi): 0 = launched, otherwise why not (the caller reports MH_EINVAL)
const char* gemm_big_epi(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s) {
    if (!(akc && a.M > BIG && a.N % BIG == 0 && a.K % 64 == 0 && a.split_k == 1 && a.k_per_split == a.K && !a.atomic && !a.accumulate && !a.R &&
          a.diag == 0.f && a.vecA && a.vecB && a.vecC && batch == 1))
        return "shape is not on the 256 x 256 kernel (K-contiguous bf16 A, M > 256, N % 256 == 0, K % 64 == 0, one batch, no split-K / accumulate / R / diag)";
    if (a.a_rpb != 0 && a.a_rpb < BIG) return "a_rows_per_batch must be >= 256";
    if (a.ldc != a.N) return "C must be contiguous (ldc == N)";
    const mh_gemm_epi& e = a.epi;
    a.tiles_m = (a.M + BIG - 1) / BIG;
    a.tiles_n = a.N / BIG;
    dim3 grid(a.tiles_m * a.tiles_n, 1, 1);
#define EPI_LAUNCH_(TC, EPI) \
    do { if (pp_enabled()) { if (bkc) PP_LAUNCH_(TC, true, true, false, EPI, grid, s, a); else PP_LAUNCH_(TC, true, false, false, EPI, grid, s, a); } \
         else if (bkc) hipLaunchKernelGGL((gemm_big_kernel<TC, true, true, false, false, EPI>), grid, dim3(NTB), 0, s, a); \
         else hipLaunchKernelGGL((gemm_big_kernel<TC, true, false, false, false, EPI>), grid, dim3(NTB), 0, s, a); } while (0)
    switch (e.kind) {
    case MH_EPI_DROPADD:
        if (dtC != MH_F32 || !e.resid || !(e.p >= 0.f && e.p < 1.f) || (e.offset & 7) || ((uintptr_t)e.resid & 15)) return "DROPADD: f32 C, a 16-byte aligned residual, 0 <= p < 1, offset % 8 == 0";
        EPI_LAUNCH_(float, MH_EPI_DROPADD);
        return nullptr;
    case MH_EPI_MASKPOS:
        if (dtC != MH_F32 || !e.mask || !e.token || !e.pos || e.rows_per_batch < BIG || e.first < 0) return "MASKPOS: f32 C, mask / token / pos, rows_per_batch >= 256";
        if (((uintptr_t)e.token & 15) || ((uintptr_t)e.pos & 15)) return "MASKPOS: token / pos must be 16-byte aligned";
        EPI_LAUNCH_(float, MH_EPI_MASKPOS);
        return nullptr;
    case MH_EPI_SQERR:
        if (dtC != MH_BF16 || !e.mask || !e.tgt || !e.sq || e.rows_per_batch <= 0 || e.rows_per_batch % BIG || a.M % e.rows_per_batch || a.act != MH_ACT_NONE) return "SQERR: bf16 C, mask / tgt / sq, rows_per_batch % 256 == 0";
        if (((uintptr_t)e.tgt & 15) || e.tgt_bs % 4) return "SQERR: tgt must be 16-byte aligned";
        EPI_LAUNCH_(bf16_t, MH_EPI_SQERR);
        return nullptr;
    default:
        return "unknown epilogue kind";
    }
#undef EPI_LAUNCH_
}

// plain bf16 product whose A and / or C rows are row windows of larger batches (mh_gemm_desc.a_rows_per_batch / c_rows_per_batch):
// 0 = launched, otherwise why not.  Only the direct-to-LDS kernels take row windows without a fused epilogue.
const char* gemm_big_window(GemmArgs& a, int akc, int bkc, int batch, hipStream_t s) {
    if (!pp_enabled()) return "row windows without an epilogue need the direct-to-LDS kernels (MH_GEMM_PP != 0)";
    if (!(akc && a.M > BIG && a.M % BIG == 0 && a.N % BIG == 0 && a.K % 64 == 0 && a.split_k == 1 && a.k_per_split == a.K && !a.atomic && !a.accumulate &&
          !a.R && a.diag == 0.f && a.vecA && a.vecB && a.vecC && batch == 1))
        return "shape is not on the 256 x 256 kernel (K-contiguous bf16 A, M % 256 == 0, N % 256 == 0, K % 64 == 0, one batch, no split-K / accumulate / R / diag)";
    if ((a.a_rpb != 0 && a.a_rpb < BIG) || (a.c_rpb != 0 && a.c_rpb < BIG)) return "rows_per_batch must be >= 256";
    a.epi.kind = MH_EPI_NONE;
    a.tiles_m = a.M / BIG;
    a.tiles_n = a.N / BIG;
    dim3 grid(a.tiles_m * a.tiles_n, 1, 1);
    if (bkc) PP_LAUNCH_(bf16_t, true, true, false, 0, grid, s, a);
    else PP_LAUNCH_(bf16_t, true, false, false, 0, grid, s, a);
    return nullptr;
}

// true when the large-tile kernel took the launch
bool gemm_try_big_bf16(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s) {
    const bool shape_ok = (a.M % BIG == 0 || (akc && a.M > BIG)) && a.N % BIG == 0 && a.K % 64 == 0 && a.k_per_split % 64 == 0;
    if (!shape_ok || !a.vecA || !a.vecB || !a.vecC || a.R || a.diag != 0.f) return false;
    if (a.atomic && (a.bias || a.act != MH_ACT_NONE || dtC != MH_F32)) return false;
    // accumulating into a bf16 C: these kernels stage the product as bf16 before they add (two roundings, torch's `c += a @ b`), the
    // 128 x 128 kernel adds in f32 and rounds once (addmm): one semantics for the entry point, the latter (no caller in the model)
    if (dtC == MH_BF16 && a.accumulate) return false;
    const long wgs = (long)((a.M + BIG - 1) / BIG) * (a.N / BIG) * a.split_k * batch;
    if (wgs < 128) return false;            // too few workgroups for one per CU: the 128 x 128 kernel spreads better
    if (dtC == MH_BF16) launch_big<bf16_t>(a, akc, bkc, batch, s);
    else launch_big<float>(a, akc, bkc, batch, s);
    return true;
}
