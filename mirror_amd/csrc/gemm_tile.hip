// 192 x 384 x 64 block tile for BATCHED square-ish products whose sides are multiples of 384 — the Moore-Penrose iteration
// ([3P] moore_penrose_iter_pinv, called at models/mirror.py:312) at the reference template's geometry: embed_dim 768 ->
// m = 384 landmarks (configs/pretrain/mirror.template.yaml:27-31).  A 384 x 384 bf16 operand image is 288 KiB, so the
// one-launch chain of pinv_panel.hip (one 128 KiB image per CU, m = 256) does not exist there; the iteration runs as
// launches, and on the 128 x 128 kernel each of them was 9 small tiles per matrix at 110-210 TFLOP/s (fixed costs of a
// 6-step K loop), 70 % of the template step.
//
//   768 threads = 12 waves (2 x 6), wave tile 96 x 64 = 3 x 2 MFMA blocks (v_mfma_f32_32x32x16_bf16), 96 accumulator
//   registers, three waves per SIMD;  2 workgroups per 384 x 384 matrix -> B h = 128 matrices = 256 workgroups = one per
//   CU, ONE round.  Same staging scheme as gemm_big.hip (global -> registers one K-tile ahead -> double-buffered LDS
//   written after the barrier).  LDS: K-strided operands use the padded [64][rows + 32] image + ds_read_b64_tr_b16 of
//   gemm_kernel.h; K-contiguous operands use an UNPADDED [rows][64] image with a 16-byte-chunk XOR swizzle (the padded
//   form of two K-contiguous operands would need 166 KiB).
//   Epilogue: C (+)= alpha acc + diag I + rcoef R, 32 rows at a time through an f32 LDS tile; R may be bf16 beside an f32 C,
//   and a bf16 copy of the final C can be written in the same pass (the next product's operand: no cast launch).
#include "gemm_kernel.h"

namespace {

constexpr int TM = 192, TN = 384, NTT = 768, TWM = 3, TWN = 2, TBK = 64;

// K-contiguous operand image: [rows][64] bf16, 128-byte rows, chunk c (16 B) of row r at slot c ^ swz(r)
__device__ __forceinline__ int kc_swz(int r) { return (r & 7) ^ ((r >> 3) & 1); }

template <bool KC, int ROWS>
struct TGeo {
    using G = TileGeom<1, KC, ROWS>;
    static constexpr int BYTES = KC ? ROWS * TBK * 2 : G::BYTES;
    static constexpr int NCH = ROWS * 8 / NTT;        // 16-byte chunks per thread and K-tile (both layouts: ROWS * 64 * 2 / 16 / 768)
};

template <int ROWS> using TRegs = u32x4[ROWS * 8 / NTT];

// kend: the operand's K extent.  A K that is not a multiple of 64 (the 96-wide heads of the template geometry) ends in a partly
// empty K-tile: chunks at k >= kend read as zeros (K % 8 == 0, so a 16-byte chunk is all in or all out).
template <bool KC, int ROWS>
__device__ __forceinline__ void t_load(TRegs<ROWS>& regs, const bf16_t* __restrict__ base, long ld, int tile0, int k0, int kend, int tid) {
    constexpr int NCH = TGeo<KC, ROWS>::NCH;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int cid = tid + i * NTT;
        long off;
        bool in;
        if (KC) { const int r = cid >> 3, c = cid & 7; off = (long)(tile0 + r) * ld + k0 + 8 * c; in = k0 + 8 * c < kend; }
        else { constexpr int CPR = ROWS / 8; const int r = cid / CPR, c = cid % CPR; off = (long)(k0 + r) * ld + tile0 + 8 * c; in = k0 + r < kend; }
        regs[i] = in ? *reinterpret_cast<const u32x4*>(base + off) : u32x4{0u, 0u, 0u, 0u};
    }
}
template <bool KC, int ROWS>
__device__ __forceinline__ void t_store(const TRegs<ROWS>& regs, char* tile, int tid) {
    constexpr int NCH = TGeo<KC, ROWS>::NCH;
    using G = TileGeom<1, KC, ROWS>;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int cid = tid + i * NTT;
        if (KC) { const int r = cid >> 3, c = cid & 7; *reinterpret_cast<u32x4*>(tile + r * 128 + ((c ^ kc_swz(r)) << 4)) = regs[i]; }
        else { constexpr int CPR = ROWS / 8; const int r = cid / CPR, c = cid % CPR; *reinterpret_cast<u32x4*>(tile + (r * G::PITCH + 8 * c) * 2) = regs[i]; }
    }
}
// 8 consecutive k (k0 + 8 (lane >> 5) ..) of tile row row0 + (lane & 31)
template <bool KC, int ROWS>
__device__ __forceinline__ bf16x8 t_frag(const char* tile, int row0, int k0, int lane) {
    if constexpr (KC) {
        const int r = row0 + (lane & 31), c = (k0 >> 3) + (lane >> 5);
        return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ kc_swz(r)) << 4));
    } else {
        return frag_bf16<false, ROWS>(tile, row0, k0, lane);
    }
}

template <typename TC, bool AKC, bool BKC>
__global__ __launch_bounds__(NTT) void gemm_tile_kernel(GemmArgs g, bf16_t* __restrict__ C2, int r_bf16) {
    using GA = TGeo<AKC, TM>;
    using GB = TGeo<BKC, TN>;
    constexpr int STAGE = GA::BYTES + GB::BYTES;
    static_assert(2 * STAGE <= 160 * 1024, "two stages must fit the 160 KiB of LDS");
    static_assert(32 * (TN + 4) * 4 <= 2 * STAGE, "epilogue tile must fit the staging LDS");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / 6, wn = wave % 6;
    // XCD-aware order (round 5): workgroup L runs on XCD L % 8.  The M-tiles of one matrix read the same B (all of it when N = 384: 295 KB at
    // K = 384), so they take ids 8 apart — same XCD, neighbours in time — and the second one finds B in that XCD's L2 instead of fetching
    // it from the fabric again (the pairs 2z, 2z + 1 of the plain order sat on different XCDs: 151 MB per 384-cubed product for 113)
    int tile_m, tile_n, z;
    if (g.tiles_n == 1 && g.tiles_m > 1 && gridDim.z % 8 == 0) {
        const int L = blockIdx.x + gridDim.x * blockIdx.z, T = g.tiles_m;
        z = (L / (8 * T)) * 8 + (L % 8);
        tile_m = (L / 8) % T;
        tile_n = 0;
    } else {
        tile_m = blockIdx.x / g.tiles_n; tile_n = blockIdx.x % g.tiles_n; z = blockIdx.z;
    }
    const int b1 = z / g.batch2, b2 = z % g.batch2;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + b1 * g.sA1 + b2 * g.sA2;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B) + b1 * g.sB1 + b2 * g.sB2;
    const long coff = b1 * g.sC1 + b2 * g.sC2;
    // K loop over kseg operand pairs of K each (a sum of products in ONE accumulator: the f32 C never goes through HBM in between)
    const int nts = (g.K + TBK - 1) / TBK, nt = nts * max(g.kseg, 1);
    auto a_of = [&](int t) { return A + (long)(t / nts) * g.sAk; };
    auto b_of = [&](int t) { return B + (long)(t / nts) * g.sBk; };

    f32x16 acc[TWM][TWN];
#pragma unroll
    for (int i = 0; i < TWM; i++)
#pragma unroll
        for (int j = 0; j < TWN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    u32x4 ra[GA::NCH], rb[GB::NCH];
    t_load<AKC, TM>(ra, A, g.lda, tile_m * TM, 0, g.K, tid);
    t_load<BKC, TN>(rb, B, g.ldb, tile_n * TN, 0, g.K, tid);
    t_store<AKC, TM>(ra, smem, tid);
    t_store<BKC, TN>(rb, smem + GA::BYTES, tid);
    if (nt > 1) {
        t_load<AKC, TM>(ra, a_of(1), g.lda, tile_m * TM, (1 % nts) * TBK, g.K, tid);
        t_load<BKC, TN>(rb, b_of(1), g.ldb, tile_n * TN, (1 % nts) * TBK, g.K, tid);
    }
    __syncthreads();
    for (int t = 0; t < nt; t++) {
        const int cur = t & 1;
        if (t + 1 < nt) {      // written AFTER the barrier that freed the other stage, re-issued at once (gemm_big.hip)
            t_store<AKC, TM>(ra, smem + (cur ^ 1) * STAGE, tid);
            t_store<BKC, TN>(rb, smem + (cur ^ 1) * STAGE + GA::BYTES, tid);
            if (t + 2 < nt) {
                t_load<AKC, TM>(ra, a_of(t + 2), g.lda, tile_m * TM, ((t + 2) % nts) * TBK, g.K, tid);
                t_load<BKC, TN>(rb, b_of(t + 2), g.ldb, tile_n * TN, ((t + 2) % nts) * TBK, g.K, tid);
            }
        }
        const char* at = smem + cur * STAGE;
        const char* bt = at + GA::BYTES;
#pragma unroll
        for (int ks = 0; ks < TBK; ks += 16) {
            bf16x8 af[TWM], bfr[TWN];
#pragma unroll
            for (int i = 0; i < TWM; i++) af[i] = t_frag<AKC, TM>(at, wm * TWM * 32 + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < TWN; j++) bfr[j] = t_frag<BKC, TN>(bt, wn * TWN * 32 + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < TWM; i++)
#pragma unroll
                for (int j = 0; j < TWN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- epilogue: 32 rows at a time (row block i of the wave row wm == h) through an f32 tile [32][TN + 4]
    constexpr int PITCH = TN + 4;
    float* tl = reinterpret_cast<float*>(smem);
    const int r32 = lane & 31, hh = lane >> 5;
    TC* C = reinterpret_cast<TC*>(g.C) + coff;
#pragma unroll 1
    for (int pass = 0; pass < 2 * TWM; pass++) {
        const int h = pass / TWM, i = pass % TWM;
        if (wm == h) {
#pragma unroll
            for (int ii = 0; ii < TWM; ii++)
                if (ii == i)
#pragma unroll
                    for (int j = 0; j < TWN; j++)
#pragma unroll
                        for (int reg = 0; reg < 16; reg++)
                            tl[(4 * hh + (reg & 3) + 8 * (reg >> 2)) * PITCH + wn * TWN * 32 + j * 32 + r32] = acc[ii][j][reg];
        }
        __syncthreads();
        const int row0 = tile_m * TM + h * (TWM * 32) + i * 32;
        if (g.row_softmax) {
            // N == TN: the f32 tile holds 32 whole rows.  24 threads per row, 16 columns each; row maxima and sums meet in a
            // [32][24] scratch behind the tile.  The probabilities leave as bf16 (what the softmax launch behind an f32 logits
            // GEMM wrote): the 453 MB f32 round trip of the template's sim1 never happens.
            float* red = tl + 32 * PITCH;
            const int lr = tid / 24, part = tid % 24;
            f32x4 v[4];
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                v[u] = g.alpha * *reinterpret_cast<const f32x4*>(tl + lr * PITCH + 16 * part + 4 * u);
                mx = fmaxf(fmaxf(fmaxf(mx, v[u][0]), fmaxf(v[u][1], v[u][2])), v[u][3]);
            }
            if (g.row_softmax == 2) {
                // softmax BACKWARD: the product is dP, R holds the probabilities P (bf16, laid out like C): dS = P o (dP - sum_row P dP)
                const bf16_t* prow = reinterpret_cast<const bf16_t*>(g.R) + coff + (long)(row0 + lr) * g.ldc + 16 * part;
                f32x4 y[4];
                float dot = 0.f;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    y[u] = ld4(prow + 4 * u);
                    dot += y[u][0] * v[u][0] + y[u][1] * v[u][1] + y[u][2] * v[u][2] + y[u][3] * v[u][3];
                }
                red[lr * 24 + part] = dot;
                __syncthreads();
                dot = 0.f;
#pragma unroll
                for (int u = 0; u < 24; u++) dot += red[lr * 24 + u];
                bf16_t* drow = reinterpret_cast<bf16_t*>(g.C) + coff + (long)(row0 + lr) * g.ldc + 16 * part;
#pragma unroll
                for (int u = 0; u < 4; u++) st4(drow + 4 * u, y[u] * (v[u] - dot));
                __syncthreads();
                continue;
            }
            red[lr * 24 + part] = mx;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 24; u++) mx = fmaxf(mx, red[lr * 24 + u]);
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int e = 0; e < 4; e++) { v[u][e] = __expf(v[u][e] - mx); sum += v[u][e]; }
            __syncthreads();
            red[lr * 24 + part] = sum;
            __syncthreads();
            sum = 0.f;
#pragma unroll
            for (int u = 0; u < 24; u++) sum += red[lr * 24 + u];
            const float inv = 1.f / sum;
            bf16_t* crow = reinterpret_cast<bf16_t*>(g.C) + coff + (long)(row0 + lr) * g.ldc + 16 * part;
#pragma unroll
            for (int u = 0; u < 4; u++) st4(crow + 4 * u, v[u] * inv);
            __syncthreads();
            continue;
        }
        constexpr int QPR = TN / 4;                       // quads per tile row
#pragma unroll
        for (int u = 0; u < 32 * QPR / NTT; u++) {
            const int q = tid + u * NTT, lr = q / QPR, lc = 4 * (q % QPR);
            const int grow = row0 + lr, gcol = tile_n * TN + lc;
            f32x4 v = g.alpha * *reinterpret_cast<const f32x4*>(tl + lr * PITCH + lc);
            if (g.diag != 0.f && grow >= gcol && grow < gcol + 4) v[grow - gcol] += g.diag;
            const long idx = (long)grow * g.ldc + gcol;
            if (g.R) {
                // R in C's type, or (r_bf16 = 1) bf16 beside an f32 C, or (2) f32 beside a bf16 C
                if (r_bf16 == 2 || (!r_bf16 && sizeof(TC) == 4)) v += g.rcoef * ld4(reinterpret_cast<const float*>(g.R) + coff + idx);
                else v += g.rcoef * ld4(reinterpret_cast<const bf16_t*>(g.R) + coff + idx);
            }
            if (g.accumulate) v += ld4(C + idx);
            st4(C + idx, v);
            if (C2) st4(C2 + coff + idx, v);
        }
        __syncthreads();
    }
}

}  // namespace

// true when the tile kernel took the launch (bf16 operands only; no bias / activation / split-K)
bool gemm_try_tile384(GemmArgs& a, int akc, int bkc, int dtC, int batch, void* c2, int r_bf16, hipStream_t s) {
    const bool ok = a.M % TM == 0 && a.N % TN == 0 && a.K % 8 == 0 && a.split_k == 1 && !a.atomic && !a.bias &&
                    a.act == MH_ACT_NONE && a.vecA && a.vecB && a.vecC && !(akc == 0 && bkc == 1);
    if (!ok) return false;
    if (a.row_softmax && !(a.N == TN && dtC == MH_BF16 && (a.row_softmax == 2) == (a.R != nullptr) && !a.accumulate && a.diag == 0.f && !c2)) return false;
    const long wgs = (long)(a.M / TM) * (a.N / TN) * batch;
    if (wgs < 64 && !c2 && !r_bf16 && a.kseg <= 1 && !a.row_softmax) return false;      // a few tiles: the 128 x 128 kernel spreads better
    a.tiles_m = a.M / TM;
    a.tiles_n = a.N / TN;
    dim3 grid(a.tiles_m * a.tiles_n, 1, batch);
#define TILE_(TC, AK, BK_) do { gemm_note_variant("gemm_tile_kernel<%s,%s,%s>", gemm_tn<TC>(), gemm_tf(AK), gemm_tf(BK_)); \
        hipLaunchKernelGGL((gemm_tile_kernel<TC, AK, BK_>), grid, dim3(NTT), 0, s, a, (bf16_t*)c2, r_bf16); } while (0)
    if (dtC == MH_BF16) {
        if (akc && bkc) TILE_(bf16_t, true, true); else if (akc) TILE_(bf16_t, true, false); else TILE_(bf16_t, false, false);
    } else {
        if (akc && bkc) TILE_(float, true, true); else if (akc) TILE_(float, true, false); else TILE_(float, false, false);
    }
#undef TILE_
    return true;
}
