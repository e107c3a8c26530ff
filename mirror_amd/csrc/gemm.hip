// MFMA GEMM for gfx950: one templated kernel behind mh_gemm().
//
//   block  = 256 threads = 4 waves (2 x 2), wave tile = (32*WM) x (32*WN), block tile (64*WM) x (64*WN)
//   MMA    = bf16: v_mfma_f32_32x32x16_bf16, BK = 32   |   f32: v_mfma_f32_32x32x2_f32 (exact), BK = 16
//   staging: global -> registers (16-B loads, prefetched one K-tile ahead) -> LDS (double buffered),
//            f32 operands are rounded to bf16 on the way into LDS when MMA = bf16
//   LDS images (bank maths from MI355X_MICROARCH.md §LDS):
//     K-contiguous operand  : [rows][BK]  pitch 40 bf16 / 20 f32  -> ds_read_b128 fragments, conflict free
//     K-strided operand bf16: [BK][rows]  pitch rows+32           -> ds_read_b64_tr_b16 (hardware transpose)
//     K-strided operand f32 : [BK][rows]  pitch rows+4            -> ds_read_b32, lanes consecutive
//   f32 MMA k-order trick: lane half h supplies k = 8h + s at MFMA step s for BOTH operands, so each
//   lane reads 8 consecutive k with two ds_read_b128 (any k permutation shared by A and B is legal).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
};

template <int MMA, bool KC, int ROWS>
struct TileGeom {
    static constexpr int BK = MMA ? 32 : 16;
    static constexpr int ESZ = MMA ? 2 : 4;
    static constexpr int LROWS = KC ? ROWS : BK;
    static constexpr int PITCH = KC ? (MMA ? 40 : 20) : (MMA ? ROWS + 32 : ROWS + 4);
    static constexpr int BYTES = LROWS * PITCH * ESZ;
};

// ------------------------------------------------------------------ global -> regs -> LDS
template <int MMA, typename TG, bool KC, int ROWS>
struct Stager {
    using G = TileGeom<MMA, KC, ROWS>;
    static constexpr int VEC = 16 / (int)sizeof(TG);
    static constexpr int CONTIG = KC ? G::BK : ROWS;
    static constexpr int CPR = CONTIG / VEC;
    static constexpr int NCH = G::LROWS * CPR / 256;
    static_assert(G::LROWS * CPR % 256 == 0, "tile must split evenly over 256 threads");
    uint4 regs[NCH];

    __device__ __forceinline__ void load(const TG* __restrict__ base, long ld, int tile0, int dim, int k0,
                                         int kend, bool vec_ok, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int cid = tid + i * 256;
            const int r = cid / CPR, c = cid % CPR;
            int gm, gk;
            long off;
            if (KC) { gm = tile0 + r; gk = k0 + c * VEC; off = (long)gm * ld + gk; }
            else    { gk = k0 + r; gm = tile0 + c * VEC; off = (long)gk * ld + gm; }
            const bool row_ok = KC ? (gm < dim) : (gk < kend);
            const int cstart = KC ? gk : gm;
            const int climit = KC ? kend : dim;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row_ok) {
                if (vec_ok && cstart + VEC <= climit) {
                    v = *reinterpret_cast<const uint4*>(base + off);
                } else {
                    TG tmp[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; e++) tmp[e] = (cstart + e < climit) ? base[off + e] : (TG)0;
                    v = *reinterpret_cast<uint4*>(tmp);
                }
            }
            regs[i] = v;
        }
    }

    __device__ __forceinline__ void store(char* tile, int tid) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int cid = tid + i * 256;
            const int r = cid / CPR, c = cid % CPR;
            if constexpr ((int)sizeof(TG) == G::ESZ) {
                *reinterpret_cast<uint4*>(tile + (r * G::PITCH + c * VEC) * G::ESZ) = regs[i];
            } else {  // f32 in HBM -> bf16 in LDS
                const float* f = reinterpret_cast<const float*>(&regs[i]);
                uint2 p;
                p.x = (uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
                p.y = (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
                *reinterpret_cast<uint2*>(tile + (r * G::PITCH + c * 4) * 2) = p;
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

template <typename TC> __device__ __forceinline__ void c_store(TC* p, float v, int accumulate, int atomic);
template <> __device__ __forceinline__ void c_store<float>(float* p, float v, int accumulate, int atomic) {
    if (atomic) atomicAdd(p, v);
    else if (accumulate) *p += v;
    else *p = v;
}
template <> __device__ __forceinline__ void c_store<bf16_t>(bf16_t* p, float v, int accumulate, int) {
    if (accumulate) v += bf2f(*p);
    *p = f2bf(v);
}

template <int MMA, typename TA, typename TB, typename TC, bool AKC, bool BKC, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    using GA = TileGeom<MMA, AKC, BM>;
    using GB = TileGeom<MMA, BKC, BN>;
    constexpr int BK = GA::BK;
    __shared__ __attribute__((aligned(16))) char smem[2 * (GA::BYTES + GB::BYTES)];
    constexpr int STAGE = GA::BYTES + GB::BYTES;  // stage s: A at smem + s*STAGE, B right behind it

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int z = blockIdx.z;
    const int b1 = z / g.batch2, b2 = z % g.batch2;
    const TA* A = reinterpret_cast<const TA*>(g.A) + b1 * g.sA1 + b2 * g.sA2;
    const TB* B = reinterpret_cast<const TB*>(g.B) + b1 * g.sB1 + b2 * g.sB2;
    TC* C = reinterpret_cast<TC*>(g.C) + b1 * g.sC1 + b2 * g.sC2;
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

    Stager<MMA, TA, AKC, BM> sa;
    Stager<MMA, TB, BKC, BN> sb;
    if (nt > 0) {
        sa.load(A, g.lda, tile_m * BM, g.M, kbeg, kend, g.vecA, tid);
        sb.load(B, g.ldb, tile_n * BN, g.N, kbeg, kend, g.vecB, tid);
        sa.store(smem, tid);
        sb.store(smem + GA::BYTES, tid);
    }
    __syncthreads();

    for (int t = 0; t < nt; t++) {
        const int cur = t & 1;
        const bool more = (t + 1 < nt);
        if (more) {
            const int k0 = kbeg + (t + 1) * BK;
            sa.load(A, g.lda, tile_m * BM, g.M, k0, kend, g.vecA, tid);
            sb.load(B, g.ldb, tile_n * BN, g.N, k0, kend, g.vecB, tid);
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
        if (more) {
            sa.store(smem + (cur ^ 1) * STAGE, tid);
            sb.store(smem + (cur ^ 1) * STAGE + GA::BYTES, tid);
        }
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int r = lane & 31, hh = lane >> 5;
    const int atomic = g.atomic;
    const bool lead = (split == 0);
#pragma unroll
    for (int j = 0; j < WN; j++) {
        const int col = tile_n * BN + wn * WN * 32 + j * 32 + r;
        if (col >= g.N) continue;
        const float bias = (g.bias && lead) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; i++) {
            const int rbase = tile_m * BM + wm * WM * 32 + i * 32 + 4 * hh;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = rbase + (reg & 3) + 8 * (reg >> 2);
                if (row >= g.M) continue;
                float v = g.alpha * acc[i][j][reg] + bias;
                if (lead && row == col) v += g.diag;
                if (g.act == MH_ACT_RELU) v = fmaxf(v, 0.f);
                else if (g.act == MH_ACT_GELU) v = gelu_f(v);
                c_store<TC>(C + (long)row * g.ldc + col, v, g.accumulate, atomic);
            }
        }
    }
}

// ------------------------------------------------------------------ host dispatch
template <int MMA, typename TA, typename TB, typename TC, bool AKC, bool BKC>
static void launch_w(const GemmArgs& a, int batch, hipStream_t s) {
    if (a.N <= 64) {
        dim3 grid(mh_cdiv(a.M, 128) * mh_cdiv(a.N, 64), a.split_k, batch);
        hipLaunchKernelGGL((gemm_kernel<MMA, TA, TB, TC, AKC, BKC, 2, 1>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid(mh_cdiv(a.M, 128) * mh_cdiv(a.N, 128), a.split_k, batch);
        hipLaunchKernelGGL((gemm_kernel<MMA, TA, TB, TC, AKC, BKC, 2, 2>), grid, dim3(256), 0, s, a);
    }
}

template <int MMA, typename TA, typename TB, typename TC>
static void launch_l(const GemmArgs& a, int akc, int bkc, int batch, hipStream_t s) {
    if (akc && bkc) launch_w<MMA, TA, TB, TC, true, true>(a, batch, s);
    else if (akc && !bkc) launch_w<MMA, TA, TB, TC, true, false>(a, batch, s);
    else if (!akc && bkc) launch_w<MMA, TA, TB, TC, false, true>(a, batch, s);
    else launch_w<MMA, TA, TB, TC, false, false>(a, batch, s);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int mh_gemm(const mh_gemm_desc* d, mh_stream stream) {
    MH_REQUIRE(d && d->A && d->B && d->C, "mh_gemm: null pointer");
    MH_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "mh_gemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    MH_REQUIRE(d->batch1 >= 1 && d->batch2 >= 1 && (long)d->batch1 * d->batch2 <= 65535, "mh_gemm: bad batch");
    MH_REQUIRE(d->dtA == d->dtB, "mh_gemm: dtA must equal dtB");
    MH_REQUIRE(d->mma == MH_BF16 || (d->dtA == MH_F32 && d->dtC == MH_F32), "mh_gemm: f32 MMA needs f32 operands");
    const int split = d->split_k < 1 ? 1 : d->split_k;
    MH_REQUIRE(split == 1 || (d->accumulate && d->dtC == MH_F32 && d->act == MH_ACT_NONE),
               "mh_gemm: split_k>1 needs accumulate=1, f32 C, no activation");
    MH_REQUIRE(split <= 65535, "mh_gemm: split_k too large");
    GemmArgs a;
    a.A = d->A; a.B = d->B; a.C = d->C; a.bias = d->bias;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.sA1 = d->sA1; a.sA2 = d->sA2; a.sB1 = d->sB1; a.sB2 = d->sB2; a.sC1 = d->sC1; a.sC2 = d->sC2;
    a.batch2 = d->batch2;
    a.alpha = d->alpha; a.diag = d->diag; a.act = d->act; a.accumulate = d->accumulate;
    const int BK = d->mma == MH_BF16 ? 32 : 16;
    int kps = mh_cdiv(mh_cdiv(d->K, split), BK) * BK;
    a.k_per_split = kps;
    a.split_k = mh_cdiv(d->K, kps);  // every split has work
    const int esz = d->dtA == MH_F32 ? 4 : 2;
    const int vec = 16 / esz;
    a.vecA = aligned16(d->A) && d->lda % vec == 0 && d->sA1 % vec == 0 && d->sA2 % vec == 0;
    a.vecB = aligned16(d->B) && d->ldb % vec == 0 && d->sB1 % vec == 0 && d->sB2 % vec == 0;
    const int batch = d->batch1 * d->batch2;
    a.atomic = (a.split_k > 1) || (d->accumulate && batch > 1 && d->sC1 == 0 && d->sC2 == 0);
    MH_REQUIRE(!a.atomic || (d->dtC == MH_F32 && d->act == MH_ACT_NONE && d->accumulate),
               "mh_gemm: atomic accumulation (split-K / batch broadcast into C) needs f32 C, accumulate=1, no activation");
    hipStream_t s = (hipStream_t)stream;
    if (d->mma == MH_F32) {
        launch_l<0, float, float, float>(a, d->a_kc, d->b_kc, batch, s);
    } else if (d->dtA == MH_BF16) {
        if (d->dtC == MH_BF16) launch_l<1, bf16_t, bf16_t, bf16_t>(a, d->a_kc, d->b_kc, batch, s);
        else launch_l<1, bf16_t, bf16_t, float>(a, d->a_kc, d->b_kc, batch, s);
    } else {
        if (d->dtC == MH_BF16) launch_l<1, float, float, bf16_t>(a, d->a_kc, d->b_kc, batch, s);
        else launch_l<1, float, float, float>(a, d->a_kc, d->b_kc, batch, s);
    }
    MH_LAUNCH_CHECK("mh_gemm");
    return MH_OK;
}
