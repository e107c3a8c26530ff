// Moore-Penrose iteration ([3P] moore_penrose_iter_pinv, called at models/mirror.py:312) as ONE launch per pass.
//
// The iteration  z <- 1/4 z (13I - P (15I - P (7I - P))),  P = X z  is 24 (forward) / 48 (backward) dependent m x m
// GEMMs per (batch, head).  As separate launches each 256^3 problem is latency-bound (4 K-steps, ~30 us each,
// 25 % of the whole training step).  Here one 1024-thread workgroup owns one (b, h): it keeps walking the chain,
// every product is a full m x m tile (16 waves as 4 x 4, wave tile m/4 x m/4, v_mfma_f32_32x32x16_bf16; 64
// accumulator registers per lane so that four waves per SIMD fit the register file without spilling), operands
// are the bf16 matrices the same workgroup stored a moment ago (L2-resident, written once -> no stale L1 lines),
// and sums of products (dP = -dT3 T2^T + dT2 P^T + P^T dT2 - 7 dT2, ...) simply continue the K loop in the same
// accumulators instead of read-modify-writing memory.
//
// Layout: every matrix is contiguous [m][m] bf16.  saved[k] = {z_k, P_k, T2_k, T3_k}, work[k] = {dT3, dT2, dP, dz_k}.
#include "gemm_kernel.h"

#define PC_NT 1024

template <int MM>
struct ChainGeom {
    static constexpr int WM = MM / 128, WN = MM / 128;           // 32x32 MFMA tiles per wave (wave tile MM/4 x MM/4)
    using GA_KC = TileGeom<1, true, MM>;
    using GA_KS = TileGeom<1, false, MM>;
    static constexpr int OPB = GA_KC::BYTES > GA_KS::BYTES ? GA_KC::BYTES : GA_KS::BYTES;   // one operand tile
    static constexpr int STAGE = 2 * OPB;
    static constexpr int EPI = MM * (MM + 8) * 2;                 // bf16 epilogue image
    static constexpr int SMEM = 2 * STAGE > EPI ? 2 * STAGE : EPI;
};

// acc += op(A) . op(B) over K = MM.  AKC: A(m,k) = A[m*MM + k] else A[k*MM + m]; BKC: B(k,n) = B[n*MM + k] else B[k*MM + n]
template <int MM, bool AKC, bool BKC>
__device__ __forceinline__ void chain_mm(f32x16 (&acc)[ChainGeom<MM>::WM][ChainGeom<MM>::WN], const bf16_t* __restrict__ A,
                                         const bf16_t* __restrict__ B, char* smem, int tid) {
    using CG = ChainGeom<MM>;
    constexpr int WM = CG::WM, WN = CG::WN, BK = 64;
    using SA = Stager<1, bf16_t, AKC, MM, true, PC_NT>;
    using SB = Stager<1, bf16_t, BKC, MM, true, PC_NT>;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    u32x4 ra[SA::NCH], rb[SB::NCH];
    SA::load(ra, A, MM, 0, MM, 0, MM, true, tid);
    SB::load(rb, B, MM, 0, MM, 0, MM, true, tid);
    SA::store(ra, smem, tid);
    SB::store(rb, smem + CG::OPB, tid);
    __syncthreads();
    constexpr int nt = MM / BK;
#pragma unroll 1
    for (int t = 0; t < nt; t++) {
        const int cur = t & 1;
        const bool more = (t + 1 < nt);
        if (more) {
            SA::load(ra, A, MM, 0, MM, (t + 1) * BK, MM, true, tid);
            SB::load(rb, B, MM, 0, MM, (t + 1) * BK, MM, true, tid);
        }
        const char* at = smem + cur * CG::STAGE;
        const char* bt = at + CG::OPB;
#pragma unroll
        for (int ks = 0; ks < BK; ks += 16) {
            bf16x8 af[WM], bfr[WN];
#pragma unroll
            for (int i = 0; i < WM; i++) af[i] = frag_bf16<AKC, MM>(at, wm * WM * 32 + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < WN; j++) bfr[j] = frag_bf16<BKC, MM>(bt, wn * WN * 32 + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < WM; i++)
#pragma unroll
                for (int j = 0; j < WN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            SA::store(ra, smem + (cur ^ 1) * CG::STAGE, tid);
            SB::store(rb, smem + (cur ^ 1) * CG::STAGE + CG::OPB, tid);
        }
        __syncthreads();
    }
}

template <int MM>
__device__ __forceinline__ void chain_zero(f32x16 (&acc)[ChainGeom<MM>::WM][ChainGeom<MM>::WN]) {
#pragma unroll
    for (int i = 0; i < ChainGeom<MM>::WM; i++)
#pragma unroll
        for (int j = 0; j < ChainGeom<MM>::WN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
}

// C = alpha*acc + diag*I + rcoef*R, written as bf16 (C16) and/or f32 (C32).  The tile goes through LDS as bf16 rows
// (f32: two row halves) so the global stores are 16-B row-contiguous.  Ends with the drain + barrier that makes the
// matrix readable by every wave of this workgroup.
template <int MM>
__device__ __forceinline__ void chain_store(f32x16 (&acc)[ChainGeom<MM>::WM][ChainGeom<MM>::WN], float alpha, float diag,
                                            const bf16_t* __restrict__ R, float rcoef, bf16_t* __restrict__ C16,
                                            float* __restrict__ C32, char* smem, int tid) {
    using CG = ChainGeom<MM>;
    constexpr int WM = CG::WM, WN = CG::WN;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, hh = lane >> 5;
    if (C16) {
        constexpr int P16 = MM + 8;
        bf16_t* t = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
        for (int j = 0; j < WN; j++) {
            const int col = wn * WN * 32 + j * 32 + r;
#pragma unroll
            for (int i = 0; i < WM; i++) {
                const int r0 = wm * WM * 32 + i * 32 + 4 * hh;
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    const int row = r0 + (reg & 3) + 8 * (reg >> 2);
                    float v = alpha * acc[i][j][reg];
                    if (row == col) v += diag;
                    if (R) v += rcoef * bf2f(R[row * MM + col]);
                    t[row * P16 + col] = f2bf(v);
                }
            }
        }
        __syncthreads();
        constexpr int CPR = MM / 8, NCH = MM * CPR / PC_NT;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int cid = tid + i * PC_NT;
            const int row = cid / CPR, c = cid % CPR;
            *reinterpret_cast<u32x4*>(C16 + row * MM + c * 8) = *reinterpret_cast<const u32x4*>(t + row * P16 + c * 8);
        }
        __syncthreads();
    }
    if (C32) {
        constexpr int P32 = MM + 4, HALF = MM / 2;
        float* t = reinterpret_cast<float*>(smem);
        static_assert(HALF * P32 * 4 <= CG::SMEM, "f32 half tile must fit");
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if ((wm >> 1) == half) {
#pragma unroll
                for (int j = 0; j < WN; j++) {
                    const int col = wn * WN * 32 + j * 32 + r;
#pragma unroll
                    for (int i = 0; i < WM; i++) {
                        const int r0 = (wm & 1) * WM * 32 + i * 32 + 4 * hh;   // row inside this half
#pragma unroll
                        for (int reg = 0; reg < 16; reg++) {
                            const int lr = r0 + (reg & 3) + 8 * (reg >> 2);
                            float v = alpha * acc[i][j][reg];
                            if (half * HALF + lr == col) v += diag;
                            t[lr * P32 + col] = v;
                        }
                    }
                }
            }
            __syncthreads();
            constexpr int CPR = MM / 4, NCH = HALF * CPR / PC_NT;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int cid = tid + i * PC_NT;
                const int lr = cid / CPR, c = cid % CPR;
                *reinterpret_cast<f32x4*>(C32 + (half * HALF + lr) * MM + c * 4) = *reinterpret_cast<const f32x4*>(t + lr * P32 + c * 4);
            }
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores have left this CU before any wave reads them back
    __syncthreads();
}

// forward: saved[k] = {z_k, P_k, T2_k, T3_k}, k < iters; z_0 must be in place; zf receives z_iters.
template <int MM>
__global__ __launch_bounds__(PC_NT) void pinv_chain_fwd_kernel(const bf16_t* __restrict__ X, bf16_t* __restrict__ saved,
                                                               bf16_t* __restrict__ zf, int BH, int iters) {
    using CG = ChainGeom<MM>;
    __shared__ __attribute__((aligned(16))) char smem[CG::SMEM];
    const int tid = threadIdx.x, bh = blockIdx.x;
    const long mat = (long)MM * MM;
    const bf16_t* Xb = X + bh * mat;
    f32x16 acc[CG::WM][CG::WN];
    for (int k = 0; k < iters; k++) {
        bf16_t* base = saved + ((long)k * 4 * BH + bh) * mat;   // [iters][4][BH][m][m]
        bf16_t* z = base;
        bf16_t* P = base + (long)BH * mat;
        bf16_t* T2 = base + 2L * BH * mat;
        bf16_t* T3 = base + 3L * BH * mat;
        bf16_t* zn = (k + 1 < iters) ? saved + ((long)(k + 1) * 4 * BH + bh) * mat : zf + bh * mat;
        chain_zero<MM>(acc);
        chain_mm<MM, true, false>(acc, Xb, z, smem, tid);
        chain_store<MM>(acc, 1.f, 0.f, nullptr, 0.f, P, nullptr, smem, tid);                 // P = X z
        chain_zero<MM>(acc);
        chain_mm<MM, true, false>(acc, P, P, smem, tid);
        chain_store<MM>(acc, 1.f, 15.f, P, -7.f, T2, nullptr, smem, tid);                    // T2 = 15I - 7P + P P
        chain_zero<MM>(acc);
        chain_mm<MM, true, false>(acc, P, T2, smem, tid);
        chain_store<MM>(acc, -1.f, 13.f, nullptr, 0.f, T3, nullptr, smem, tid);              // T3 = 13I - P T2
        chain_zero<MM>(acc);
        chain_mm<MM, true, false>(acc, z, T3, smem, tid);
        chain_store<MM>(acc, 0.25f, 0.f, nullptr, 0.f, zn, nullptr, smem, tid);              // z' = 1/4 z T3
    }
}

// backward: dzf = d z_iters (bf16); work[k] = {dT3, dT2, dP, dz_k}; dX (f32) = sum_k dP_k z_k^T; dz0 (f32) = d z_0
template <int MM>
__global__ __launch_bounds__(PC_NT) void pinv_chain_bwd_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ saved,
                                                               const bf16_t* __restrict__ dzf, bf16_t* __restrict__ work,
                                                               float* __restrict__ dX, float* __restrict__ dz0, int BH, int iters) {
    using CG = ChainGeom<MM>;
    __shared__ __attribute__((aligned(16))) char smem[CG::SMEM];
    const int tid = threadIdx.x, bh = blockIdx.x;
    const long mat = (long)MM * MM;
    const bf16_t* Xb = X + bh * mat;
    f32x16 acc[CG::WM][CG::WN];
    const bf16_t* dz = dzf + bh * mat;
    for (int k = iters - 1; k >= 0; k--) {
        const bf16_t* sb = saved + ((long)k * 4 * BH + bh) * mat;
        const bf16_t* z = sb;
        const bf16_t* P = sb + (long)BH * mat;
        const bf16_t* T2 = sb + 2L * BH * mat;
        const bf16_t* T3 = sb + 3L * BH * mat;
        bf16_t* wb = work + ((long)k * 4 * BH + bh) * mat;
        bf16_t* dT3 = wb;
        bf16_t* dT2 = wb + (long)BH * mat;
        bf16_t* dP = wb + 2L * BH * mat;
        bf16_t* dzk = wb + 3L * BH * mat;
        chain_zero<MM>(acc);
        chain_mm<MM, false, false>(acc, z, dz, smem, tid);
        chain_store<MM>(acc, 0.25f, 0.f, nullptr, 0.f, dT3, nullptr, smem, tid);             // dT3 = 1/4 z^T dz
        chain_zero<MM>(acc);
        chain_mm<MM, false, false>(acc, P, dT3, smem, tid);
        chain_store<MM>(acc, -1.f, 0.f, nullptr, 0.f, dT2, nullptr, smem, tid);              // dT2 = -P^T dT3
        // dP = -dT3 T2^T + dT2 P^T + P^T dT2 - 7 dT2   (the first product enters with a minus: negate afterwards)
        chain_zero<MM>(acc);
        chain_mm<MM, true, true>(acc, dT3, T2, smem, tid);
#pragma unroll
        for (int i = 0; i < CG::WM; i++)
#pragma unroll
            for (int j = 0; j < CG::WN; j++) acc[i][j] = -acc[i][j];
        chain_mm<MM, true, true>(acc, dT2, P, smem, tid);
        chain_mm<MM, false, false>(acc, P, dT2, smem, tid);
        chain_store<MM>(acc, 1.f, 0.f, dT2, -7.f, dP, nullptr, smem, tid);
        // dz_k = 1/4 dz T3^T + X^T dP   (scale the first product before the second joins)
        chain_zero<MM>(acc);
        chain_mm<MM, true, true>(acc, dz, T3, smem, tid);
#pragma unroll
        for (int i = 0; i < CG::WM; i++)
#pragma unroll
            for (int j = 0; j < CG::WN; j++) acc[i][j] = 0.25f * acc[i][j];
        chain_mm<MM, false, false>(acc, Xb, dP, smem, tid);
        chain_store<MM>(acc, 1.f, 0.f, nullptr, 0.f, dzk, k == 0 ? dz0 + bh * mat : nullptr, smem, tid);
        dz = dzk;
    }
    chain_zero<MM>(acc);
    for (int k = 0; k < iters; k++) {
        const bf16_t* z = saved + ((long)k * 4 * BH + bh) * mat;
        const bf16_t* dP = work + ((long)k * 4 * BH + bh) * mat + 2L * BH * mat;
        chain_mm<MM, true, true>(acc, dP, z, smem, tid);                                    // dX += dP_k z_k^T
    }
    chain_store<MM>(acc, 1.f, 0.f, nullptr, 0.f, nullptr, dX + bh * mat, smem, tid);
}

extern "C" int mh_pinv_chain_fwd(const void* X, void* saved, void* zf, int BH, int m, int iters, mh_stream s) {
    MH_REQUIRE(m == 128 || m == 256, "mh_pinv_chain_fwd: m=%d unsupported (128 or 256; other sizes use mh_gemm)", m);
    MH_REQUIRE(iters >= 1 && BH >= 0, "mh_pinv_chain_fwd: bad arguments");
    if (BH == 0) return MH_OK;
    if (m == 256) hipLaunchKernelGGL(pinv_chain_fwd_kernel<256>, dim3(BH), dim3(PC_NT), 0, (hipStream_t)s, (const bf16_t*)X, (bf16_t*)saved, (bf16_t*)zf, BH, iters);
    else hipLaunchKernelGGL(pinv_chain_fwd_kernel<128>, dim3(BH), dim3(PC_NT), 0, (hipStream_t)s, (const bf16_t*)X, (bf16_t*)saved, (bf16_t*)zf, BH, iters);
    MH_LAUNCH_CHECK("mh_pinv_chain_fwd");
    return MH_OK;
}

extern "C" int mh_pinv_chain_bwd(const void* X, const void* saved, const void* dzf, void* work, float* dX, float* dz0, int BH,
                                 int m, int iters, mh_stream s) {
    MH_REQUIRE(m == 128 || m == 256, "mh_pinv_chain_bwd: m=%d unsupported (128 or 256; other sizes use mh_gemm)", m);
    MH_REQUIRE(iters >= 1 && BH >= 0, "mh_pinv_chain_bwd: bad arguments");
    if (BH == 0) return MH_OK;
    if (m == 256) hipLaunchKernelGGL(pinv_chain_bwd_kernel<256>, dim3(BH), dim3(PC_NT), 0, (hipStream_t)s, (const bf16_t*)X, (const bf16_t*)saved, (const bf16_t*)dzf, (bf16_t*)work, dX, dz0, BH, iters);
    else hipLaunchKernelGGL(pinv_chain_bwd_kernel<128>, dim3(BH), dim3(PC_NT), 0, (hipStream_t)s, (const bf16_t*)X, (const bf16_t*)saved, (const bf16_t*)dzf, (bf16_t*)work, dX, dz0, BH, iters);
    MH_LAUNCH_CHECK("mh_pinv_chain_bwd");
    return MH_OK;
}
