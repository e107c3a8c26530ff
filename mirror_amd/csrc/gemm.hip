// mh_gemm(): argument checks, split-K / vectorisation decisions, dispatch to the three kernel families.
#include <cstdlib>
#include "gemm_kernel.h"

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool gemm_try_tile384(GemmArgs& a, int akc, int bkc, int dtC, int batch, void* c2, int r_bf16, hipStream_t s);   // gemm_tile.hip
const char* gemm_big_epi(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s);                     // gemm_big.hip
const char* gemm_big_window(GemmArgs& a, int akc, int bkc, int batch, hipStream_t s);                            // gemm_big.hip

static int launch_one(const mh_gemm_desc* d, hipStream_t s) {
    const int split = d->split_k < 1 ? 1 : d->split_k;
    GemmArgs a{};
    a.A = d->A; a.B = d->B; a.C = d->C; a.bias = d->bias;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.sA1 = d->sA1; a.sA2 = d->sA2; a.sB1 = d->sB1; a.sB2 = d->sB2; a.sC1 = d->sC1; a.sC2 = d->sC2;
    a.batch2 = d->batch2;
    a.alpha = d->alpha; a.diag = d->diag; a.act = d->act; a.accumulate = d->accumulate;
    a.R = d->R; a.rcoef = d->rcoef;
    a.ws = d->workspace; a.ws_floats = d->workspace_floats;
    a.kseg = d->k_segments; a.sAk = d->sA_seg; a.sBk = d->sB_seg;
    a.row_softmax = d->row_softmax;
    a.shared_chip = d->shared_chip;
    a.c_rpb = 0; a.c_skip = 0;
    a.w_last = d->window_batches > 0 ? d->window_batches - 1 : (1 << 30);
    const int BK = d->mma == MH_BF16 ? 64 : 16;
    const int kps = mh_cdiv(mh_cdiv(d->K, split), BK) * BK;
    a.k_per_split = kps;
    a.split_k = mh_cdiv(d->K, kps);  // every split has work
    const int va = d->dtA == MH_F32 ? 4 : 8, vb = d->dtB == MH_F32 ? 4 : 8;
    a.vecA = aligned16(d->A) && d->lda % va == 0 && d->sA1 % va == 0 && d->sA2 % va == 0;
    a.vecB = aligned16(d->B) && d->ldb % vb == 0 && d->sB1 % vb == 0 && d->sB2 % vb == 0;
    const int cvec = d->dtC == MH_F32 ? 4 : 8;
    a.vecC = aligned16(d->C) && d->ldc % cvec == 0 && d->sC1 % cvec == 0 && d->sC2 % cvec == 0;
    const int batch = d->batch1 * d->batch2;
    a.atomic = (a.split_k > 1) || (d->accumulate && batch > 1 && d->sC1 == 0 && d->sC2 == 0);
    MH_REQUIRE(!a.atomic || (d->dtC == MH_F32 && d->act == MH_ACT_NONE && d->accumulate),
               "mh_gemm: atomic accumulation (split-K / batch broadcast into C) needs f32 C, accumulate=1, no activation");
    if (d->epi && d->epi->kind != MH_EPI_NONE) {
        MH_REQUIRE(d->mma == MH_BF16 && d->dtA == MH_BF16 && d->dtB == MH_BF16 && !d->C2 && !d->r_bf16 && d->k_segments <= 1 && !d->row_softmax,
                   "mh_gemm: a fused epilogue needs plain bf16 operands");
        a.epi = *d->epi;
        a.a_rpb = d->a_rows_per_batch; a.a_skip = d->a_row_skip;
        const char* why = gemm_big_epi(a, d->a_kc, d->b_kc, d->dtC, batch, s);
        MH_REQUIRE(!why, "mh_gemm(epi %d): %s", d->epi->kind, why);
        MH_LAUNCH_CHECK("mh_gemm(epi)");
        return MH_OK;
    }
    if (d->a_rows_per_batch != 0 || d->c_rows_per_batch != 0) {      // plain product of row windows (no epilogue): the 256 x 256 direct-to-LDS kernels
        MH_REQUIRE(d->mma == MH_BF16 && d->dtA == MH_BF16 && d->dtB == MH_BF16 && d->dtC == MH_BF16 && !d->C2 && !d->r_bf16 && d->k_segments <= 1 &&
                       !d->row_softmax && !d->bias && d->act == MH_ACT_NONE,
                   "mh_gemm: row windows (a_rows_per_batch / c_rows_per_batch) need plain bf16 operands and a bf16 result, no bias / activation");
        a.a_rpb = d->a_rows_per_batch; a.a_skip = d->a_row_skip;
        a.c_rpb = d->c_rows_per_batch; a.c_skip = d->c_row_skip;
        const char* why = gemm_big_window(a, d->a_kc, d->b_kc, batch, s);
        MH_REQUIRE(!why, "mh_gemm(row windows): %s", why);
        MH_LAUNCH_CHECK("mh_gemm(windows)");
        return MH_OK;
    }
    const bool want_tile = d->C2 || d->r_bf16 || d->k_segments > 1 || d->row_softmax;
    if (d->mma == MH_BF16 && d->dtA == MH_BF16 && d->dtB == MH_BF16 && gemm_try_tile384(a, d->a_kc, d->b_kc, d->dtC, batch, d->C2, d->r_bf16, s)) {
        MH_LAUNCH_CHECK("mh_gemm(tile)");
        return MH_OK;
    }
    MH_REQUIRE(!want_tile, "mh_gemm: C2 / r_bf16 / k_segments / row_softmax need the 192 x 384 tile kernel (bf16 operands, M %% 192 == 0, N %% 384 == 0, K %% 64 == 0, no bias / split-K)");
    if (d->mma == MH_F32) gemm_launch_f32(a, d->a_kc, d->b_kc, batch, s);
    else if (d->dtA == MH_BF16 && d->dtB == MH_BF16) gemm_launch_bf16(a, d->a_kc, d->b_kc, d->dtC, batch, s);
    else if (d->dtA == MH_F32 && d->dtB == MH_F32) gemm_launch_mixed_ff(a, d->a_kc, d->b_kc, d->dtC, batch, s);
    else if (d->dtA == MH_F32) gemm_launch_mixed_fb(a, d->a_kc, d->b_kc, d->dtC, batch, s);
    else gemm_launch_mixed_bf(a, d->a_kc, d->b_kc, d->dtC, batch, s);
    MH_LAUNCH_CHECK("mh_gemm");
    return MH_OK;
}

// The K % 64 remainder of a weight gradient over B * 4097 rows is ONE row per slide: as a second tiled launch it cost 30 us
// (256 workgroups of f32 atomics for a rank-16 update).  C[m][n] += alpha * sum_z sum_{k < KT} A_z[k][m] B_z[k][n], both operands
// with the contraction index as their row (M- / N-contiguous rows), every batch reducing into the same C.  Runs after the
// main launch in stream order, so a plain read-modify-write is enough.
template <typename TA, typename TB>
__global__ __launch_bounds__(256) void rank_update_kernel(const TA* __restrict__ A, long lda, long sA, const TB* __restrict__ B, long ldb,
                                                          long sB, float* __restrict__ C, long ldc, int M, int N, int KT, int batch,
                                                          float alpha) {
    const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float acc = 0.f;
    // batch * KT (<= 16 * 16) independent products: eight loads of each operand in flight, not one dependent pair at a time
    const int total = batch * KT;
    int i = 0;
    for (; i + 8 <= total; i += 8) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int z = (i + u) / KT, k = (i + u) - z * KT;
            a[u] = ldf(A + z * sA + (long)k * lda + m);
            b[u] = ldf(B + z * sB + (long)k * ldb + n);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += a[u] * b[u];
    }
    for (; i < total; i++) {
        const int z = i / KT, k = i - z * KT;
        acc += ldf(A + z * sA + (long)k * lda + m) * ldf(B + z * sB + (long)k * ldb + n);
    }
    C[(long)m * ldc + n] += alpha * acc;
}

static bool rank_update_ok(const mh_gemm_desc* t) {
    const int batch = t->batch1 * t->batch2;
    if (t->K > 16 || t->a_kc || t->b_kc || t->dtC != MH_F32 || !t->accumulate || t->bias || t->R || t->diag != 0.f || t->act != MH_ACT_NONE)
        return false;
    if ((t->batch1 != 1 && t->batch2 != 1) || (batch > 1 && (t->sC1 != 0 || t->sC2 != 0)) || t->M > 65535 || batch * t->K > 256) return false;
    return true;
}
static bool try_rank_update(const mh_gemm_desc* t, hipStream_t s) {
    const int batch = t->batch1 * t->batch2;
    if (!rank_update_ok(t)) return false;
    const long sA = t->batch1 > 1 ? t->sA1 : t->sA2, sB = t->batch1 > 1 ? t->sB1 : t->sB2;     // the one batch axis in use
    dim3 grid(mh_cdiv(t->N, 256), t->M);
#define RU_(TA, TB) hipLaunchKernelGGL((rank_update_kernel<TA, TB>), grid, dim3(256), 0, s, (const TA*)t->A, (long)t->lda, sA, (const TB*)t->B, (long)t->ldb, sB, (float*)t->C, (long)t->ldc, t->M, t->N, t->K, batch, t->alpha)
    if (t->dtA == MH_BF16 && t->dtB == MH_BF16) RU_(bf16_t, bf16_t);
    else if (t->dtA == MH_F32 && t->dtB == MH_F32) RU_(float, float);
    else if (t->dtA == MH_F32) RU_(float, bf16_t);
    else RU_(bf16_t, float);
#undef RU_
    return true;
}

extern "C" int64_t mh_gemm_workspace_bytes(const mh_gemm_desc* d) {
    if (!d || d->mma != MH_BF16 || d->dtA != MH_BF16 || d->dtB != MH_BF16 || d->dtC != MH_F32 || !d->accumulate) return 0;
    if (d->M % 256 || d->N % 256 || d->M <= 0) return 0;
    const int split = d->split_k < 1 ? 1 : d->split_k;
    const bool bcast = (d->sC1 == 0 || d->batch1 == 1) && (d->sC2 == 0 || d->batch2 == 1);
    const int64_t parts = (int64_t)split * (bcast ? (int64_t)d->batch1 * d->batch2 : 1);
    if (parts < 8) return 0;
    return parts * d->M * d->N * 2;      // bf16 partial tiles
}

extern "C" int mh_gemm(const mh_gemm_desc* d, mh_stream stream) {
    MH_REQUIRE(d && d->A && d->B && d->C, "mh_gemm: null pointer");
    MH_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "mh_gemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    MH_REQUIRE(d->batch1 >= 1 && d->batch2 >= 1 && (long)d->batch1 * d->batch2 <= 65535, "mh_gemm: bad batch");
    MH_REQUIRE(d->mma == MH_BF16 || (d->dtA == MH_F32 && d->dtB == MH_F32 && d->dtC == MH_F32), "mh_gemm: f32 MMA needs f32 operands");
    const int split = d->split_k < 1 ? 1 : d->split_k;
    MH_REQUIRE(split == 1 || (d->accumulate && d->dtC == MH_F32 && d->act == MH_ACT_NONE),
               "mh_gemm: split_k>1 needs accumulate=1, f32 C, no activation");
    MH_REQUIRE(split <= 65535, "mh_gemm: split_k too large");
    MH_REQUIRE(!d->R || split == 1, "mh_gemm: the R addend cannot be combined with split-K");
    MH_REQUIRE(d->act == MH_ACT_NONE || d->act == MH_ACT_RELU, "mh_gemm: only ReLU is fused (GELU runs as mh_gelu_fwd)");
    hipStream_t s = (hipStream_t)stream;
    // A ragged K (e.g. a weight gradient over B*(N+1) rows) would push the whole launch onto the guarded kernel:
    // run the BK-multiple part on the fast path and add the short tail with a second (accumulating) launch.
    const int BK = d->mma == MH_BF16 ? 64 : 16;
    const int tail = d->K % BK;
    if (tail != 0 && d->K >= 8 * BK && d->dtC == MH_F32 && d->act == MH_ACT_NONE) {
        mh_gemm_desc m = *d, t = *d;
        const long ea = d->dtA == MH_F32 ? 4 : 2, eb = d->dtB == MH_F32 ? 4 : 2;
        const int kmain = d->K - tail;
        m.K = kmain;
        t.K = tail;
        t.A = (const char*)d->A + ea * (d->a_kc ? (long)kmain : (long)kmain * d->lda);
        t.B = (const char*)d->B + eb * (d->b_kc ? (long)kmain : (long)kmain * d->ldb);
        t.bias = nullptr;
        t.diag = 0.f;
        t.R = nullptr;
        t.accumulate = 1;
        t.split_k = 1;
        // offer the remainder to the main launch's fold pass (gemm_big.hip: partial tiles + fold): one launch less when it takes it
        GemmTail* pt = gemm_pending_tail();
        pt->KT = 0;
        constexpr bool merge = true;
        if (merge && m.accumulate && rank_update_ok(&t)) {
            const int batch_t = t.batch1 * t.batch2;
            *pt = GemmTail{t.A, t.B, (long)t.lda, t.batch1 > 1 ? (long)t.sA1 : (long)t.sA2, (long)t.ldb, t.batch1 > 1 ? (long)t.sB1 : (long)t.sB2,
                           t.K, batch_t, t.dtA == MH_F32, t.dtB == MH_F32, t.alpha};
        }
        const int rc = launch_one(&m, s);
        const bool taken = merge && pt->KT == 0 && m.accumulate && rank_update_ok(&t);
        pt->KT = 0;
        if (rc != MH_OK) return rc;
        if (taken) {
            MH_LAUNCH_CHECK("mh_gemm(tail in fold)");
            return MH_OK;
        }
        if (m.accumulate && try_rank_update(&t, s)) {
            MH_LAUNCH_CHECK("mh_gemm(tail)");
            return MH_OK;
        }
        return launch_one(&t, s);
    }
    return launch_one(d, s);
}
