// mh_gemm(): argument checks, split-K / vectorisation decisions, dispatch to the three kernel families.
#include "gemm_kernel.h"

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int launch_one(const mh_gemm_desc* d, hipStream_t s) {
    const int split = d->split_k < 1 ? 1 : d->split_k;
    GemmArgs a;
    a.A = d->A; a.B = d->B; a.C = d->C; a.bias = d->bias;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.sA1 = d->sA1; a.sA2 = d->sA2; a.sB1 = d->sB1; a.sB2 = d->sB2; a.sC1 = d->sC1; a.sC2 = d->sC2;
    a.batch2 = d->batch2;
    a.alpha = d->alpha; a.diag = d->diag; a.act = d->act; a.accumulate = d->accumulate;
    a.R = d->R; a.rcoef = d->rcoef;
    a.ws = d->workspace; a.ws_floats = d->workspace_floats;
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
    if (d->mma == MH_F32) gemm_launch_f32(a, d->a_kc, d->b_kc, batch, s);
    else if (d->dtA == MH_BF16 && d->dtB == MH_BF16) gemm_launch_bf16(a, d->a_kc, d->b_kc, d->dtC, batch, s);
    else if (d->dtA == MH_F32 && d->dtB == MH_F32) gemm_launch_mixed_ff(a, d->a_kc, d->b_kc, d->dtC, batch, s);
    else if (d->dtA == MH_F32) gemm_launch_mixed_fb(a, d->a_kc, d->b_kc, d->dtC, batch, s);
    else gemm_launch_mixed_bf(a, d->a_kc, d->b_kc, d->dtC, batch, s);
    MH_LAUNCH_CHECK("mh_gemm");
    return MH_OK;
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
        const int rc = launch_one(&m, s);
        if (rc != MH_OK) return rc;
        return launch_one(&t, s);
    }
    return launch_one(d, s);
}
