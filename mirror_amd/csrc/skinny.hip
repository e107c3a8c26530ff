// Skinny-M linears of the RNA encoder / style heads (models/mirror.py:70-100, :217-224, :845-857): every tensor
// there is [B, D] with B = the per-GPU batch (16), so the GEMMs are weight-streaming problems (25 M parameters read
// once per pass) — the 128x128-tile kernel ran them on 8-64 workgroups at 2 TF/s.
//
//   mh_skinny_fwd   y[M<=32, N] = act(x[M,K] . W[N,K]^T + b)      weights go HBM -> VGPR in MFMA B-fragment layout
//                   (v_mfma_f32_16x16x32_bf16: lane l holds k = 8(l>>4)+j of column l&15 = one 16-B load of a W row),
//                   one wave per 16 output columns per K-slice, 4 K-slices per block reduced through LDS.
//                   The data gradient uses the same kernel on the transposed shadow W^T[K,N].
//   mh_skinny_wgrad dW[N,K] (+)= dy[M,N]^T . x[M,K]               rank-M outer products, 64x256 output tile per block,
//                   operands staged once in LDS, 8x8 outputs per thread, f32 read-modify-write (no atomics).
//   mh_transpose    bf16 [R,C] -> [C,R] (keeps the W^T shadows fresh)
#include "common.h"

typedef __bf16 sk_bf16x8 __attribute__((ext_vector_type(8)));
typedef float sk_f4 __attribute__((ext_vector_type(4)));
typedef unsigned sk_u4 __attribute__((ext_vector_type(4)));

// 8 consecutive k of an x row as the MFMA A fragment: bf16 as stored, f32 rounded to bf16 here (what a separate cast launch in
// front of every [B, D]-row Linear with an f32 input — residual streams, LayerNorm / style outputs — used to do)
template <typename TX> __device__ __forceinline__ sk_bf16x8 sk_load8(const TX* p);
template <> __device__ __forceinline__ sk_bf16x8 sk_load8<bf16_t>(const bf16_t* p) { return *reinterpret_cast<const sk_bf16x8*>(p); }
template <> __device__ __forceinline__ sk_bf16x8 sk_load8<float>(const float* p) {
    const sk_f4 a = *reinterpret_cast<const sk_f4*>(p), b = *reinterpret_cast<const sk_f4*>(p + 4);
    return sk_bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
}

// ANY (round 5): K and the row strides are arbitrary (the reference template's 10234 genes and its 1975-wide MLP, the 3000 prototypes as
// a contraction): a lane's 8 k values arrive as 8 guarded 2-byte (4-byte for f32 x) loads instead of one 16-byte load — the same bytes
// and the same coalescing across lanes, 8 x the load instructions, which a weight-streaming kernel of this size does not notice
// (the guarded 128 x 128-tile GEMM ran these on 1-4 workgroups: 1570 us for the 31 MB of the gene embedding).
template <typename TX> __device__ __forceinline__ sk_bf16x8 sk_load8_any(const TX* p, int valid);
template <> __device__ __forceinline__ sk_bf16x8 sk_load8_any<bf16_t>(const bf16_t* p, int valid) {
    sk_bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const bf16_t v = j < valid ? p[j] : (bf16_t)0;
        r[j] = __builtin_bit_cast(__bf16, v);
    }
    return r;
}
template <> __device__ __forceinline__ sk_bf16x8 sk_load8_any<float>(const float* p, int valid) {
    sk_bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = (__bf16)(j < valid ? p[j] : 0.f);
    return r;
}

template <typename TX, typename TY, int MT, bool ANY = false>   // MT = number of 16-row tiles of x (1 or 2)
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const TX* __restrict__ x, long ldx, const bf16_t* __restrict__ w,
                                                         long ldw, const float* __restrict__ bias, const float* __restrict__ addend,
                                                         long ldadd, TY* __restrict__ y, long ldy, int M, int N, int K, int act) {
    __shared__ float red[4][MT][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16;
    const int col = lane & 15, kq = lane >> 4;
    const int nrow = min(n0 + col, N - 1);                 // ragged last column group: re-read row N-1, never stored
    const bf16_t* wp = w + (long)nrow * ldw + 8 * kq;
    const TX* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) xp[t] = x + (long)min(16 * t + col, M - 1) * ldx + 8 * kq;
    sk_f4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = (sk_f4){0.f, 0.f, 0.f, 0.f};
    // wave `wave` owns k-steps wave, wave+4, ... (32 k per step): four independent 16-B streams per lane
    const int steps = ANY ? (K + 31) / 32 : K / 32;
#pragma unroll 4
    for (int s = wave; s < steps; s += 4) {
        if constexpr (ANY) {
            const int valid = K - (32 * s + 8 * kq);           // <= 0: this lane's 8 k are past the end
            const sk_bf16x8 b = sk_load8_any<bf16_t>(wp + 32 * s, valid);
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const sk_bf16x8 a = sk_load8_any<TX>(xp[t] + 32 * s, valid);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
            }
        } else {
            const sk_bf16x8 b = *reinterpret_cast<const sk_bf16x8*>(wp + 32 * s);
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const sk_bf16x8 a = sk_load8<TX>(xp[t] + 32 * s);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // C/D map of 16x16x32: col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave][t][kq * 4 + r][col] = acc[t][r];
    __syncthreads();
    for (int i = threadIdx.x; i < MT * 256; i += 256) {
        const int t = i >> 8, rr = (i >> 4) & 15, cc = i & 15;
        const int m = 16 * t + rr, n = n0 + cc;
        if (m < M && n < N) {
            float v = red[0][t][rr][cc] + red[1][t][rr][cc] + red[2][t][rr][cc] + red[3][t][rr][cc];
            if (bias) v += bias[n];
            if (addend) v += addend[(long)m * ldadd + n];      // a data gradient that sums what another consumer of x already produced
            if (act == MH_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == MH_ACT_GELU) v = gelu_f(v);
            stf(y + (long)m * ldy + n, v);
        }
    }
}

extern "C" int mh_skinny_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, const float* addend,
                             int64_t ldadd, void* y, int64_t ldy, int M, int N, int K, int act, int dt_x, int dt_y, mh_stream s) {
    MH_REQUIRE(M >= 1 && M <= 32 && K >= 1, "mh_skinny_fwd: M=%d (needs 1..32), K=%d", M, K);
    if (N == 0) return MH_OK;
    // 16-byte fragments need K % 32 == 0 and 16-byte aligned rows; anything else takes the element-wise instance (ANY)
    const bool vec = K % 32 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0;
    dim3 grid(mh_cdiv(N, 16));
#define SKF(TX, TY, MT, ANY) hipLaunchKernelGGL((skinny_fwd_kernel<TX, TY, MT, ANY>), grid, dim3(256), 0, (hipStream_t)s, (const TX*)x, (long)ldx, (const bf16_t*)w, (long)ldw, bias, addend, (long)ldadd, (TY*)y, (long)ldy, M, N, K, act)
#define SKF1(TX, TY, MT) do { if (vec) SKF(TX, TY, MT, false); else SKF(TX, TY, MT, true); } while (0)
#define SKF2(TX) do { if (dt_y == MH_F32) { if (M <= 16) SKF1(TX, float, 1); else SKF1(TX, float, 2); } else { if (M <= 16) SKF1(TX, bf16_t, 1); else SKF1(TX, bf16_t, 2); } } while (0)
    if (dt_x == MH_F32) SKF2(float); else SKF2(bf16_t);
#undef SKF2
#undef SKF1
#undef SKF
    MH_LAUNCH_CHECK("mh_skinny_fwd");
    return MH_OK;
}

// dW tile = 64 (n) x 256 (k) per block; thread (tn = tid>>5, tk = tid&31) owns n = 8*tn.., k = 8*tk..
#define SW_TN 64
#define SW_TK 256
// operand element as the bf16 MFMA path sees it: f32 sources are rounded to bf16 (no cast launch in front of the gradient)
__device__ __forceinline__ float sk_ld(const void* p, long i, int f32) {
    return f32 ? bf2f(f2bf(reinterpret_cast<const float*>(p)[i])) : bf2f(reinterpret_cast<const bf16_t*>(p)[i]);
}
__device__ __forceinline__ void skinny_wgrad_tile(const void* __restrict__ dy, long lddy, const void* __restrict__ x,
                                                  long ldx, float* __restrict__ dw, long lddw, float* __restrict__ db, int M,
                                                  int N, int K, int accumulate, int bx, int by, int dy_f32, int x_f32) {
    __shared__ __attribute__((aligned(16))) float sdy[32][SW_TN];
    __shared__ __attribute__((aligned(16))) float sx[32][SW_TK];
    const int n0 = bx * SW_TN, k0 = by * SW_TK;
    for (int i = threadIdx.x; i < M * SW_TN; i += 256) {
        const int m = i / SW_TN, c = i % SW_TN;
        sdy[m][c] = (n0 + c < N) ? sk_ld(dy, (long)m * lddy + n0 + c, dy_f32) : 0.f;
    }
    for (int i = threadIdx.x; i < M * SW_TK; i += 256) {
        const int m = i / SW_TK, c = i % SW_TK;
        sx[m][c] = (k0 + c < K) ? sk_ld(x, (long)m * ldx + k0 + c, x_f32) : 0.f;
    }
    __syncthreads();
    // bias gradient db[n] += sum_m dy[m][n] rides along (the k = 0 column of blocks owns it): one launch less per Linear
    if (db && by == 0 && threadIdx.x < SW_TN && n0 + threadIdx.x < N) {
        float t = 0.f;
        for (int m = 0; m < M; m++) t += sdy[m][threadIdx.x];
        db[n0 + threadIdx.x] += t;
    }
    const int tn = threadIdx.x >> 5, tk = threadIdx.x & 31;
    float acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = 0.f;
    for (int m = 0; m < M; m++) {
        const sk_f4 a0 = *reinterpret_cast<const sk_f4*>(&sdy[m][8 * tn]), a1 = *reinterpret_cast<const sk_f4*>(&sdy[m][8 * tn + 4]);
        const sk_f4 b0 = *reinterpret_cast<const sk_f4*>(&sx[m][8 * tk]), b1 = *reinterpret_cast<const sk_f4*>(&sx[m][8 * tk + 4]);
        const float a[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const float b[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] += a[i] * b[j];
    }
    const bool kvec = (k0 + 8 * tk + 8 <= K) && (lddw % 4 == 0);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int n = n0 + 8 * tn + i;
        if (n >= N) continue;
        float* dst = dw + (long)n * lddw + k0 + 8 * tk;
        if (kvec) {
            sk_f4 o0 = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]}, o1 = {acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
            if (accumulate) { o0 += *reinterpret_cast<const sk_f4*>(dst); o1 += *reinterpret_cast<const sk_f4*>(dst + 4); }
            *reinterpret_cast<sk_f4*>(dst) = o0;
            *reinterpret_cast<sk_f4*>(dst + 4) = o1;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (k0 + 8 * tk + j < K) dst[j] = (accumulate ? dst[j] : 0.f) + acc[i][j];
        }
    }
}

__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const void* __restrict__ dy, long lddy, const void* __restrict__ x,
                                                           long ldx, float* __restrict__ dw, long lddw, float* __restrict__ db, int M,
                                                           int N, int K, int accumulate, int dy_f32, int x_f32) {
    skinny_wgrad_tile(dy, lddy, x, ldx, dw, lddw, db, M, N, K, accumulate, blockIdx.x, blockIdx.y, dy_f32, x_f32);
}

// Many weight gradients in ONE launch: the [B, D]-row linears of the RNA branch and the heads (models/mirror.py:70-100, :217-224,
// :845-857) each had its own 8-32-workgroup launch (18 per step, ~22 us each on the side stream for 35 MB of f32 read-modify-write
// in all); their (dy, x) pairs are queued during the backward and the tiles of all of them form one grid.  The items travel in
// the kernel arguments (no device table: nothing to copy per step, capturable into a HIP graph as it is).
struct SkinnyMany {
    mh_skinny_wgrad_item it[MH_SKINNY_MANY_MAX];
    int tile0[MH_SKINNY_MANY_MAX + 1];      // first flat tile of each item
    int n;
};
__global__ __launch_bounds__(256) void skinny_wgrad_many_kernel(SkinnyMany a) {
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.tile0[i + 1]) i++;        // uniform scan over <= 32 entries
    const mh_skinny_wgrad_item& e = a.it[i];
    const int t = blockIdx.x - a.tile0[i], tn = (e.N + SW_TN - 1) / SW_TN;
    skinny_wgrad_tile(e.dy, (long)e.lddy, e.x, (long)e.ldx, e.dw, (long)e.lddw, e.db, e.M, e.N, e.K, 1, t % tn, t / tn, e.dt_dy == MH_F32, e.dt_x == MH_F32);
}

extern "C" int mh_skinny_wgrad_many(const mh_skinny_wgrad_item* items, int n, mh_stream s) {
    MH_REQUIRE(n >= 0 && n <= MH_SKINNY_MANY_MAX && (items || n == 0), "mh_skinny_wgrad_many: n=%d (at most %d items per call)", n, MH_SKINNY_MANY_MAX);
    if (n == 0) return MH_OK;
    SkinnyMany a;
    int tiles = 0;
    for (int i = 0; i < n; i++) {
        const mh_skinny_wgrad_item& e = items[i];
        MH_REQUIRE(e.dy && e.x && e.dw && e.M >= 1 && e.M <= 32 && e.N >= 1 && e.K >= 1 && ((uintptr_t)e.dw & 15) == 0,
                   "mh_skinny_wgrad_many: item %d: M=%d (needs 1..32), N=%d, K=%d, 16-byte aligned dW", i, e.M, e.N, e.K);
        a.it[i] = e;
        a.tile0[i] = tiles;
        tiles += mh_cdiv(e.N, SW_TN) * mh_cdiv(e.K, SW_TK);
    }
    a.tile0[n] = tiles;
    a.n = n;
    hipLaunchKernelGGL(skinny_wgrad_many_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)s, a);
    MH_LAUNCH_CHECK("mh_skinny_wgrad_many");
    return MH_OK;
}

extern "C" int mh_skinny_wgrad(const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, int64_t lddw, float* db, int M,
                               int N, int K, int accumulate, int dt_dy, int dt_x, mh_stream s) {
    MH_REQUIRE(M >= 1 && M <= 32, "mh_skinny_wgrad: M=%d (needs 1..32)", M);
    MH_REQUIRE(((uintptr_t)dw & 15) == 0, "mh_skinny_wgrad: dW must be 16-byte aligned");
    if (N == 0 || K == 0) return MH_OK;
    dim3 grid(mh_cdiv(N, SW_TN), mh_cdiv(K, SW_TK));
    hipLaunchKernelGGL(skinny_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)s, dy, (long)lddy, x,
                       (long)ldx, dw, (long)lddw, db, M, N, K, accumulate, (int)(dt_dy == MH_F32), (int)(dt_x == MH_F32));
    MH_LAUNCH_CHECK("mh_skinny_wgrad");
    return MH_OK;
}

// out[c][r] = in[r][c] (bf16), 64x64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R, int Cc) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? in[(long)r * Cc + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < Cc) out[(long)c * R + r] = tile[tx][i];
    }
}

extern "C" int mh_transpose_bf16(const void* in, void* out, int R, int Cc, mh_stream s) {
    if (R == 0 || Cc == 0) return MH_OK;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(mh_cdiv(Cc, 64), mh_cdiv(R, 64)), dim3(256), 0, (hipStream_t)s,
                       (const bf16_t*)in, (bf16_t*)out, R, Cc);
    MH_LAUNCH_CHECK("mh_transpose_bf16");
    return MH_OK;
}

// Batched form: table[i] = {src_off, dst_off, R, C} (element offsets into one bf16 arena each); one launch refreshes
// every W^T shadow after the optimizer step.
template <bool VEC>
__global__ __launch_bounds__(256) void transpose_bf16_many_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                                  const long* __restrict__ table) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][66];
    const long* e = table + 4 * blockIdx.z;
    const int R = (int)e[2], Cc = (int)e[3];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    if (r0 >= R || c0 >= Cc) return;
    const bf16_t* in = src + e[0];
    bf16_t* out = dst + e[1];
    if (VEC) {     // R % 8 == 0, Cc % 8 == 0, 16-byte aligned tensors: 16-byte global accesses both ways
        typedef uint32_t tq __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int cid = threadIdx.x + 256 * i, r = cid >> 3, ch = cid & 7;
            tq v = {0u, 0u, 0u, 0u};
            if (r0 + r < R && c0 + 8 * ch < Cc) v = *reinterpret_cast<const tq*>(in + (long)(r0 + r) * Cc + c0 + 8 * ch);
            uint32_t* t32 = reinterpret_cast<uint32_t*>(&tile[r][8 * ch]);     // pitch 132 B: 4-byte aligned only
            t32[0] = v[0]; t32[1] = v[1]; t32[2] = v[2]; t32[3] = v[3];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int cid = threadIdx.x + 256 * i, c = cid >> 3, ch = cid & 7;
            if (c0 + c < Cc && r0 + 8 * ch < R) {
                tq v;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    v[j] = (uint32_t)tile[8 * ch + 2 * j][c] | ((uint32_t)tile[8 * ch + 2 * j + 1][c] << 16);
                *reinterpret_cast<tq*>(out + (long)(c0 + c) * R + r0 + 8 * ch) = v;
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? in[(long)r * Cc + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < Cc) out[(long)c * R + r] = tile[tx][i];
    }
}

extern "C" int mh_transpose_bf16_many(const void* src, void* dst, const int64_t* table, int n, int max_r, int max_c, int vec_ok,
                                      mh_stream s) {
    if (n == 0) return MH_OK;
    MH_REQUIRE(n <= 65535, "mh_transpose_bf16_many: too many tensors");
    dim3 grid(mh_cdiv(max_c, 64), mh_cdiv(max_r, 64), n);
    // vec_ok: the caller vouches that every table entry has R % 8 == 0, C % 8 == 0 and offsets that are multiples of 8
    if (vec_ok && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0)
        hipLaunchKernelGGL((transpose_bf16_many_kernel<true>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)src, (bf16_t*)dst, (const long*)table);
    else
        hipLaunchKernelGGL((transpose_bf16_many_kernel<false>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)src, (bf16_t*)dst, (const long*)table);
    MH_LAUNCH_CHECK("mh_transpose_bf16_many");
    return MH_OK;
}
