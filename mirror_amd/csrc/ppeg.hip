// PPEG (models/mirror.py:317-331): proj7(x) + x + proj5(x) + proj3(x) on the sqrt(N) x sqrt(N) token grid.
// The three depthwise convs and the identity are merged on the fly into ONE 7x7 depthwise filter
// (5x5 / 3x3 zero-embedded at the centre, +1 at the centre tap, biases summed): exact, and the tokens
// stay token-major [B, 1+S*S, D] so channel reads are coalesced (no NCHW round trip).
#include "common.h"

__global__ __launch_bounds__(256) void ppeg_merge_kernel(const float* __restrict__ w7, const float* __restrict__ w5,
                                                         const float* __restrict__ w3, const float* __restrict__ b7,
                                                         const float* __restrict__ b5, const float* __restrict__ b3,
                                                         float* __restrict__ merged, float* __restrict__ bsum, int D) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < 49 * D) {
        const int tap = idx / D, c = idx % D;
        const int ky = tap / 7, kx = tap % 7;
        float v = w7[c * 49 + tap];
        if (ky >= 1 && ky <= 5 && kx >= 1 && kx <= 5) v += w5[c * 25 + (ky - 1) * 5 + (kx - 1)];
        if (ky >= 2 && ky <= 4 && kx >= 2 && kx <= 4) v += w3[c * 9 + (ky - 2) * 3 + (kx - 2)];
        if (ky == 3 && kx == 3) v += 1.f;
        merged[idx] = v;
    }
    if (idx < D) bsum[idx] = b7[idx] + b5[idx] + b3[idx];
}

extern "C" int mh_ppeg_merge(const float* w7, const float* w5, const float* w3, const float* b7, const float* b5,
                             const float* b3, float* merged, float* bsum, int D, mh_stream s) {
    hipLaunchKernelGGL(ppeg_merge_kernel, dim3(mh_cdiv(49 * D, 256)), dim3(256), 0, (hipStream_t)s, w7, w5, w3, b7, b5, b3, merged, bsum, D);
    MH_LAUNCH_CHECK("mh_ppeg_merge");
    return MH_OK;
}

// y[b, 1 + yy*S + xx, c] = bsum[c] + sum_tap merged[tap'][c] * x[b, 1 + (yy+ky-3)*S + (xx+kx-3), c]
// flip: tap' = 48 - tap and no bias (adjoint, used for the data gradient). cls row (token 0) is copied.
// Block: 256 channels x PX_T pixels along x for one (b, yy); weights for the thread's channel live in registers.
#define PX_T 8
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void ppeg_kernel(const TX* __restrict__ x, TY* __restrict__ y, const float* __restrict__ merged,
                                                   const float* __restrict__ bsum, int S, int D, int flip) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    const int tiles_x = (S + PX_T - 1) / PX_T;
    const int yy = blockIdx.y / tiles_x, x0 = (blockIdx.y % tiles_x) * PX_T;
    const long b = blockIdx.z;
    const long n = 1 + (long)S * S;
    const TX* xb = x + b * n * D + c;
    TY* yb = y + b * n * D + c;
    if (blockIdx.y == 0) stf(yb, ldf(xb));  // cls token passes through
    float w[49];
#pragma unroll
    for (int t = 0; t < 49; t++) w[t] = merged[(flip ? 48 - t : t) * D + c];
    float acc[PX_T];
    const float bias = flip ? 0.f : bsum[c];
#pragma unroll
    for (int i = 0; i < PX_T; i++) acc[i] = bias;
#pragma unroll
    for (int ky = 0; ky < 7; ky++) {
        const int sy = yy + ky - 3;
        if (sy < 0 || sy >= S) continue;
        const TX* row = xb + (1 + (long)sy * S) * D;
#pragma unroll
        for (int u = 0; u < PX_T + 6; u++) {
            const int sx = x0 - 3 + u;
            const float v = (sx >= 0 && sx < S) ? ldf(row + (long)sx * D) : 0.f;
#pragma unroll
            for (int i = 0; i < PX_T; i++) {
                const int kx = u - i;
                if (kx >= 0 && kx < 7) acc[i] += w[ky * 7 + kx] * v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PX_T; i++) {
        const int xx = x0 + i;
        if (xx < S) stf(yb + (1 + (long)yy * S + xx) * D, acc[i]);
    }
}

// Column-strip variant: a thread owns one channel and PX_T pixels along x and walks DOWN the grid with a 7-row register
// window (row r lives in slot r mod 7; the walk is unrolled by 7 so every slot index is a compile-time constant).  Each
// input row is loaded once per strip: (PX_T + 6) / PX_T = 1.75 loads per output instead of 12.25.
#define PY_T 16   // grid rows per block
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void ppeg_strip_kernel(const TX* __restrict__ x, TY* __restrict__ y, const float* __restrict__ merged,
                                                         const float* __restrict__ bsum, int S, int D, int flip) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    const int tiles_x = (S + PX_T - 1) / PX_T;
    const int y0 = (blockIdx.y / tiles_x) * PY_T, x0 = (blockIdx.y % tiles_x) * PX_T;
    const long b = blockIdx.z;
    const long n = 1 + (long)S * S;
    const TX* xb = x + b * n * D + c;
    TY* yb = y + b * n * D + c;
    if (blockIdx.y == 0) stf(yb, ldf(xb));  // cls token passes through
    float w[49];
#pragma unroll
    for (int t = 0; t < 49; t++) w[t] = merged[(flip ? 48 - t : t) * D + c];
    const float bias = flip ? 0.f : bsum[c];
    float win[7][PX_T + 6];
    auto load_row = [&](float (&dst)[PX_T + 6], int sy) {
        const bool rok = sy >= 0 && sy < S;
        const TX* row = xb + (1 + (long)(rok ? sy : 0) * S) * D;
#pragma unroll
        for (int u = 0; u < PX_T + 6; u++) {
            const int sx = x0 - 3 + u;
            dst[u] = (rok && sx >= 0 && sx < S) ? ldf(row + (long)sx * D) : 0.f;
        }
    };
    // rows y0-3 .. y0+2 -> slots 4, 5, 6, 0, 1, 2
    load_row(win[4], y0 - 3); load_row(win[5], y0 - 2); load_row(win[6], y0 - 1);
    load_row(win[0], y0); load_row(win[1], y0 + 1); load_row(win[2], y0 + 2);
    for (int yb0 = y0; yb0 < y0 + PY_T && yb0 < S; yb0 += 7) {
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int yy = yb0 + j;
            if (yy >= y0 + PY_T || yy >= S) break;
            load_row(win[(j + 3) % 7], yy + 3);
            float acc[PX_T];
#pragma unroll
            for (int i = 0; i < PX_T; i++) acc[i] = bias;
#pragma unroll
            for (int ky = 0; ky < 7; ky++)
#pragma unroll
                for (int kx = 0; kx < 7; kx++) {
                    const float wv = w[ky * 7 + kx];
#pragma unroll
                    for (int i = 0; i < PX_T; i++) acc[i] += wv * win[(j + ky + 4) % 7][i + kx];
                }
#pragma unroll
            for (int i = 0; i < PX_T; i++)
                if (x0 + i < S) stf(yb + (1 + (long)yy * S + x0 + i) * D, acc[i]);
        }
    }
}

// Channel-pair form of the strip kernel (f32 in / out, D even): a thread owns TWO adjacent channels, so window, weights and
// accumulators are float2 and every multiply-add is a v_pk_fma_f32 — the depthwise 7 x 7 filter is 49 FMAs per output
// element (1.6 G per launch at c2: 42 us of plain v_fma issue, 21 us packed) and nothing else in the kernel is close.
#define P2_T 4    // pixels along x per thread (the 7 x (P2_T + 6) float2 window is 140 registers)
typedef float pp2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void ppeg_strip2_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ merged,
                                                          const float* __restrict__ bsum, int S, int D, int flip) {
    const int c = 2 * (blockIdx.x * 256 + threadIdx.x);
    if (c >= D) return;
    const int tiles_x = (S + P2_T - 1) / P2_T;
    const int y0 = (blockIdx.y / tiles_x) * PY_T, x0 = (blockIdx.y % tiles_x) * P2_T;
    const long b = blockIdx.z;
    const long n = 1 + (long)S * S;
    const float* xb = x + b * n * D + c;
    float* yb = y + b * n * D + c;
    if (blockIdx.y == 0) *reinterpret_cast<pp2*>(yb) = *reinterpret_cast<const pp2*>(xb);  // cls token passes through
    pp2 w[49];
#pragma unroll
    for (int t = 0; t < 49; t++) w[t] = *reinterpret_cast<const pp2*>(merged + (flip ? 48 - t : t) * D + c);
    const pp2 bias = flip ? (pp2){0.f, 0.f} : *reinterpret_cast<const pp2*>(bsum + c);
    pp2 win[7][P2_T + 6];
    auto load_row = [&](pp2 (&dst)[P2_T + 6], int sy) {
        const bool rok = sy >= 0 && sy < S;
        const float* row = xb + (1 + (long)(rok ? sy : 0) * S) * D;
#pragma unroll
        for (int u = 0; u < P2_T + 6; u++) {
            const int sx = x0 - 3 + u;
            dst[u] = (rok && sx >= 0 && sx < S) ? *reinterpret_cast<const pp2*>(row + (long)sx * D) : (pp2){0.f, 0.f};
        }
    };
    load_row(win[4], y0 - 3); load_row(win[5], y0 - 2); load_row(win[6], y0 - 1);
    load_row(win[0], y0); load_row(win[1], y0 + 1); load_row(win[2], y0 + 2);
    for (int yb0 = y0; yb0 < y0 + PY_T && yb0 < S; yb0 += 7) {
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int yy = yb0 + j;
            if (yy >= y0 + PY_T || yy >= S) break;
            load_row(win[(j + 3) % 7], yy + 3);
            pp2 acc[P2_T];
#pragma unroll
            for (int i = 0; i < P2_T; i++) acc[i] = bias;
#pragma unroll
            for (int ky = 0; ky < 7; ky++)
#pragma unroll
                for (int kx = 0; kx < 7; kx++) {
                    const pp2 wv = w[ky * 7 + kx];
#pragma unroll
                    for (int i = 0; i < P2_T; i++) acc[i] = __builtin_elementwise_fma(wv, win[(j + ky + 4) % 7][i + kx], acc[i]);
                }
#pragma unroll
            for (int i = 0; i < P2_T; i++)
                if (x0 + i < S) *reinterpret_cast<pp2*>(yb + (1 + (long)yy * S + x0 + i) * D) = acc[i];
        }
    }
}

// Input-stationary form of the channel-pair kernel: instead of a 7-row input window (140 registers, the newest row loaded
// and consumed in the same row step) a thread keeps SEVEN partial output rows (56 registers) and two input rows: input row
// sy adds its seven filter rows into output rows sy - 3 .. sy + 3, output row sy - 3 is then complete and stored, and row
// sy + 1 was requested a whole row step earlier.  Rows outside the strip are skipped by uniform branches.
#ifndef P2_FORM
#define P2_FORM 1     // 1: input-stationary rows (ppeg_rows2_kernel), 0: 7-row input window (ppeg_strip2_kernel)
#endif
#ifndef PR_T
#define PR_T 32       // grid rows per strip of ppeg_rows2_kernel
#endif
__global__ __launch_bounds__(256) void ppeg_rows2_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ merged,
                                                         const float* __restrict__ bsum, int S, int D, int flip, int nb, int pr) {
    // pr = grid rows per strip, chosen by the host so that the launch is whole rounds of resident workgroups (mh_ppeg_fwd)
    // 1-D grid, XCD-aware: workgroup L runs on XCD L % 8 and neighbours in time are L +- 8.  Six of the ten columns a thread loads per
    // row belong to the x-tiles next to its own, so the x-tiles of one (batch, strip) are walked by ONE XCD back to back (their halo
    // columns then come out of that XCD's L2; with x-tiles on consecutive workgroup ids every halo column was fetched from the fabric
    // by two or three XCDs: 73.2 -> 65.5 us alone on the chip at B = 16, S = 64, D = 512; 2 x 64-row strips or 2-pixel tiles: 91 / 68 us).  groups = (batch, strip) pairs, a multiple of 8 or the plain order.
    const int tiles_x = (S + P2_T - 1) / P2_T, strips = (S + pr - 1) / pr;
    const int per_cb = tiles_x * strips * nb;
    const int cb = blockIdx.x / per_cb, L = blockIdx.x % per_cb;
    const int c = 2 * (cb * 256 + threadIdx.x);
    if (c >= D) return;
    int xt, grp;
    if ((strips * nb) % 8 == 0) { const int q = L >> 3; xt = q % tiles_x; grp = (q / tiles_x) * 8 + (L & 7); }
    else { xt = L % tiles_x; grp = L / tiles_x; }
    const int y0 = (grp % strips) * pr, x0 = xt * P2_T;
    const int rows = min(pr, S - y0);
    const long b = grp / strips;
    const bool first = (y0 == 0 && x0 == 0);
    const long n = 1 + (long)S * S;
    const float* xb = x + b * n * D + c;
    float* yb = y + b * n * D + c;
    if (first) *reinterpret_cast<pp2*>(yb) = *reinterpret_cast<const pp2*>(xb);  // cls token passes through
    pp2 w[49];
#pragma unroll
    for (int t = 0; t < 49; t++) w[t] = *reinterpret_cast<const pp2*>(merged + (flip ? 48 - t : t) * D + c);
    const pp2 bias = flip ? (pp2){0.f, 0.f} : *reinterpret_cast<const pp2*>(bsum + c);
    pp2 in[2][P2_T + 6];
    auto load_row = [&](pp2 (&dst)[P2_T + 6], int sy) {
        const bool rok = sy >= 0 && sy < S;
        const float* row = xb + (1 + (long)(rok ? sy : 0) * S) * D;
#pragma unroll
        for (int u = 0; u < P2_T + 6; u++) {
            const int sx = x0 - 3 + u;
            dst[u] = (rok && sx >= 0 && sx < S) ? *reinterpret_cast<const pp2*>(row + (long)sx * D) : (pp2){0.f, 0.f};
        }
    };
    pp2 acc[7][P2_T];          // output row y0 + o lives in slot o % 7
#pragma unroll
    for (int i = 0; i < P2_T; i++) acc[0][i] = bias;
    const int steps = rows + 6;   // input rows y0 - 3 .. y0 + rows + 2
    load_row(in[0], y0 - 3);
    for (int kb = 0; kb < steps; kb += 14) {
#pragma unroll
        for (int j = 0; j < 14; j++) {
            const int k = kb + j;
            if (k >= steps) break;
            const int sy = y0 - 3 + k;
            if (k + 1 < steps) load_row(in[(j + 1) & 1], sy + 1);
            if (sy >= 0 && sy < S) {      // a row of zero padding adds nothing
#pragma unroll
                for (int ky = 0; ky < 7; ky++) {
                    const int o = k - ky;
                    if (o >= 0 && o < rows) {
#pragma unroll
                        for (int kx = 0; kx < 7; kx++) {
                            const pp2 wv = w[ky * 7 + kx];
#pragma unroll
                            for (int i = 0; i < P2_T; i++)
                                acc[(j - ky + 14) % 7][i] = __builtin_elementwise_fma(wv, in[j & 1][i + kx], acc[(j - ky + 14) % 7][i]);
                        }
                    }
                }
            }
            const int od = k - 6;         // this output row has seen all seven input rows
            if (od >= 0) {
#pragma unroll
                for (int i = 0; i < P2_T; i++)
                    if (x0 + i < S) *reinterpret_cast<pp2*>(yb + (1 + (long)(y0 + od) * S + x0 + i) * D) = acc[(j + 1) % 7][i];
            }
#pragma unroll
            for (int i = 0; i < P2_T; i++) acc[(j + 1) % 7][i] = bias;      // slot of output row k + 1
        }
    }
}

extern "C" int mh_ppeg_fwd(const void* x, void* y, const float* merged, const float* bsum, int B, int S, int D, int flip,
                           int dt_x, int dt_y, mh_stream s) {
    MH_REQUIRE(S >= 1 && D >= 1, "mh_ppeg_fwd: bad shape S=%d D=%d", S, D);
    if (B == 0) return MH_OK;
    if (dt_x == MH_F32 && dt_y == MH_F32 && D % 2 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)merged | (uintptr_t)bsum) & 7) == 0) {
#if P2_FORM
        // rows per strip: two workgroups of ~200 VGPRs fit a CU, so the launch runs in rounds of 512; fewest (rounds x row steps of a
        // strip, 6 of them halo), PR_T rows when that is as good (c2: 2 strips of 32 rows = 512 workgroups; config 4's 91 x 91 grid at B = 8:
        // 3 strips were 552 workgroups = a second round for 40 of them)
        int pr = PR_T;
        {
            const long per = (long)mh_cdiv(D / 2, 256) * mh_cdiv(S, P2_T) * B;
            long best = -1;
            for (int st = 1; st <= mh_cdiv(S, 8); st++) {
                const int rows = mh_cdiv(S, st);
                if (rows > 64) continue;
                const long cost = mh_cdiv(per * mh_cdiv(S, rows), 512) * (rows + 6);
                if (best < 0 || cost < best || (cost == best && rows == PR_T)) { best = cost; pr = rows; }
            }
#ifdef MH_EXP
            if (const char* e = getenv("MH_PPEG_PR")) pr = atoi(e);      // tools/exp/ppeg_rows_per_strip.sh
#endif
        }
        dim3 g2(mh_cdiv(D / 2, 256) * mh_cdiv(S, pr) * mh_cdiv(S, P2_T) * B);
        hipLaunchKernelGGL(ppeg_rows2_kernel, g2, dim3(256), 0, (hipStream_t)s, (const float*)x, (float*)y, merged, bsum, S, D, flip, B, pr);
#else
        dim3 g2(mh_cdiv(D / 2, 256), mh_cdiv(S, PY_T) * mh_cdiv(S, P2_T), B);
        hipLaunchKernelGGL(ppeg_strip2_kernel, g2, dim3(256), 0, (hipStream_t)s, (const float*)x, (float*)y, merged, bsum, S, D, flip);
#endif
        MH_LAUNCH_CHECK("mh_ppeg_fwd");
        return MH_OK;
    }
    // PY_T = 16 rows per strip is not a multiple of the unroll (7): the walk is 7 + 7 + 2 rows with the window slots
    // following the row index, so a strip must start at a slot-0 row -> strips are re-based every PY_T rows
    dim3 grid(mh_cdiv(D, 256), mh_cdiv(S, PY_T) * mh_cdiv(S, PX_T), B);
#define PP(TX, TY) hipLaunchKernelGGL((ppeg_strip_kernel<TX, TY>), grid, dim3(256), 0, (hipStream_t)s, (const TX*)x, (TY*)y, merged, bsum, S, D, flip)
    if (dt_x == MH_F32 && dt_y == MH_F32) PP(float, float);
    else if (dt_x == MH_BF16 && dt_y == MH_BF16) PP(bf16_t, bf16_t);
    else if (dt_x == MH_F32 && dt_y == MH_BF16) PP(float, bf16_t);
    else PP(bf16_t, float);
#undef PP
    MH_LAUNCH_CHECK("mh_ppeg_fwd");
    return MH_OK;
}

// dmerged[tap][c] += sum_{b,yy,xx} dout[b,yy,xx,c] * x[b,yy+ky-3,xx+kx-3,c]; dbsum[c] += sum dout.
// Block = 64 channels x 4 grid rows (one wave per row, lanes = channels: 256-B coalesced loads).  Each thread
// sweeps its row with a 7x7 register window of x (column X lives in slot X mod 7; the sweep is unrolled by 7 so
// every slot index is a compile-time constant): 7 loads + 49 FMAs per pixel.  The 4 rows are reduced through
// LDS, then one f32 atomic per (tap, channel) per block.
#define WG_ROWS 4
template <typename TX, typename TO>
__global__ __launch_bounds__(256) void ppeg_wgrad_kernel(const TX* __restrict__ x, const TO* __restrict__ dout,
                                                         float* __restrict__ dmerged, float* __restrict__ dbsum, int S, int D) {
    __shared__ float red[4][50][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int yy = blockIdx.y * WG_ROWS + slice;
    const long b = blockIdx.z;
    const long n = 1 + (long)S * S;
    const bool live = (c < D) && (yy < S);
    const TX* xb = x + b * n * D + (live ? c : 0);
    const TO* db = dout + b * n * D + (live ? c : 0);
    float acc[49], win[7][7];
#pragma unroll
    for (int t = 0; t < 49; t++) acc[t] = 0.f;
    float bacc = 0.f;
    if (live) {
        bool rowok[7];
#pragma unroll
        for (int ky = 0; ky < 7; ky++) {
            const int sy = yy + ky - 3;
            rowok[ky] = (sy >= 0 && sy < S);
#pragma unroll
            for (int sl = 0; sl < 7; sl++)
                win[ky][sl] = (sl < 3 && sl < S && rowok[ky]) ? ldf(xb + (1 + (long)sy * S + sl) * D) : 0.f;
        }
        for (int xx0 = 0; xx0 < S; xx0 += 7) {
#pragma unroll
            for (int u = 0; u < 7; u++) {
                const int xx = xx0 + u;
                const int xn = xx + 3;  // column entering the window
#pragma unroll
                for (int ky = 0; ky < 7; ky++)
                    win[ky][(u + 3) % 7] = (xn < S && rowok[ky]) ? ldf(xb + (1 + (long)(yy + ky - 3) * S + xn) * D) : 0.f;
                const float g = (xx < S) ? ldf(db + (1 + (long)yy * S + xx) * D) : 0.f;
                bacc += g;
#pragma unroll
                for (int ky = 0; ky < 7; ky++)
#pragma unroll
                    for (int kx = 0; kx < 7; kx++) acc[ky * 7 + kx] += g * win[ky][(u + kx + 4) % 7];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 49; t++) red[slice][t][lane] = acc[t];
    red[slice][49][lane] = bacc;
    __syncthreads();
    if (c < D) {
        for (int t = slice; t < 50; t += 4) {
            const float v = red[0][t][lane] + red[1][t][lane] + red[2][t][lane] + red[3][t][lane];
            if (t < 49) atomicAdd(dmerged + t * D + c, v);
            else atomicAdd(dbsum + c, v);
        }
    }
}

// Column-strip weight gradient: a thread owns one channel and PX_T pixels along x, walks down PW_ROWS grid rows with a
// register window of input rows: 8 + 14 loads feed 49 x 8 FMAs (the row-sweep kernel needs 64).
// The window is a ring of EIGHT rows and the gradient row is double-buffered: row yy + 4 and the gradients of row yy + 1
// are requested while row yy is accumulated, so no FMA waits for a load issued in its own iteration (with the 7-row
// window every row step began with a full memory latency, at two waves per SIMD: 173 us for 268 MB).
// (A channel-pair v_pk_fma form like ppeg_strip2_kernel was slower here: 98 accumulator + 140 window registers leave one
// wave per SIMD and nothing to hide the row loads behind.)
#ifndef PW_ROWS
#define PW_ROWS 64   // multiple of 8 (the ring is unrolled by its length)
#endif
#ifndef PW_X
#define PW_X 4       // pixels along x per thread
#endif
#ifndef PW_RED
#define PW_RED 1     // 1: workgroup = 64 channels x 4 strips with an LDS reduction, 0: 256 channels x 1 strip
#endif
template <typename TX, typename TO>
__global__ __launch_bounds__(256) void ppeg_wgrad_strip_kernel(const TX* __restrict__ x, const TO* __restrict__ dout,
                                                               float* __restrict__ dmerged, float* __restrict__ dbsum, int S, int D, int pr) {
    // pr = grid rows per strip (<= PW_ROWS): the host cuts the S rows into EQUAL strips (S = 91: 46 + 45 rows, not 64 + 27 — the long
    // strips set the launch's time)
    const int tiles_x = (S + PW_X - 1) / PW_X;
#if PW_RED
    // a workgroup = 64 channels x 4 strips (one per wave); the four partial filters are summed through LDS, so a
    // workgroup issues 50 x 64 atomics instead of 50 x 256
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: addresses stay scalar
    const int c = blockIdx.x * 64 + lane;
    const int strip = blockIdx.y * 4 + wave;
    const bool live = c < D && strip < tiles_x * ((S + pr - 1) / pr);
#else
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    const int strip = blockIdx.y;
    const bool live = true;
#endif
    const int y0 = (strip / tiles_x) * pr, x0 = (strip % tiles_x) * PW_X;
    const int y1 = live ? min(y0 + pr, S) : y0;      // one past the last row of this strip
    const long b = blockIdx.z;
    const long n = 1 + (long)S * S;
    const TX* xb = x + b * n * D + (live ? c : 0);
    const TO* gb = dout + b * n * D + (live ? c : 0);
    float acc[49];
#pragma unroll
    for (int t = 0; t < 49; t++) acc[t] = 0.f;
    float bacc = 0.f;
    float win[8][PW_X + 6];      // row r of the grid lives in slot (r - y0 + 3) & 7
    float g[2][PW_X];
    auto load_row = [&](float (&dst)[PW_X + 6], int sy) {
        const bool rok = sy >= 0 && sy < S;
        const TX* row = xb + (1 + (long)(rok ? sy : 0) * S) * D;
#pragma unroll
        for (int u = 0; u < PW_X + 6; u++) {
            const int sx = x0 - 3 + u;
            dst[u] = (rok && sx >= 0 && sx < S) ? ldf(row + (long)sx * D) : 0.f;
        }
    };
    auto load_g = [&](float (&dst)[PW_X], int yy) {
        const bool rok = yy < y1;
        const TO* row = gb + (1 + (long)(rok ? yy : 0) * S) * D;
#pragma unroll
        for (int i = 0; i < PW_X; i++) dst[i] = (rok && x0 + i < S) ? ldf(row + (long)(x0 + i) * D) : 0.f;
    };
    if (live) {
#pragma unroll
        for (int r = 0; r < 7; r++) load_row(win[r], y0 - 3 + r);
        load_g(g[0], y0);
    }
    for (int yb0 = y0; yb0 < y1; yb0 += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int yy = yb0 + j;
            if (yy >= y1) break;
            if (yy + 1 < y1) {                      // the next row step exists: its newest window row and its gradients
                load_row(win[(j + 7) & 7], yy + 4);
                load_g(g[(j + 1) & 1], yy + 1);
            }
#pragma unroll
            for (int i = 0; i < PW_X; i++) bacc += g[j & 1][i];
#pragma unroll
            for (int ky = 0; ky < 7; ky++)
#pragma unroll
                for (int kx = 0; kx < 7; kx++) {
                    float a = acc[ky * 7 + kx];
#pragma unroll
                    for (int i = 0; i < PW_X; i++) a += g[j & 1][i] * win[(j + ky) & 7][i + kx];
                    acc[ky * 7 + kx] = a;
                }
        }
    }
#if PW_RED
    __shared__ float red[3][50][64];
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < 49; t++) red[wave - 1][t][lane] = acc[t];
        red[wave - 1][49][lane] = bacc;
    }
    __syncthreads();
    if (wave == 0 && c < D) {
#pragma unroll
        for (int t = 0; t < 49; t++) atomicAdd(dmerged + t * D + c, acc[t] + red[0][t][lane] + red[1][t][lane] + red[2][t][lane]);
        atomicAdd(dbsum + c, bacc + red[0][49][lane] + red[1][49][lane] + red[2][49][lane]);
    }
#else
#pragma unroll
    for (int t = 0; t < 49; t++) atomicAdd(dmerged + t * D + c, acc[t]);
    atomicAdd(dbsum + c, bacc);
#endif
}

extern "C" int mh_ppeg_wgrad(const void* x, const void* dout, float* dmerged, float* dbsum, int B, int S, int D, int dt_x,
                             int dt_o, mh_stream s) {
    if (B == 0) return MH_OK;
    const int pr = mh_cdiv(S, mh_cdiv(S, PW_ROWS));       // equal strips of at most PW_ROWS rows
#if PW_RED
    dim3 grid(mh_cdiv(D, 64), mh_cdiv(mh_cdiv(S, pr) * mh_cdiv(S, PW_X), 4), B);
#else
    dim3 grid(mh_cdiv(D, 256), mh_cdiv(S, pr) * mh_cdiv(S, PW_X), B);
#endif
#define PW(TX, TO) hipLaunchKernelGGL((ppeg_wgrad_strip_kernel<TX, TO>), grid, dim3(256), 0, (hipStream_t)s, (const TX*)x, (const TO*)dout, dmerged, dbsum, S, D, pr)
    if (dt_x == MH_F32 && dt_o == MH_F32) PW(float, float);
    else if (dt_x == MH_BF16 && dt_o == MH_BF16) PW(bf16_t, bf16_t);
    else if (dt_x == MH_F32 && dt_o == MH_BF16) PW(float, bf16_t);
    else PW(bf16_t, float);
#undef PW
    MH_LAUNCH_CHECK("mh_ppeg_wgrad");
    return MH_OK;
}

// ------------------------------------------------------------------ merged gradient -> the six parameter gradients
// The forward folds the 7x7, 5x5 and 3x3 depthwise kernels (and their biases) into one 7x7 (mh_ppeg_merge), so the gradient
// of the merged kernel IS the 7x7's, its centre 5x5 the 5x5's, its centre 3x3 the 3x3's, and all three biases get dbsum
// (models/mirror.py:324-331).  One launch that ACCUMULATES into the six buffers (the training engine's gradient arena)
// instead of two zero fills, a transposing copy, two window copies, two clones and six `grad += new` launches of autograd.
__global__ __launch_bounds__(256) void ppeg_grad_scatter_kernel(const float* __restrict__ dm, const float* __restrict__ dbs,
                                                                float* __restrict__ dw7, float* __restrict__ dw5, float* __restrict__ dw3,
                                                                float* __restrict__ db7, float* __restrict__ db5, float* __restrict__ db3, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < D * 49) {
        const int c = i / 49, t = i - c * 49, ky = t / 7, kx = t - ky * 7;
        const float v = dm[(long)t * D + c];          // dmerged is tap-major [49, D]
        dw7[i] += v;
        if (ky >= 1 && ky <= 5 && kx >= 1 && kx <= 5) dw5[c * 25 + (ky - 1) * 5 + (kx - 1)] += v;
        if (ky >= 2 && ky <= 4 && kx >= 2 && kx <= 4) dw3[c * 9 + (ky - 2) * 3 + (kx - 2)] += v;
    }
    if (i < D) {
        const float b = dbs[i];
        db7[i] += b;
        db5[i] += b;
        db3[i] += b;
    }
}

extern "C" int mh_ppeg_grad_scatter(const float* dmerged, const float* dbsum, float* dw7, float* dw5, float* dw3, float* db7,
                                    float* db5, float* db3, int D, mh_stream s) {
    MH_REQUIRE(dmerged && dbsum && dw7 && dw5 && dw3 && db7 && db5 && db3, "mh_ppeg_grad_scatter: null pointer");
    if (D == 0) return MH_OK;
    hipLaunchKernelGGL(ppeg_grad_scatter_kernel, dim3(mh_cdiv(D * 49, 256)), dim3(256), 0, (hipStream_t)s, dmerged, dbsum, dw7, dw5, dw3, db7, db5, db3, D);
    MH_LAUNCH_CHECK("mh_ppeg_grad_scatter");
    return MH_OK;
}
