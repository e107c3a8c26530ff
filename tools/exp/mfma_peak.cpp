// MFMA issue-rate probe: how many v_mfma_f32_32x32x16_bf16 per second does the chip sustain with no memory traffic,
// as a function of the number of workgroups (one per CU) and waves per SIMD?  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, int with_lds) {
    __shared__ bf16x8 lds[512 * 4];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
    lds[threadIdx.x] = a; lds[threadIdx.x + 512] = b; lds[threadIdx.x + 1024] = a; lds[threadIdx.x + 1536] = b;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
        if (with_lds) {   // per 8 MFMAs: 6 ds_read_b128 (the GEMM's fragment traffic: 12 per 32 MFMAs -> 3 per 8)
            a = lds[(threadIdx.x + it) & 511];
            b = lds[512 + ((threadIdx.x + 2 * it) & 511)];
        }
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int lds = 0; lds < 2; lds++)
        for (int threads : {256, 512})
            for (int wgs : {32, 64, 128, 192, 256, 512}) {
                mfma_loop<8><<<wgs, threads>>>(out, 100, lds);
                hipDeviceSynchronize();
                hipEventRecord(e0, 0);
                mfma_loop<8><<<wgs, threads>>>(out, iters, lds);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flops = (double)wgs * (threads / 64) * iters * 8 * 32768.0;
                printf("lds=%d threads=%d wgs=%3d  %8.1f us  %8.1f TF/s  (%.2f TF/s per WG)\n", lds, threads, wgs, ms * 1e3, flops / ms / 1e9,
                       flops / ms / 1e9 / wgs);
            }
    return 0;
}
