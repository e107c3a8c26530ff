// LDS-DMA smoke test: global -> LDS by global_load_lds_dwordx4 (1 KiB per wave instruction), LDS -> global by ordinary stores.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/glds_test tools/exp/glds_test.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, char* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) char img[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* g = src + (long)blockIdx.x * 131072;
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
        const char* gq = g + q * 32768 + wave * 8192 + lane * 16;
        char* lq = img + q * 32768 + wave * 8192;
#pragma unroll
        for (int i = 0; i < 8; i += 4) {
            const __attribute__((address_space(1))) void* gp = (const __attribute__((address_space(1))) void*)(gq + i * 1024);
            __attribute__((address_space(3))) void* lp = (__attribute__((address_space(3))) void*)(lq + i * 1024);
            __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
            __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
            __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    char* d = dst + (long)blockIdx.x * 131072;
    for (int i = tid; i < 8192; i += 256) *reinterpret_cast<u32x4*>(d + i * 16) = *reinterpret_cast<const u32x4*>(img + ((i * 16) ^ 0));
}
int main() {
    const int nb = 128; const size_t n = (size_t)nb * 131072;
    std::vector<unsigned> h(n / 4), o(n / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)(i * 2654435761u);
    char *s, *d; hipMalloc(&s, n); hipMalloc(&d, n); hipMemcpy(s, h.data(), n, hipMemcpyHostToDevice); hipMemset(d, 0, n);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) { hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(nb), dim3(256), 131072, 0, s, d); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("run %d: %.1f us\n", r, ms * 1e3); }
    hipMemcpy(o.data(), d, n, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < h.size(); i++) bad += h[i] != o[i];
    printf("mismatches: %zu of %zu (%s)\n", bad, h.size(), hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
