// Read-bandwidth probe: every workgroup streams the same `size`-byte buffer (16-byte loads, fully coalesced) `reps` times.
// size <= 4 MB stays in each XCD's L2; 16..128 MB exercises the Infinity Cache; beyond that HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void rd(const u32x4* __restrict__ buf, long n16, int reps, unsigned* out, int shift) {
    u32x4 acc = {0, 0, 0, 0};
    // each WG starts at a different offset so that WGs of one XCD do not all hit the same line at once
    const long start = ((long)blockIdx.x * shift) % n16;
    for (int r = 0; r < reps; r++)
        for (long i = threadIdx.x; i < n16; i += 512 * 4) {
            long j0 = start + i; if (j0 >= n16) j0 -= n16;
            long j1 = j0 + 512; if (j1 >= n16) j1 -= n16;
            long j2 = j1 + 512; if (j2 >= n16) j2 -= n16;
            long j3 = j2 + 512; if (j3 >= n16) j3 -= n16;
            const u32x4 a = buf[j0], b = buf[j1], c = buf[j2], d = buf[j3];
            acc ^= a ^ b ^ c ^ d;
        }
    if (acc[0] == 0x12345678u && acc[1] == 1u) out[0] = acc[2];
}
int main() {
    unsigned* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (long kb : {64L, 256L, 1024L, 2048L, 4096L, 16384L, 65536L, 262144L, 1048576L}) {
        const long bytes = kb * 1024, n16 = bytes / 16;
        u32x4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
        for (int wgs : {64, 256, 512}) {
            const int reps = (int)(kb <= 16384 ? (64L * 1024 * 1024) / bytes * 4 : (kb <= 65536 ? 4 : 1));
            rd<<<wgs, 512>>>(buf, n16, 1, out, 4099);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            rd<<<wgs, 512>>>(buf, n16, reps, out, 4099);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("buf %8ld KB  wgs=%3d  %9.1f us  %7.2f TB/s\n", kb, wgs, ms * 1e3, (double)bytes * reps * wgs / ms / 1e9);
        }
        hipFree(buf);
    }
    return 0;
}
