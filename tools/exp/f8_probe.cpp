// Operand-layout probe for v_mfma_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3): which (row, k) does byte `e` of lane `l` hold?
// Hypotheses for both operands (A as [i][k], B as [n][k]):  H1: k = 32 (l >> 5) + e        (32 consecutive k per lane)
//                                                            H2: k = 16 (l >> 5) + (e & 15) + 32 (e >> 4)   (two 16-k groups)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const unsigned char* A, const unsigned char* B, float* C, int hyp) {
    const int l = threadIdx.x, r = l & 31, hl = l >> 5;
    unsigned char a[32], b[32];
    for (int e = 0; e < 32; e++) {
        const int k = hyp == 1 ? 32 * hl + e : 16 * hl + (e & 15) + 32 * (e >> 4);
        a[e] = A[r * 64 + k];
        b[e] = B[r * 64 + k];
    }
    i32x8 av, bv;
    for (int w = 0; w < 8; w++) {
        av[w] = a[4 * w] | (a[4 * w + 1] << 8) | (a[4 * w + 2] << 16) | (a[4 * w + 3] << 24);
        bv[w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, 127, 0, 127);   // fp8 x fp8, scales 2^0
    // C layout as for the other 32x32 MFMAs: column = l & 31, row = 8 (e >> 2) + 4 hl + (e & 3)
    for (int e = 0; e < 16; e++) C[(8 * (e >> 2) + 4 * hl + (e & 3)) * 32 + r] = acc[e];
}
static float f8(unsigned char v) {   // OCP e4m3fn decode
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}
int main() {
    unsigned char hA[32 * 64], hB[32 * 64];
    srand(1);
    for (int i = 0; i < 32 * 64; i++) { hA[i] = 0x30 + (rand() % 24); hB[i] = (rand() & 1 ? 0x80 : 0) | (0x28 + (rand() % 24)); }   // |x| in [0.25, 2)
    double ref[32 * 32];
    for (int i = 0; i < 32; i++)
        for (int n = 0; n < 32; n++) {
            double s = 0;
            for (int k = 0; k < 64; k++) s += (double)f8(hA[i * 64 + k]) * f8(hB[n * 64 + k]);
            ref[i * 32 + n] = s;
        }
    unsigned char *dA, *dB; float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int hyp = 1; hyp <= 2; hyp++) {
        probe<<<1, 64>>>(dA, dB, dC, hyp);
        float hC[32 * 32];
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        double err = 0, mag = 0;
        for (int i = 0; i < 32 * 32; i++) { err = fmax(err, fabs(hC[i] - ref[i])); mag = fmax(mag, fabs(ref[i])); }
        printf("hypothesis %d: max |C - ref| = %.4g (max |ref| %.4g)  C[0][0]=%.4f ref=%.4f\n", hyp, err, mag, hC[0], ref[0]);
    }
    return 0;
}
