// Standalone timing harness for gemm_big.hip main-loop experiments:
//   hipcc -O3 --offload-arch=gfx950 -DGEMM_EXP=n -I mirror_amd/csrc tools/exp/gemm_exp.cpp -o tools/exp/bin/gemm_exp_n
#include "../../mirror_amd/csrc/gemm_big.hip"
#include <vector>
#include <cstdlib>
void mh_set_error(const char* fmt, ...) {}
static void run(int M, int N, int Kd, int akc, int bkc) {
    bf16_t *A, *B, *C;
    hipMalloc(&A, (size_t)M * Kd * 2); hipMalloc(&B, (size_t)N * Kd * 2); hipMalloc(&C, (size_t)M * N * 2);
    hipMemset(A, 0x3c, (size_t)M * Kd * 2); hipMemset(B, 0x3c, (size_t)N * Kd * 2);
    GemmArgs a = {};
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = Kd;
    a.lda = akc ? Kd : M; a.ldb = bkc ? Kd : N; a.ldc = N;
    a.batch2 = 1; a.alpha = 1.f; a.split_k = 1; a.k_per_split = Kd; a.vecA = a.vecB = a.vecC = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) gemm_try_big_bf16(a, akc, bkc, MH_BF16, 1, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int r = 0; r < reps; r++) gemm_try_big_bf16(a, akc, bkc, MH_BF16, 1, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const int tiles = ((M + 255) / 256) * (N / 256);
    printf("EXP=%d M=%6d N=%5d K=%5d kc=%d%d  %8.1f us  %7.1f TF/s  (%.2f us per k-step per round)\n", GEMM_EXP, M, N, Kd, akc, bkc, ms * 1e3,
           2.0 * M * N * Kd / ms / 1e9, ms * 1e3 / ((tiles + 255) / 256) / (Kd / 64));
    hipFree(A); hipFree(B); hipFree(C);
}
int main() {
    run(69632, 1536, 512, 1, 1);
    run(65536, 512, 1024, 1, 1);
    run(69632, 512, 1536, 1, 0);
    run(8192, 8192, 8192, 1, 1);
    return 0;
}
