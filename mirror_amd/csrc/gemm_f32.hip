// GEMM family: exact-f32 MFMA (v_mfma_f32_32x32x2_f32), f32 operands and output (parity policy, pinv in "bf16" policy)
#include "gemm_kernel.h"
void gemm_launch_f32(GemmArgs& a, int akc, int bkc, int batch, hipStream_t s) {
    launch_l<0, float, float, float>(a, akc, bkc, batch, s);
}
