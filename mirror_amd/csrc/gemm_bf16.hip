// GEMM family: bf16 operands, v_mfma_f32_32x32x16_bf16, f32 accumulate, bf16 or f32 output
#include "gemm_kernel.h"
#include <cstdlib>
bool gemm_try_big_bf16(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s);   // gemm_big.hip
void gemm_launch_bf16(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s) {
    if (gemm_try_big_bf16(a, akc, bkc, dtC, batch, s)) return;
    if (dtC == MH_BF16) launch_l<1, bf16_t, bf16_t, bf16_t>(a, akc, bkc, batch, s);
    else launch_l<1, bf16_t, bf16_t, float>(a, akc, bkc, batch, s);
}
