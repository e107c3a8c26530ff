// GEMM family: bf16 A x f32 B operands; f32 operands are rounded to bf16 while staged into LDS, bf16 MFMA, f32 accumulate
// (pinv iterations on f32-stored matrices, f32 gradients meeting bf16 activations: no separate cast kernels)
#include "gemm_kernel.h"
void gemm_launch_mixed_bf(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s) {
    if (dtC == MH_BF16) launch_l<1, bf16_t, float, bf16_t>(a, akc, bkc, batch, s);
    else launch_l<1, bf16_t, float, float>(a, akc, bkc, batch, s);
}
