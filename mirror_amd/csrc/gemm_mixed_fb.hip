// GEMM family: f32 A x bf16 B operands; f32 operands are rounded to bf16 while staged into LDS, bf16 MFMA, f32 accumulate
// (pinv iterations on f32-stored matrices, f32 gradients meeting bf16 activations: no separate cast kernels)
#include "gemm_kernel.h"
void gemm_launch_mixed_fb(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s) {
    if (dtC == MH_BF16) launch_l<1, float, bf16_t, bf16_t>(a, akc, bkc, batch, s);
    else launch_l<1, float, bf16_t, float>(a, akc, bkc, batch, s);
}
