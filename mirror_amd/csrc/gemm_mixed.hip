// GEMM family: f32 operands rounded to bf16 while staged into LDS, bf16 MFMA, f32 accumulate (pinv "fast" path,
// f32 gradients meeting bf16 activations)
#include "gemm_kernel.h"
void gemm_launch_mixed(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s) {
    if (dtC == MH_BF16) launch_l<1, float, float, bf16_t>(a, akc, bkc, batch, s);
    else launch_l<1, float, float, float>(a, akc, bkc, batch, s);
}
