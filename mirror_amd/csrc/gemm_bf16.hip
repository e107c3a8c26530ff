// GEMM family: bf16 operands, v_mfma_f32_32x32x16_bf16, f32 accumulate, bf16 or f32 output
#include "gemm_kernel.h"
#include <cstdlib>
bool gemm_try_big_bf16(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s);   // gemm_big.hip
void gemm_launch_bf16(GemmArgs& a, int akc, int bkc, int dtC, int batch, hipStream_t s) {
    static const bool big = [] { const char* e = getenv("MH_GEMM_BIG"); return !(e && e[0] == '0'); }();   // A/B switch
    if (big && gemm_try_big_bf16(a, akc, bkc, dtC, batch, s)) return;
    if (dtC == MH_BF16) launch_l<1, bf16_t, bf16_t, bf16_t>(a, akc, bkc, batch, s);
    else launch_l<1, bf16_t, bf16_t, float>(a, akc, bkc, batch, s);
}
