// Error string + device probe for libmirror_hip.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mh_last_error(void) { return g_err; }

// The kernel instance the calling thread's last mh_gemm launched ("gemm_pq_kernel<float,false,false,part>", ...): the launch sites
// record it, so profilers name launches without restating the dispatch rules (mh_gemm_variant_name)
static thread_local char g_variant[160] = "";
void gemm_note_variant(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_variant, sizeof(g_variant), fmt, ap);
    va_end(ap);
}
extern "C" const char* mh_gemm_variant_name(void) { return g_variant; }
extern "C" int mh_version(void) { return 118; }   // 118: mh_gemm_desc.r_bf16 = 2 (f32 R beside a bf16 C); 117: mh_skinny_fwd takes any K / row stride; 116: mh_keymask_plan; 115: lm_scale of mh_layernorm_fwd_lm / _bwd_lm; 114: mlm / heads of mh_pinv_s2_bwd; 113: mlm of mh_nys_sim2; 112: row_mask of mh_layernorm_fwd_lm / _bwd_lm; 111: drop_rows_per_batch of mh_layernorm_bwd_drop; 110: mh_layernorm_bwd_drop, dt_dy of mh_layernorm_bwd_fan, mh_gemm_w4 (experiment); 109 (round 5, late): mh_adam(tick, hole), mh_noise_draws, mh_layernorm_bwd_fan, bias-gradient outputs of mh_mask_apply_bwd / mh_mse_masked_bwd / mh_layernorm_bwd_lm; 107-108 (round 5): mirror_amd/_lib.py refuses any other generation (ABI_VERSION), mh_resconv_bwd, res_conv inside mh_nys_attn3_fwd; 106 (round 4, late): step glue riding on other launches (mh_adam clamp / counter, mh_rownorm_ shadow, scratch_zeroed of the pinv backward tails, addends of mh_skinny_fwd / mh_reparam_bwd), mh_exp_*; 105 (round 4): lm_ld of the mh_nys_* entry points, mh_lm_merge, gadd in dy's dtype, xpm optional;   // 104 (round 3): mh_gemm_desc.epi / row windows, mh_layernorm_fwd_lm(xpm_bf16), mh_skinny_fwd(dt_x, dt_y), mh_pinv_chain_fwd(z0f, stats64), new entry points

// 1 when the library was built with -DMH_EXP (make EXP=1): the timing-experiment switches (MH_EXP_CHAIN_SKIP, and MH_EXP_SKIP
// in the Python host) only exist in such a build; the default build answers 0 and the host refuses the variables.
extern "C" int mh_exp_build(void) {
#ifdef MH_EXP
    return 1;
#else
    return 0;
#endif
}

extern "C" int mh_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        mh_set_error("mh_device_ok: no HIP device visible");
        return 0;
    }
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strstr(p.gcnArchName, "gfx950")) ok++;
    }
    if (!ok) mh_set_error("mh_device_ok: no gfx950 device (this library is MI355X-only)");
    return ok;
}
