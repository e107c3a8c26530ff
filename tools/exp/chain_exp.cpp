// Standalone timing harness for pinv_panel.hip variants: hipcc -DEXP=n -I mirror_amd/csrc tools/exp/chain_exp.cpp ...
#include "../../mirror_amd/csrc/pinv_panel.hip"
#include <vector>
#include <cstdlib>
void mh_set_error(const char* fmt, ...) {}
int main(int argc, char** argv) {
    const int BH = argc > 1 ? atoi(argv[1]) : 128, iters = 6, m = 256;
    const size_t mat = (size_t)m * m;
    bf16_t *xp, *saved, *zfT, *work, *up;
    float *dX, *dz0;
    hipMalloc(&xp, BH * mat * 2); hipMalloc(&saved, iters * 4 * BH * mat * 2); hipMalloc(&zfT, BH * mat * 2);
    hipMalloc(&work, iters * 4 * BH * mat * 2); hipMalloc(&up, BH * mat * 2); hipMalloc(&dX, BH * mat * 4); hipMalloc(&dz0, BH * mat * 4);
    std::vector<bf16_t> h(BH * mat);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0x3b00 + (rand() & 0xff);   // small positive bf16
    hipMemcpy(xp, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(saved, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(up, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; pass++) {
        for (int r = 0; r < 3; r++) {
            if (pass == 0) mh_pinv_chain_fwd(xp, saved, zfT, BH, m, iters, 0);
            else mh_pinv_chain_bwd(xp, saved, up, work, dX, dz0, BH, m, iters, 0);
        }
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        const int reps = 10;
        for (int r = 0; r < reps; r++) {
            if (pass == 0) mh_pinv_chain_fwd(xp, saved, zfT, BH, m, iters, 0);
            else mh_pinv_chain_bwd(xp, saved, up, work, dX, dz0, BH, m, iters, 0);
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("EXP=%d BH=%d %s %.1f us\n", EXP, BH, pass ? "bwd" : "fwd", ms / reps * 1e3);
    }
    return 0;
}
