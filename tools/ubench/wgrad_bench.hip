// wgrad_bench.hip -- times the weight-gradient GEMM kernels of gnr_wgrad.hip on the MLP's layer shapes
// (tuning tool, not part of the library).  Includes the translation unit directly so that compile-time
// experiment switches (-DGNR_WG_...) can be compared side by side:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-D...] tools/ubench/wgrad_bench.hip -o wgrad_bench && ./wgrad_bench
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../gazenerf_amd/csrc/gnr_wgrad.hip"

namespace gnr {
int fail(const char*, ...) { return 1; }
}  // namespace gnr

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const long M = 16384L * 64;                 // one bench micro-batch: 16 384 rays x 64 samples
    const long chunks = M / 32;
    const bool x3 = argc > 1 && atoi(argv[1]) == 3;
    struct Shape { const char* name; int lda, n_valid, ldb, k_valid; bool vec; int count; };
    // per stream and micro-batch: 7 plain 384^2 layers + RGB_layer_0 (with the density rider), RGB_layer_1,
    // RGB_layer_2, and the encoding columns of layers 0 and 5
    const Shape shapes[] = {{"384x384", 384, 384, 384, 384, false, 7}, {"384x384+vec", 384, 384, 384, 384, true, 1},
                            {"RGB1 192x384", 192, 192, 384, 384, false, 1}, {"RGB2 258x192", 288, 258, 192, 192, false, 1},
                            {"enc 384x64", 384, 384, 64, 64, false, 2}};
    float *A, *B, *dW, *cs, *vec, *vout, *scratch;
    CK(hipMalloc(&A, M * 384 * 4)); CK(hipMalloc(&B, M * 384 * 4)); CK(hipMalloc(&dW, 384 * 640 * 4));
    CK(hipMalloc(&cs, 4096 * 4)); CK(hipMalloc(&vec, M * 4)); CK(hipMalloc(&vout, 4096 * 4));
    CK(hipMalloc(&scratch, gnr::wgrad_scratch_floats() * 4));
    // data like the real operands: post-ReLU activations and masked gradients are ~half zeros (the chip clocks to
    // its power budget: dense random operands run the same kernel at a lower clock).  argv[2] = 1: dense instead.
    const bool dense = argc > 2 && atoi(argv[2]) == 1;
    std::vector<float> h(M * 384);
    unsigned s = 12345;
    auto fill = [&]() {
        for (auto& v : h) {
            s = s * 1664525u + 1013904223u;
            v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
            if (!dense && v < 0.0f) v = 0.0f;
        }
    };
    fill();
    CK(hipMemcpy(A, h.data(), M * 384 * 4, hipMemcpyHostToDevice));
    fill();
    CK(hipMemcpy(B, h.data(), M * 384 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vec, h.data(), M * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total_ms = 0, total_flop = 0;
    for (const Shape& sh : shapes) {
        auto run = [&]() {
            gnr::launch_wgrad(A, sh.lda, sh.n_valid, B, sh.ldb, sh.k_valid, 1, chunks, dW, 640, 0, 0, cs, 384,
                              sh.vec ? vec : nullptr, sh.vec ? vout : nullptr, scratch, 0, x3);
        };
        run(); run();
        CK(hipDeviceSynchronize());
        const int reps = 10;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) run();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double flop = 2.0 * M * sh.n_valid * sh.k_valid;
        unsigned long long cyc = 0;
#ifdef GNR_WG_CLOCK
        CK(hipMemcpy(&cyc, scratch + (size_t)1024 * 16384 + (size_t)1024 * 192 + (size_t)1024 * 192 - 2, 8, hipMemcpyDeviceToHost));
#endif
        // checksums, to compare builds: sum |dW|, sum |colsum|, sum |vec_out|, and a position-weighted sum of dW
        std::vector<float> hw(384 * 640), hc(384), hv(384);
        CK(hipMemcpy(hw.data(), dW, hw.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc.data(), cs, 384 * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hv.data(), vout, 384 * 4, hipMemcpyDeviceToHost));
        double sw = 0, swp = 0, sc = 0, sv = 0;
        for (int n = 0; n < sh.n_valid; ++n)
            for (int k = 0; k < sh.k_valid; ++k) { const double v = hw[n * 640 + k]; sw += fabs(v); swp += v * ((n * 131 + k * 7) % 97); }
        for (int n = 0; n < sh.n_valid; ++n) sc += fabs(hc[n]) + 1e-3 * n * hc[n];
        if (sh.vec) for (int k = 0; k < sh.k_valid; ++k) sv += fabs(hv[k]) + 1e-3 * k * hv[k];
        printf("%-14s %8.3f ms  %7.1f TF (useful)  x%d   wg0 %.0f kcycles   chk %.6e %.6e %.6e %.6e\n", sh.name, ms, flop / ms / 1e9,
               sh.count, cyc / 1e3, sw, swp, sc, sv);
        total_ms += ms * sh.count; total_flop += flop * sh.count;
    }
    printf("stream total   %8.3f ms  %7.1f TF = %.3f of 157.3\n", total_ms, total_flop / total_ms / 1e9, total_flop / total_ms / 1e9 / 157.3);
    return 0;
}
