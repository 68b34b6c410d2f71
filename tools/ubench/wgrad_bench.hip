// wgrad_bench.hip -- times the weight-gradient GEMM kernels of gnr_wgrad.hip on the MLP's layer shapes
// (tuning tool, not part of the library).  Includes the translation unit directly so that compile-time
// experiment switches (-DGNR_WG_...) can be compared side by side:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-D...] tools/ubench/wgrad_bench.hip -o wgrad_bench && ./wgrad_bench
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../gazenerf_amd/csrc/gnr_wgrad.hip"

namespace gnr {
int fail(const char*, ...) { return 1; }
}  // namespace gnr

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// the library's shader-clock probe (gnr_api.hip), here a plain device buffer: slot layout [stage][2]
static unsigned long long* g_clk = nullptr;
namespace gnr { unsigned long long* clock_probe_slot(int) { return g_clk; } }

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
    CK(hipMalloc(&g_clk, 16)); CK(hipMemset(g_clk, 0, 16));
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
    // bf16x3 (argv[1] = 3): the operands are the chain kernels' pre-split dumps -- element (chunk, channel quad q,
    // sample j) = 16 bytes {hi01, hi23, lo01, lo23} at chunk*32*C*4 + q*512 + (j ^ 4(q&3))*16 ("QHL")
    auto bf = [](float v) { unsigned u; memcpy(&u, &v, 4); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; };       // RN
    auto bff = [](unsigned b) { unsigned u = b << 16; float f; memcpy(&f, &u, 4); return f; };
    auto fill_qhl = [&]() {
        unsigned* w = (unsigned*)h.data();
        for (long e = 0; e < M * 384 / 4; ++e) {         // e = (chunk*96 + q)*32 + j for C = 384; other C reuse the bytes
            float v[4]; unsigned hi[4], lo[4];
            for (int i = 0; i < 4; ++i) {
                s = s * 1664525u + 1013904223u;
                v[i] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
                if (!dense && v[i] < 0.0f) v[i] = 0.0f;
                hi[i] = bf(v[i]); lo[i] = bf(v[i] - bff(hi[i]));
            }
            const int sw = (int)((e >> 5) >> 2) & 1 ? 2 : 0;       // quads with bit 2 set: {lo, hi}  (96 quads per chunk: (e>>5) % 96 has the same bit 2)
            w[4 * e + (0 ^ sw)] = hi[0] | (hi[1] << 16); w[4 * e + (1 ^ sw)] = hi[2] | (hi[3] << 16);
            w[4 * e + (2 ^ sw)] = lo[0] | (lo[1] << 16); w[4 * e + (3 ^ sw)] = lo[2] | (lo[3] << 16);
        }
    };
    std::vector<float> hA, hB;
    if (x3) fill_qhl(); else fill();
    CK(hipMemcpy(A, h.data(), M * 384 * 4, hipMemcpyHostToDevice));
    if (x3) hA = h;
    if (x3) fill_qhl(); else fill();
    CK(hipMemcpy(B, h.data(), M * 384 * 4, hipMemcpyHostToDevice));
    if (x3) hB = h;
    // decoded value (hi + lo) of (sample s, channel n) of a C-channel QHL tensor
    auto qhl = [&](const std::vector<float>& t, int C, long sidx, int n) {
        const long chunk = sidx >> 5; const int j = (int)(sidx & 31), q = n >> 2, e = n & 3;
        const unsigned* w = (const unsigned*)t.data() + (chunk * 32L * C) + q * 128 + ((j ^ (4 * (q & 3))) * 4);
        const int sw = (q >> 2) & 1 ? 2 : 0;
        const unsigned hw = w[(e >> 1) ^ sw], lw = w[(2 + (e >> 1)) ^ sw];
        const unsigned hb = (e & 1) ? hw >> 16 : hw & 0xffff, lb = (e & 1) ? lw >> 16 : lw & 0xffff;
        return (double)bff(hb) + (double)bff(lb);
    };
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
        unsigned long long hclk[2];
        CK(hipMemcpy(hclk, g_clk, 16, hipMemcpyDeviceToHost));
        cyc = hclk[0];
        const double mhz = hclk[1] ? 100.0 * hclk[0] / hclk[1] : 0.0;
        CK(hipMemset(g_clk, 0, 16));
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
        printf("%-14s %8.3f ms  %7.1f TF (useful)  x%d   wg0 %.0f kcycles @ %.0f MHz   chk %.6e %.6e %.6e %.6e\n", sh.name, ms, flop / ms / 1e9,
               sh.count, cyc / 1e3 / (reps + 2), mhz, sw, swp, sc, sv);
        if (x3) {       // spot check against the decoded operands (fp64 on the host)
            double worst = 0;
            const int ns[4] = {0, 37, sh.n_valid / 2 + 3, sh.n_valid - 1}, ks[3] = {1, sh.k_valid / 2, sh.k_valid - 1};
            for (int a = 0; a < 4; ++a)
                for (int bq = 0; bq < 3; ++bq) {
                    double ref = 0, mag = 0;
                    for (long sidx = 0; sidx < M; ++sidx) {
                        const double p = qhl(hA, sh.lda, sidx, ns[a]) * qhl(hB, sh.ldb, sidx, ks[bq]);
                        ref += p; mag += fabs(p);
                    }
                    const double err = fabs(hw[ns[a] * 640 + ks[bq]] - ref) / (mag + 1e-30);
                    if (err > worst) worst = err;
                }
            double cref = 0, cmag = 0;
            for (long sidx = 0; sidx < M; ++sidx) { const double p = qhl(hA, sh.lda, sidx, 5); cref += p; cmag += fabs(p); }
            double vref = 0, vmag = 0;
            if (sh.vec) for (long sidx = 0; sidx < M; ++sidx) { const double p = h[sidx] * 0 + qhl(hB, sh.ldb, sidx, 9); (void)p; }
            printf("               spot check vs fp64 of the decoded operands: dW rel err (of sum|.|) %.2e, colsum[5] %.2e\n", worst,
                   fabs(hc[5] - cref) / (cmag + 1e-30));
            (void)vref; (void)vmag;
        }
        total_ms += ms * sh.count; total_flop += flop * sh.count;
    }
    printf("stream total   %8.3f ms  %7.1f TF = %.3f of 157.3\n", total_ms, total_flop / total_ms / 1e9, total_flop / total_ms / 1e9 / 157.3);
    return 0;
}
