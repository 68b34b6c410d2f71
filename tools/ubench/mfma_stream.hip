// Microbenchmark: how close can one wave per SIMD get to the fp32-MFMA issue rate with the
// chain kernel's instruction pattern?  (scratch tool; not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const f32x4* __restrict__ W, float* out, int iters, long wrows) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[12];
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float b[16];
    for (int r = 0; r < 16; ++r) b[r] = 1.0f + lane * 1e-3f + r;
    f32x4 ring[8];
    const f32x4* p = W + lane;
    for (int i = 0; i < 8; ++i) ring[i] = MODE >= 1 ? p[i * 64] : f32x4{1.f, 2.f, 3.f, 4.f};
    unsigned row = 0; const unsigned wmask = (unsigned)wrows - 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {            // one "k-group": 12 tiles, in pairs
#pragma unroll
            for (int nt = 0; nt < 12; nt += 2) {
                const int i = g * 12 + nt, i1 = i + 1;
                f32x4 a0 = ring[i % 8], a1 = ring[i1 % 8];
                if (MODE == 1) { a0 = f32x4{1.f, 2.f, 3.f, 4.f}; a1 = a0; }   // loads issued but not consumed
                acc[nt] = MF(a0.x, b[4 * g + 0], acc[nt]);
                acc[nt + 1] = MF(a1.x, b[4 * g + 0], acc[nt + 1]);
                acc[nt] = MF(a0.y, b[4 * g + 1], acc[nt]);
                acc[nt + 1] = MF(a1.y, b[4 * g + 1], acc[nt + 1]);
                acc[nt] = MF(a0.z, b[4 * g + 2], acc[nt]);
                acc[nt + 1] = MF(a1.z, b[4 * g + 2], acc[nt + 1]);
                acc[nt] = MF(a0.w, b[4 * g + 3], acc[nt]);
                acc[nt + 1] = MF(a1.w, b[4 * g + 3], acc[nt + 1]);
                if (MODE >= 1 && !(MODE == 4 && (threadIdx.x >> 6) != 0)) {
                    ring[i % 8] = p[((row + i + 8) & wmask) * 64];
                    ring[i1 % 8] = p[((row + i1 + 8) & wmask) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        row += 48;
        if (MODE == 3) row = 0;      // tiny working set: always the same 56 rows (L1-resident)
    }
    float s = 0;
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (MODE == 1) for (int i = 0; i < 8; ++i) s += ring[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, const f32x4* W, float* out, long wrows) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, W, out, 10, wrows);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, W, out, iters, wrows);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 4 * iters * 192.0;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", name, ms, mfma * 4096 / ms / 1e9, mfma * 4096 / ms / 1e9 / 157.3 * 100);
}

int main() {
    const long maxrows = 16384;
    f32x4* W; float* out;
    hipMalloc(&W, maxrows * 64 * sizeof(f32x4) + (1 << 20)); hipMalloc(&out, 256 * 8 * 256 * 4);
    std::vector<float> h(maxrows * 256, 0.5f);
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("registers only (12 acc, paired)", W, out, maxrows);
    char name[128];
    for (long kb : {16L, 64L, 256L, 1024L, 2048L, 4096L, 8192L, 16384L}) {
        snprintf(name, sizeof name, "A from ring, cyclic stream of %5ld KB", kb);
        run<2>(name, W, out, kb);
    }
    run<4>("8 MB stream, only wave 0 of each WG loads", W, out, 8192);
    return 0;
}
