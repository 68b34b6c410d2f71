// Microbenchmark 5 (round 4): what does the fp32 training forward's DUMP PATTERN cost in HBM-side write traffic?
//
// rocprofv3's WRITE_SIZE counted 44.9 GB per launch of fwd16_kernel<true> for 33.9 GB of dumps (1.32x) and 22.4 for 16.9 GB
// in bwd16_chain_kernel (1.33x); the guide calls WRITE_SIZE uncalibrated on gfx950.  This kernel issues exactly that store
// pattern with nothing else in it, on a known byte count, so that `rocprofv3 --pmc WRITE_SIZE` can be calibrated -- and the
// alternatives can be priced before the kernels are changed:
//
//   MODE 0  the shipped pattern: a wave owns 16 samples of a 32-sample chunk (chunk-channel-major rows of 128 B); one
//           `buffer_store_dword nt` writes four 64-byte HALF rows (lane group g -> channel 16 t + 4 g + e); the other half of
//           every row comes from the partner wave of the same workgroup, `delay` x 64 cycles later
//   MODE 1  the same bytes as channel-quad-major 16-byte stores: lane (g, j) writes its four registers of tile t at once,
//           256 contiguous bytes per lane group, 1 KiB per instruction (24 instead of 96 instructions per layer)
//   MODE 2  round 2's pattern: a wave owns all 32 samples, dword stores of two full 128-byte rows per instruction
//   MODE 3  plain coalesced float4 streaming stores (the reference point: the counter on a pattern nobody doubts)
//   MODE 4  MODE 0's addresses, but the two waves of a chunk are ONE wave issuing both halves back to back
// Every mode writes layers x chunks x 384 channels x 32 samples x 4 B, nontemporal, each byte exactly once.
//
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip
//   ./store_pattern [mode] [delay] [chunks] [layers]            (prints ms, GB, GB/s, shader clock)
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out -- ./store_pattern <mode> ...
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int C = 384, CHUNK = 32, NT = C / 16;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void* p) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7ffffff0, 0x00020000); }

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* __restrict__ out, long n_chunks, int layers, int delay, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long layer_floats = n_chunks * CHUNK * C;
    if (MODE == 0 || MODE == 1 || MODE == 4) {
        const int j = lane & 15, g = lane >> 4;
        const long unit = (long)blockIdx.x * 4 + wave;           // MODE 0/1: a 16-sample sub-chunk; MODE 4: a chunk pair-half
        const long chunk = MODE == 4 ? unit : (unit >> 1);
        const int hh = (int)(unit & 1);
        if (chunk >= n_chunks) return;
        if (MODE != 4 && hh == 1)
            for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(1);                   // 64 cycles each
        for (int l = 0; l < layers; ++l) {
            float* base = out + l * layer_floats + chunk * (CHUNK * (long)C);
            const __amdgpu_buffer_rsrc_t rs = rsrc(base);
            const float v = (float)(l + lane);
            if (MODE == 0 || MODE == 4) {
                for (int half = 0; half < (MODE == 4 ? 2 : 1); ++half) {
                    const unsigned lane_off = (unsigned)((4 * g) * CHUNK + 16 * (MODE == 4 ? half : hh) + j) * 4u;
#pragma unroll 4
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + e), rs, lane_off + (unsigned)e * 128u,
                                                                  t * 2048, 2);
                }
            } else {
                // [tile t][half hh][lane group g][sample j][4 channels]: 2 KiB per tile, 1 KiB per (tile, half)
                const unsigned lane_off = (unsigned)(hh * 1024 + g * 256 + j * 16);
#pragma unroll 4
                for (int t = 0; t < NT; ++t) {
                    const f32x4 q = {v, v + 1, v + 2, v + 3};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, q), rs, lane_off, t * 2048, 2);
                }
            }
        }
    } else if (MODE == 2) {
        const int j = lane & 31, h = lane >> 5;
        const long chunk = (long)blockIdx.x * 4 + wave;
        if (chunk >= n_chunks) return;
        for (int l = 0; l < layers; ++l) {
            float* base = out + l * layer_floats + chunk * (CHUNK * (long)C);
            const __amdgpu_buffer_rsrc_t rs = rsrc(base);
            const unsigned lane_off = (unsigned)((4 * h) * CHUNK + j) * 4u;
            const float v = (float)(l + lane);
#pragma unroll 4
            for (int t = 0; t < C / 32; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + r), rs,
                                                          lane_off + (unsigned)((r & 3) + 8 * (r >> 2)) * 128u, t * 4096, 2);
        }
    } else {
        const long chunk = (long)blockIdx.x * 4 + wave;
        if (chunk >= n_chunks) return;
        for (int l = 0; l < layers; ++l) {
            f32x4* base = (f32x4*)(out + l * layer_floats + chunk * (CHUNK * (long)C));
            const f32x4 q = {(float)l, (float)lane, 2.f, 3.f};
#pragma unroll 4
            for (int i = 0; i < CHUNK * C / 4 / 64; ++i) __builtin_nontemporal_store(q, base + i * 64 + lane);
        }
    }
    if (clk && (blockIdx.x & 63) == 0 && threadIdx.x == 0) {
        atomicAdd(clk, __builtin_readcyclecounter() - c0);
        atomicAdd(clk + 1, __builtin_amdgcn_s_memrealtime() - r0);
    }
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int delay = argc > 2 ? atoi(argv[2]) : 0;
    const long n_chunks = argc > 3 ? atol(argv[3]) : 32768;          // 16 384 rays x 64 samples / 32
    const int layers = argc > 4 ? atoi(argv[4]) : 8;
    const size_t bytes = (size_t)layers * n_chunks * CHUNK * C * 4;
    float* out;
    unsigned long long* clk;
    if (hipMalloc(&out, bytes) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(out, 0, bytes);
    const long units = (mode == 0 || mode == 1) ? 2 * n_chunks : n_chunks;
    const unsigned grid = (unsigned)((units + 3) / 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    double mhz = 0;
    for (int it = 0; it < 4; ++it) {
        hipMemset(clk, 0, 16);
        hipEventRecord(a);
        switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, n_chunks, layers, delay, clk); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, n_chunks, layers, delay, clk); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, n_chunks, layers, delay, clk); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, n_chunks, layers, delay, clk); break;
            default: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, n_chunks, layers, delay, clk); break;
        }
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        unsigned long long h[2];
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        if (ms < best) { best = ms; mhz = h[1] ? 100.0 * h[0] / h[1] : 0; }
    }
    // spot check: every float of layer 0 / chunk 0 was written (non-zero pattern or l + lane >= 0 ... count zeros instead)
    printf("mode %d delay %d: %.3f GB written in %.3f ms = %.0f GB/s (best of 4), shader clock %.0f MHz\n", mode, delay, bytes / 1e9, best,
           bytes / 1e6 / best, mhz);
    return 0;
}
