// Microbenchmark 4 (round 3): do a wave's end-of-tile STORES hide under the MFMAs of the other waves on its SIMD?
//
// The upsampler GEMM (gnr_conv16.hip) runs its main loops at 0.85-0.90 of the fp32-MFMA peak, but its epilogue stores add
// their full duration on top (profiles/r3_n1_conv16_experiments.txt).  This kernel is that pattern with nothing else in
// it: every wave alternates NM register-only MFMAs (v_mfma_f32_16x16x4_f32, 8 accumulator tiles = the (2,4) instance)
// with a burst of NS 1-KiB store instructions (global_store_dwordx4, each (wave, repetition) its own region of a buffer
// far larger than L2 + Infinity Cache), W waves per SIMD, all SIMDs of the chip.
//   time(MFMA only), time(stores only), time(both): 'both' ~ max(...) means the stores hide, ~ sum means they do not.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_store mfma_store.hip && ./mfma_store
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int W, bool LOADS, int MODE>       // MODE 0: store the accumulators; 1: store a register the MFMAs do not write; 2: + wait for the stores; 3: sleep instead of storing;
                                             // 4: SLEEP for the tile's duration instead of its MFMAs (is it the MFMAs or the gap?); 5: one 1-KiB load per 64 MFMAs (memory never idle);
                                             // 6: the same bytes as four dword stores per 1 KiB (256 B per instruction, one data register each); 7: as two b64 stores
__global__ __launch_bounds__(256, W) void k(float* __restrict__ out, const float* __restrict__ in, int reps, int nm, int ns,
                                            long region_floats, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + lane * 1e-3f, b = 0.5f + lane * 1e-4f;
    f32x4 trick = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        if (MODE == 4) {
            for (int m = 0; m < nm; m += 8) __builtin_amdgcn_s_sleep(4 * W);          // 8 MFMAs x 32 cycles x W waves = 256 W cycles
        } else {
            for (int m = 0; m < nm; m += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = MF16(a, b, acc[i]);
                if (MODE == 5 && (m & 63) == 0) trick += *(const f32x4*)(in + ((wave * 977 + r * 131 + m) % (region_floats / 256)) * 256 + lane * 4);
            }
        }
        float* dst = out + ((wave * reps + r) % (region_floats / (2048L))) * 2048L + lane * 4;      // 8 KiB per (wave, rep)
        f32x4 extra = f32x4{0.f, 0.f, 0.f, 0.f};
        if (LOADS) {
            const float* src = in + ((wave * reps + r) % (region_floats / 2048L)) * 2048L + lane * 4;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s < ns) extra += *(const f32x4*)(src + s * 256);
        }
        if (MODE == 3) {
            for (int s = 0; s < ns; ++s) __builtin_amdgcn_s_sleep(16);          // 8 x 1024 cycles for ns = 8
        } else {
            if (MODE == 6) {
                float* d1 = dst - lane * 4 + lane;                 // dense dwords: lane l writes float l of each 64-float row
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (s < ns) {
                        const f32x4 v = acc[s] + extra;
                        d1[s * 256] = v.x; d1[s * 256 + 64] = v.y; d1[s * 256 + 128] = v.z; d1[s * 256 + 192] = v.w;
                    }
            } else if (MODE == 7) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                float* d2 = dst - lane * 4 + lane * 2;
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (s < ns) {
                        const f32x4 v = acc[s] + extra;
                        *(f32x2*)(d2 + s * 256) = f32x2{v.x, v.y}; *(f32x2*)(d2 + s * 256 + 128) = f32x2{v.z, v.w};
                    }
            } else {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s < ns) *(f32x4*)(dst + s * 256) = (MODE == 0 ? acc[s] : f32x4{a, b, a, b}) + extra;
            }
            if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    float chk = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) chk += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (chk + trick.x == 123.456f) out[0] = chk;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

static unsigned long long* g_clk;
static double g_mhz;
template <int W, bool LOADS, int MODE>
static float run(float* out, const float* in, int reps, int nm, int ns, long region_floats) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * W;                       // W workgroups of 4 waves per CU: W waves per SIMD
    hipLaunchKernelGGL((k<W, LOADS, MODE>), dim3(blocks), dim3(256), 0, 0, out, in, 2, nm, ns, region_floats, (unsigned long long*)nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<W, LOADS, MODE>), dim3(blocks), dim3(256), 0, 0, out, in, reps, nm, ns, region_floats, g_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, g_clk, 16, hipMemcpyDeviceToHost);
    g_mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;       // s_memrealtime ticks at 100 MHz
    return ms;
}

template <int W, bool LOADS, int MODE = 0>
static void table(float* out, const float* in, long region_floats, int nm) {
    const int reps = 64;
    const float t_m = run<W, LOADS, MODE>(out, in, reps, nm, 0, region_floats);
    const double flop = 256.0 * 4 * W * reps * (double)nm * 2048.0;
    printf("mode %d, %d waves/SIMD, %4d MFMAs per tile%s: MFMA only %.3f ms (%.1f TF, %.0f MHz)", MODE, W, nm, LOADS ? ", epilogue loads + stores" : ", epilogue stores", t_m,
           flop / t_m / 1e9, g_mhz);
    for (int ns : {2, 8}) {
        const float t_s = run<W, LOADS, MODE>(out, in, reps, 0, ns, region_floats);
        const float t_b = run<W, LOADS, MODE>(out, in, reps, nm, ns, region_floats);
        const double gb = 256.0 * 4 * W * reps * ns * 1024.0 / 1e9;
        printf(" | %d KiB/tile: stores only %.3f ms (%.2f TB/s), both %.3f ms (%.0f MHz) = MFMA + %.0f %% of the stores' time", ns, t_s, gb / t_s, t_b, g_mhz,
               100.0 * (t_b - t_m) / t_s);
    }
    printf("\n");
}

int main() {
    const long region_floats = 1L << 30;              // 4 GiB
    float *out, *in;
    hipMalloc(&out, region_floats * 4);
    hipMalloc(&in, region_floats * 4);
    hipMemset(out, 0, region_floats * 4);
    hipMemset(in, 0, region_floats * 4);
    hipMalloc(&g_clk, 16);
    hipMemset(g_clk, 0, 16);
    for (int nm : {128, 1056}) {
        table<1, false>(out, in, region_floats, nm);
        table<2, false>(out, in, region_floats, nm);
        table<4, false>(out, in, region_floats, nm);
        table<5, false>(out, in, region_floats, nm);
        table<5, true>(out, in, region_floats, nm);
        table<5, false, 1>(out, in, region_floats, nm);
        table<5, false, 2>(out, in, region_floats, nm);
        table<5, false, 3>(out, in, region_floats, nm);
        table<5, false, 4>(out, in, region_floats, nm);
        table<5, false, 5>(out, in, region_floats, nm);
        table<5, false, 6>(out, in, region_floats, nm);
        table<5, false, 7>(out, in, region_floats, nm);
        table<2, false, 6>(out, in, region_floats, nm);
    }
    return 0;
}
