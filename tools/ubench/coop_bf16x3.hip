// Microbenchmark for the NEXT design of the bf16x3 chain ("workgroup-cooperative layers"): what MFMA utilisation does
// the instruction mix reach if the four waves of a workgroup split the OUTPUT channels of a layer (3 n-tiles each)
// over S_TILES x 32 shared samples, stream only their own weight rows straight into registers, and exchange
// activations through LDS?  Synthetic data, real instruction counts per K=16 step and per wave:
//     6 buffer_load_dwordx4 (3 n-tiles x hi/lo, double-buffered), 2*S_TILES ds_read_b128 (B operands),
//     9*S_TILES v_mfma_f32_32x32x16_bf16;  per layer (24 K-steps): conversion of the wave's 3 x S_TILES output
//     tiles (relu + hi/lo split) + 4 ds_write_b128 per tile + 2 barriers.
// Compare with the current kernel (register chain + LDS weight ring): 0.50-0.52 of the bf16x3 peak.
// Measured (MI355X): S=64 45 %, S=96 56 % (68 % without the conversion), 8 waves x 1 sample tile 33 %.
// hipcc --offload-arch=gfx950 -O3 -o coop_bf16x3 coop_bf16x3.hip && ./coop_bf16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    const f32x2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, bf16x2));
}

// WAVES = 4: one wave per SIMD, every wave covers all S_TILES sample tiles of the workgroup.
// WAVES = 8: two waves per SIMD; wave = (n-group, sample half), each covers S_TILES of the 2*S_TILES tiles in LDS.
template <int S_TILES, bool CONVERT, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, 1) void coop(const u32x4* __restrict__ W, float* out, int layers, unsigned rows_mask) {
    extern __shared__ __attribute__((aligned(16))) u32x4 act_all[];     // [sample half][K-step 24][s-tile][hi/lo][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, half_id = tid >> 8;
    for (int i = tid; i < (WAVES / 4) * 24 * S_TILES * 2 * 64; i += 64 * WAVES) act_all[i] = u32x4{0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u};
    __syncthreads();
    u32x4* act = act_all + half_id * (24 * S_TILES * 2 * 64);
    f32x16 acc[3][S_TILES];
    for (int n = 0; n < 3; ++n) for (int s = 0; s < S_TILES; ++s) for (int r = 0; r < 16; ++r) acc[n][s][r] = 0.f;
    // this wave's private weight stream: rows of 1 KiB, 6 per K-step
    const u32x4* p = W + lane;
    unsigned row = (unsigned)(blockIdx.x * 4 + wave) * 97u;
    u32x4 w[2][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) w[0][i] = p[((row + i) & rows_mask) * 64];
    for (int l = 0; l < layers; ++l) {
#pragma unroll 2
        for (int ks = 0; ks < 24; ks += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int cur = half, nxt = half ^ 1;
                row += 6;
#pragma unroll
                for (int i = 0; i < 6; ++i) w[nxt][i] = p[((row + i) & rows_mask) * 64];
                u32x4 bh[S_TILES], bl[S_TILES];
#pragma unroll
                for (int s = 0; s < S_TILES; ++s) {
                    bh[s] = act[(((ks + half) * S_TILES + s) * 2 + 0) * 64 + lane];
                    bl[s] = act[(((ks + half) * S_TILES + s) * 2 + 1) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < S_TILES; ++s)
#pragma unroll
                    for (int n = 0; n < 3; ++n) {
                        acc[n][s] = mf(w[cur][2 * n], bh[s], acc[n][s]);
                        acc[n][s] = mf(w[cur][2 * n + 1], bh[s], acc[n][s]);
                        acc[n][s] = mf(w[cur][2 * n], bl[s], acc[n][s]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // layer boundary: everyone has read the activations -> convert own outputs -> publish -> next layer
        __builtin_amdgcn_s_barrier();
        if (CONVERT) {
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int s = 0; s < S_TILES; ++s) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        u32x4 h, lo;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float a = acc[n][s][8 * u + 2 * q], b = acc[n][s][8 * u + 2 * q + 1];
                            a = __builtin_amdgcn_fmed3f(a, 0.0f, 1e30f); b = __builtin_amdgcn_fmed3f(b, 0.0f, 1e30f);
                            unsigned hh, ll;
                            split_pair(a * 1e-3f, b * 1e-3f, hh, ll);
                            h[q] = hh; lo[q] = ll;
                        }
                        const int kstep = (2 * (3 * wave + n) + u) % 24;
                        act[((kstep * S_TILES + s) * 2 + 0) * 64 + lane] = h;
                        act[((kstep * S_TILES + s) * 2 + 1) * 64 + lane] = lo;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[n][s][r] = 0.5f;      // "bias"
                }
        }
        __builtin_amdgcn_s_barrier();
    }
    float sum = 0.f;
    for (int n = 0; n < 3; ++n) for (int s = 0; s < S_TILES; ++s) for (int r = 0; r < 16; ++r) sum += acc[n][s][r];
    out[blockIdx.x * 256 + tid] = sum + w[0][0].x;
}

template <int S_TILES, bool CONVERT, int WAVES = 4>
void run(const char* name, const u32x4* W, float* out, unsigned rows) {
    const int layers = 40, blocks = 256 * 4;
    const size_t lds = (size_t)(WAVES / 4) * 24 * S_TILES * 2 * 64 * 16;
    hipFuncSetAttribute((const void*)coop<S_TILES, CONVERT, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((coop<S_TILES, CONVERT, WAVES>), dim3(blocks), dim3(64 * WAVES), lds, 0, W, out, layers, rows - 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * WAVES * layers * 24 * 9 * S_TILES;         // per wave-instruction: 32x32x16 x 2 FLOP
    const double tf = mfma * 32768.0 / (ms * 1e-3) / 1e12;                        // raw bf16 TFLOP/s
    printf("%-58s %7.3f ms  %7.1f TF raw bf16 = %5.1f %% of 2516.8  (LDS %zu KiB)\n", name, ms, tf, tf / 2516.8 * 100, lds / 1024);
}

int main() {
    const unsigned rows = 8192;                       // 8 MiB cyclic weight buffer (L2/MALL resident like the real stream)
    u32x4* W; float* out;
    hipMalloc(&W, (size_t)rows * 64 * 16 + (1 << 20)); hipMalloc(&out, 1024 * 256 * 4);
    std::vector<unsigned> h((size_t)rows * 256, 0x3c003c00u);
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<2, true>("S=64  (2 sample tiles/wave), with conversion + exchange", W, out, rows);
    run<3, true>("S=96  (3 sample tiles/wave), with conversion + exchange", W, out, rows);
    run<3, false>("S=96, no conversion (loads + reads + MFMA + barriers only)", W, out, rows);
    run<1, true, 8>("S=64, 8 waves (2 per SIMD), 1 sample tile per wave", W, out, rows);
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
