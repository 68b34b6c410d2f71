// Microbenchmark: what does it cost a wave (one per SIMD, MFMA-bound) to put 2 KiB per phase into the workgroup's LDS
// ring -- by LDS-DMA (buffer_load_dwordx4 ... lds, what fwd3/bwd3 do) or by buffer_load_dwordx4 into registers +
// ds_write_b128 two phases later?  Phase = barrier, feed, 8 ds_read_b128 of ring rows, 12 v_mfma_f32_32x32x16_bf16
// (the real kernels' per-phase mix without the activation conversion).
// hipcc --offload-arch=gfx950 -O3 -o feed_cost feed_cost.hip && ./feed_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }

// MODE 0: no feed (ring filled once)   1: LDS-DMA, 4 batches in flight   2: register loads, written two phases later
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const u32x4* __restrict__ W, unsigned bytes, float* out, int phases) {
    extern __shared__ __attribute__((aligned(16))) u32x4 ring[];       // 6 slots x 8 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 6 * 512; i += 256) ring[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    f32x16 acc[12];
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const u32x4 b0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    i32x4 rs;
    const unsigned long long a = (unsigned long long)W;
    rs.x = (int)(unsigned)a; rs.y = (int)(unsigned)(a >> 32); rs.z = (int)bytes; rs.w = 0x00020000;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)bytes, 0x00020000);
    const unsigned voff = lane * 16u + wave * 2048u;
    unsigned soff = (blockIdx.x * 8192u * 13u) % (bytes - 65536u);
    unsigned wr = wave * 2048u, rd = 0;
    u32x4 st[2][2];
    if (MODE == 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            st[s][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
            st[s][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024u, soff, 0));
            soff += 8192u;
        }
    }
    for (int ph = 0; ph < phases; ph += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            if (MODE == 1) wait_vm<6>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (MODE == 1) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                             "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                             "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n\t"
                             "s_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(rs), "s"(soff), "s"(wr) : "memory");
                soff += 8192u; if (soff > bytes - 65536u) soff = 0;
                wr += 8192u; if (wr >= 6 * 8192u) wr -= 6 * 8192u;
            } else if (MODE == 2) {
                char* dst = (char*)ring + wr + lane * 16;
                *(u32x4*)dst = st[par][0];
                *(u32x4*)(dst + 1024) = st[par][1];
                st[par][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
                st[par][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024u, soff, 0));
                soff += 8192u; if (soff > bytes - 65536u) soff = 0;
                wr += 8192u; if (wr >= 6 * 8192u) wr -= 6 * 8192u;
            }
            const char* src = (const char*)ring + rd + lane * 16;
            u32x4 g[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = *(const u32x4*)(src + i * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[3 * i + 0] = mf(g[2 * i], b0, acc[3 * i + 0]);
                acc[3 * i + 1] = mf(g[2 * i + 1], b0, acc[3 * i + 1]);
                acc[3 * i + 2] = mf(g[2 * i], b0, acc[3 * i + 2]);
            }
            __builtin_amdgcn_sched_barrier(0);
            rd += 8192u; if (rd >= 6 * 8192u) rd = 0;
        }
    }
    if (MODE == 1) wait_vm<0>();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += acc[i][0];
    if (MODE == 2) s += __builtin_bit_cast(float, st[0][0].x) + __builtin_bit_cast(float, st[1][1].y);
    out[blockIdx.x * 256 + tid] = s;
}


// Software-pipelined variant (what the real kernels do: ring reads run one phase ahead, under the MFMAs).
// FEED 0: none   1: 2 x dwordx4 LDS-DMA right after the barrier   2: the same two pieces after MFMA 1 and MFMA 7
//      3: 8 x buffer_load_dword ... lds (256 B each), one after each of the first 8 MFMAs
//      4: 2 x dwordx4 after the barrier, ring reads NOT interleaved (all 8 after the MFMAs' issue)
// DEP 0: every MFMA of a phase has its own accumulator   1: the real kernels' pattern: a pair of accumulators takes three
// MFMAs each, alternating (n0 n1 n0 n1 n0 n1): every dependent MFMA issues one MFMA after its producer
// 2: four accumulators in rotation (n0 n1 n2 n3 n0 ...): dependent MFMAs three apart
__device__ unsigned long long g_clk[2];
template <int FEED, int DEP>
__global__ __launch_bounds__(256, 1) void kp(const u32x4* __restrict__ W, unsigned bytes, float* out, int phases) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) u32x4 ring[];       // 6 slots x 8 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 6 * 512; i += 256) ring[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    f32x16 acc[12];
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const u32x4 b0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    i32x4 rs;
    const unsigned long long a = (unsigned long long)W;
    rs.x = (int)(unsigned)a; rs.y = (int)(unsigned)(a >> 32); rs.z = (int)bytes; rs.w = 0x00020000;
    const unsigned voff = lane * 16u + wave * 2048u;
    const unsigned voff4 = lane * 4u + wave * 2048u;
    unsigned soff = (blockIdx.x * 8192u * 13u) % (bytes - 65536u);
    unsigned wr = wave * 2048u, rd = 0;
    u32x4 g[2][8];
    {
        const char* src = (const char*)ring + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) g[0][i] = *(const u32x4*)(src + i * 1024);
        rd = 8192u;
    }
    auto dma16 = [&](unsigned off) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff + off), "s"(rs), "s"(soff), "s"(wr + off) : "memory");
    };
    auto dma4 = [&](unsigned off) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                     "buffer_load_dword %1, %2, %3 offen lds\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff4 + off), "s"(rs), "s"(soff), "s"(wr + off) : "memory");
    };
    for (int ph = 0; ph < phases; ph += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            if (FEED) { if (FEED == 3) wait_vm<24>(); else wait_vm<6>(); }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (FEED == 1 || FEED == 4) { dma16(0); dma16(1024); }
            const char* src = (const char*)ring + rd + lane * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int ai = DEP == 0 ? i : DEP == 1 ? 2 * (i / 6) + (i & 1) : (i & 3);
                acc[ai] = mf(g[par][(2 * (i / 3) + (i % 3 == 1)) & 7], b0, acc[ai]);
                if (FEED == 2 && i == 1) dma16(0);
                if (FEED == 2 && i == 7) dma16(1024);
                if (FEED == 3 && i < 8) dma4(256u * i);
                if (FEED != 4 && i >= 2 && i < 10) g[par ^ 1][i - 2] = *(const u32x4*)(src + (i - 2) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (FEED == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) g[par ^ 1][i] = *(const u32x4*)(src + i * 1024);
            }
            if (FEED) {
                soff += 8192u; if (soff > bytes - 65536u) soff = 0;
                wr += 8192u; if (wr >= 6 * 8192u) wr -= 6 * 8192u;
            }
            rd += 8192u; if (rd >= 6 * 8192u) rd = 0;
        }
    }
    if (FEED) wait_vm<0>();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) { g_clk[0] = __builtin_readcyclecounter() - c0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int FEED, int DEP = 0> void runp(const char* name, const u32x4* W, unsigned bytes, float* out) {
    const int phases = 20000, blocks = 256;
    hipFuncSetAttribute((const void*)kp<FEED, DEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kp<FEED, DEP>), dim3(blocks), dim3(256), 6 * 8192, 0, W, bytes, out, phases);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long hc[2];
    hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), 16);
    const double mhz = 100.0 * hc[0] / hc[1];
    const double cyc = ms * 1e-3 * mhz * 1e6 / phases;          // true shader cycles (in-kernel s_memtime / s_memrealtime)
    printf("%-64s %8.3f ms @ %4.0f MHz  %6.0f cycles per phase  MFMA %.0f %%\n", name, ms, mhz, cyc, 384.0 / cyc * 100);
}

template <int MODE> void run(const char* name, const u32x4* W, unsigned bytes, float* out) {
    const int phases = 20000, blocks = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 6 * 8192, 0, W, bytes, out, phases);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double cyc = ms * 1e-3 * 2.4e9 / phases;
    printf("%-52s %8.3f ms  %6.0f cycles per phase (12 MFMAs = 384)  MFMA %.0f %%\n", name, ms, cyc, 384.0 / cyc * 100);
}

int main() {
    const unsigned bytes = 16u << 20;
    u32x4* W; float* out;
    hipMalloc(&W, bytes); hipMalloc(&out, 256 * 256 * 4);
    std::vector<unsigned> h(bytes / 4, 0x3c003c00u);
    hipMemcpy(W, h.data(), bytes, hipMemcpyHostToDevice);
    run<0>("no feed (barrier + 8 ds_read_b128 + 12 MFMA)", W, bytes, out);
    run<1>("+ LDS-DMA: 2 x buffer_load_dwordx4 ... lds", W, bytes, out);
    run<2>("+ registers: 2 x buffer_load_dwordx4, 2 x ds_write_b128", W, bytes, out);
    runp<0>("pipelined reads, no feed", W, bytes, out);
    runp<1>("pipelined reads + 2 x dwordx4 DMA after the barrier", W, bytes, out);
    runp<2>("pipelined reads + 2 x dwordx4 DMA after MFMA 1 / 7", W, bytes, out);
    runp<3>("pipelined reads + 8 x dword DMA, one per MFMA", W, bytes, out);
    runp<4>("2 x dwordx4 DMA after the barrier, reads after the MFMAs", W, bytes, out);
    runp<0, 1>("no feed, accumulators n0 n1 n0 n1 n0 n1 (the real pattern)", W, bytes, out);
    runp<2, 1>("DMA after MFMA 1 / 7, accumulators n0 n1 n0 n1 n0 n1", W, bytes, out);
    runp<0, 2>("no feed, four accumulators in rotation", W, bytes, out);
    runp<2, 2>("DMA after MFMA 1 / 7, four accumulators in rotation", W, bytes, out);
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
