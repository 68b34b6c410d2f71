// Microbenchmark 3 (round 3): does a SECOND wave per SIMD hide the issue cost of the weight feed?
//
// The fp32 chain kernels run one wave per SIMD (32 samples x 384 channels x 2 register sets = 384 VGPRs) and lose
// 12-16 % of the matrix pipe to the non-MFMA instructions between the MFMAs (mfma_stream2.hip).  Halving the samples
// per wave (16 columns: v_mfma_f32_16x16x4_f32, 8 passes, same FLOP rate) halves the register sets to 96 + 96 and
// lets two waves share a SIMD: while one issues loads / VALU the other's MFMA keeps the pipe busy.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_2w mfma_2w.hip && ./mfma_2w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define MF32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }

// ---- baseline: one wave per SIMD, 32x32x2, buffer loads in batches of 6 rows (= the shipped fwd_kernel pattern) ----
__global__ __launch_bounds__(256, 1) void k_base32(const f32x4* __restrict__ W, float* out, int iters, unsigned wmask) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[12];
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float b[16];
    for (int r = 0; r < 16; ++r) b[r] = 1.0f + lane * 1e-3f + r;
    f32x4 g0[6], g1[6];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
    const unsigned lo = lane * 16;
    unsigned row = 0;
    auto ld = [&](unsigned r) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, (r & wmask) * 1024u, 0));
    };
    for (int i = 0; i < 6; ++i) g0[i] = ld(row + i);
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
        for (int i = 0; i < 6; ++i) g1[i] = ld(row + 6 + i);
        wait_vm<6>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 6; nt += 2) {
            acc[nt] = MF32(g0[nt].x, b[0], acc[nt]); acc[nt + 1] = MF32(g0[nt + 1].x, b[0], acc[nt + 1]);
            acc[nt] = MF32(g0[nt].y, b[1], acc[nt]); acc[nt + 1] = MF32(g0[nt + 1].y, b[1], acc[nt + 1]);
            acc[nt] = MF32(g0[nt].z, b[2], acc[nt]); acc[nt + 1] = MF32(g0[nt + 1].z, b[2], acc[nt + 1]);
            acc[nt] = MF32(g0[nt].w, b[3], acc[nt]); acc[nt + 1] = MF32(g0[nt + 1].w, b[3], acc[nt + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) g0[i] = ld(row + 12 + i);
        wait_vm<6>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 6; nt += 2) {
            acc[6 + nt] = MF32(g1[nt].x, b[0], acc[6 + nt]); acc[7 + nt] = MF32(g1[nt + 1].x, b[0], acc[7 + nt]);
            acc[6 + nt] = MF32(g1[nt].y, b[1], acc[6 + nt]); acc[7 + nt] = MF32(g1[nt + 1].y, b[1], acc[7 + nt]);
            acc[6 + nt] = MF32(g1[nt].z, b[2], acc[6 + nt]); acc[7 + nt] = MF32(g1[nt + 1].z, b[2], acc[7 + nt]);
            acc[6 + nt] = MF32(g1[nt].w, b[3], acc[6 + nt]); acc[7 + nt] = MF32(g1[nt + 1].w, b[3], acc[7 + nt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        row += 12;
    }
    float s = 0.f;
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- two waves per SIMD, 16x16x4: 24 accumulator tiles (96 regs) + 24 input tiles (96 regs) ----
// FEED 0: registers only (no loads: the ceiling)   1: buffer loads, batches of RB rows   2: ds_read_b128 from a static
// LDS image   3: LDS ring filled by LDS-DMA (8 waves = one 512-thread workgroup, wave w fetches row w of each 8-row batch)
// RID: extra VALU instructions per row of 4 MFMAs (are riders free now?)
template <int FEED, int RB, int RID, int NTHR, int UNR = 1>
__global__ __launch_bounds__(NTHR, NTHR == 256 ? 2 : 1) void k_w2(const f32x4* __restrict__ W, float* out, int iters, unsigned wmask) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x4 acc[24], hin[24];
    for (int i = 0; i < 24; ++i) for (int r = 0; r < 4; ++r) { acc[i][r] = 0.f; hin[i][r] = 1.0f + lane * 1e-3f + r + i; }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
    const unsigned lo = lane * 16;
    unsigned row = (blockIdx.x * 97u) & wmask;
    float side = 0.f;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    if (FEED == 2) { for (int i = tid; i < 64 * 64; i += NTHR) lds[i] = W[i]; __syncthreads(); }
    f32x4 g[2][RB];
    auto ld = [&](unsigned r) -> f32x4 {
        if (FEED == 2) return lds[(r & 63) * 64 + lane];
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, (r & wmask) * 1024u, 0));
    };
    // LDS-DMA ring (FEED 3): 6 slots x 8 rows x 1 KiB
    i32x4 rsd;
    {
        const unsigned long long a = (unsigned long long)W;
        rsd.x = (int)(unsigned)a; rsd.y = (int)(unsigned)(a >> 32); rsd.z = 0x7ffffff0; rsd.w = 0x00020000;
    }
    constexpr int NSLOT = 6, DEPTH = 4;
    unsigned soff = ((blockIdx.x * 97u) & wmask) * 1024u;
    auto dma = [&](int slot) {
        const unsigned m0v = (unsigned)(size_t)lds + slot * 8192u + wave * 1024u;
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                     :: "v"(lo + wave * 1024u), "s"(rsd), "s"(soff), "s"(m0v) : "memory");
        soff += 8192u; if (soff > wmask * 1024u) soff = 0;
    };
    if (FEED == 3) {
#pragma unroll
        for (int k = 0; k <= DEPTH; ++k) dma(k);
        wait_vm<DEPTH>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < RB; ++i) g[0][i] = lds[i * 64 + lane];
    } else if (FEED != 0) {
        for (int i = 0; i < RB; ++i) g[0][i] = ld(row + i);
    } else {
        for (int i = 0; i < RB; ++i) { g[0][i] = f32x4{1.f, 2.f, 3.f, 4.f}; g[1][i] = f32x4{1.f, 2.f, 3.f, 4.f}; }
    }
    // one iteration = NSLOT batches (so that the ring slot is a compile-time constant)
    for (int it = 0; it < iters; it += UNR) {
#pragma clang loop unroll(full)
        for (int kbu = 0; kbu < UNR; ++kbu)
#pragma clang loop unroll(full)
        for (int kbl = 0; kbl < NSLOT; ++kbl) {
            const int kb = kbu * NSLOT + kbl;
            if (FEED == 1 || FEED == 2) {
#pragma unroll
                for (int i = 0; i < RB; ++i) g[(kb + 1) & 1][i] = ld(row + RB + i);
                if (FEED == 1) wait_vm<RB>();
            }
            if (FEED == 3) {
                // batch kb+1 visible after: own piece landed + barrier.  Then request batch kb+DEPTH+1 into the slot of kb-1.
                wait_vm<DEPTH - 1>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                dma((kb + DEPTH + 1) % NSLOT);
#pragma unroll
                for (int i = 0; i < RB; ++i) g[(kb + 1) & 1][i] = lds[((kb + 1) % NSLOT) * 512 + i * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RB; i += 4) {
                // four rows interleaved: consecutive MFMAs never share an accumulator
                const int n0 = (kb * RB + i) % 24, t = (kb * 5 + i) % 24;
                f32x4 a0 = g[kb & 1][i], a1 = g[kb & 1][i + 1], a2 = g[kb & 1][i + 2], a3 = g[kb & 1][i + 3];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[n0] = MF16(a0[e], hin[t][e], acc[n0]);
                    acc[(n0 + 1) % 24] = MF16(a1[e], hin[t][e], acc[(n0 + 1) % 24]);
                    acc[(n0 + 2) % 24] = MF16(a2[e], hin[t][e], acc[(n0 + 2) % 24]);
                    acc[(n0 + 3) % 24] = MF16(a3[e], hin[t][e], acc[(n0 + 3) % 24]);
                    if (RID < 8 && RID > e) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) side = __builtin_fmaf(side, 1.0001f, a0[e]);
                    }
                    if (RID >= 8 && RID - 8 > e) {      // independent riders: four accumulators
                        s4[0] = __builtin_fmaf(s4[0], 1.0001f, a0[e]); s4[1] = __builtin_fmaf(s4[1], 1.0001f, a1[e]);
                        s4[2] = __builtin_fmaf(s4[2], 1.0001f, a2[e]); s4[3] = __builtin_fmaf(s4[3], 1.0001f, a3[e]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            row += RB;
        }
    }
    if (FEED == 3) wait_vm<0>();
    float s = side + s4[0] + s4[1] + s4[2] + s4[3];
    for (int i = 0; i < 24; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * NTHR + tid] = s;
}

static double g_base = 0;
template <class F>
void timeit(const char* name, F launch, double mfma_flop) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(400);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = mfma_flop * 400 / ms / 1e9;
    printf("%-76s %8.3f ms  %6.1f TF (%.1f%%)\n", name, ms, tf, tf / 157.3 * 100);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("  !! %s\n", hipGetErrorString(e));
}

template <int FEED, int RB, int RID, int NTHR, int UNR = 1>
void run_w2(const char* name, const f32x4* W, float* out, unsigned wrows) {
    const int blocks = NTHR == 256 ? 256 * 2 * 4 : 256 * 4;          // 4 rounds of a full chip
    const size_t lds = FEED == 2 ? 64 * 1024 : FEED == 3 ? 48 * 1024 : 0;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_w2<FEED, RB, RID, NTHR, UNR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // per iteration per wave: 6 batches x RB rows x 4 MFMA x 2048 FLOP
    const double flop = (double)blocks * (NTHR / 64) * 6.0 * RB * 4 * 2048.0;
    timeit(name, [&](int it) { hipLaunchKernelGGL((k_w2<FEED, RB, RID, NTHR, UNR>), dim3(blocks), dim3(NTHR), lds, 0, W, out, it, wrows - 1); }, flop);
}

int main() {
    const long maxrows = 8192;
    f32x4* W; float* out;
    hipMalloc(&W, maxrows * 64 * sizeof(f32x4) + (1 << 20)); hipMalloc(&out, 256 * 8 * 512 * 4);
    std::vector<float> h(maxrows * 256, 0.5f);
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    {
        const int blocks = 256 * 8;
        const double flop = (double)blocks * 4 * 4.0 * 12 * 4 * 4096.0;   // iters*4 half-groups x 12 rows x 4 MFMA
        timeit("1 wave/SIMD 32x32x2, buffer loads, 6-row batches (shipped pattern)", [&](int it) { hipLaunchKernelGGL(k_base32, dim3(blocks), dim3(256), 0, 0, W, out, it, 8191u); }, flop);
    }
    run_w2<1, 4, 0, 256>("2 waves/SIMD 16x16x4, buffer loads, 4-row batches", W, out, 8192);
    run_w2<1, 4, 0, 256, 8>("  loop body unrolled to 768 MFMAs (~8 KB of code)", W, out, 8192);
    run_w2<1, 4, 0, 256, 40>("  loop body unrolled to 3840 MFMAs (~40 KB of code)", W, out, 8192);
    run_w2<1, 4, 0, 256, 80>("  loop body unrolled to 7680 MFMAs (~80 KB of code)", W, out, 8192);
    run_w2<1, 4, 0, 256, 200>("  loop body unrolled to 19200 MFMAs (~200 KB of code)", W, out, 8192);
    run_w2<1, 8, 12, 256>("  8-row batches + 16 INDEPENDENT VALU riders per 16 MFMAs", W, out, 8192);
    run_w2<1, 8, 10, 256>("  8-row batches + 8 INDEPENDENT VALU riders per 16 MFMAs", W, out, 8192);
    run_w2<0, 8, 12, 256>("  registers only + 16 INDEPENDENT VALU riders per 16 MFMAs", W, out, 8192);
    run_w2<0, 8, 0, 256>("2 waves/SIMD 16x16x4, registers only", W, out, 8192);
    run_w2<1, 8, 0, 256>("2 waves/SIMD 16x16x4, buffer loads, 8-row batches", W, out, 8192);
    run_w2<1, 12, 0, 256>("2 waves/SIMD 16x16x4, buffer loads, 12-row batches", W, out, 8192);
    run_w2<1, 8, 2, 256>("  + 8 VALU riders per 16 MFMAs", W, out, 8192);
    run_w2<1, 8, 4, 256>("  + 16 VALU riders per 16 MFMAs", W, out, 8192);
    run_w2<2, 8, 0, 256>("2 waves/SIMD 16x16x4, ds_read_b128 from static LDS", W, out, 8192);
    run_w2<3, 8, 0, 512>("2 waves/SIMD 16x16x4, LDS-DMA ring (512-thread workgroup), 8-row batches", W, out, 8192);
    run_w2<3, 8, 4, 512>("  + 16 VALU riders per 16 MFMAs", W, out, 8192);
    run_w2<1, 8, 0, 512>("512-thread workgroup, buffer loads, 8-row batches", W, out, 8192);
    return 0;
}
