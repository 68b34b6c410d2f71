// Microbenchmark 2: which part of "MFMA A-operand streamed from memory" costs ~10%?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

// MODE 0: ring loads consumed by the MFMAs (baseline pattern), RING template
// MODE 1: ring loads consumed by ONE v_add each (MFMA operands are loop-invariant registers)
// MODE 2: whole k-group (12 rows) loaded ahead, one wait, 48 MFMAs back-to-back (double-buffered groups)
// MODE 3: rows come from LDS (ds_read_b128), written once; ring of 8
template <int MODE, int RING>
__global__ __launch_bounds__(256, 1) void k(const f32x4* __restrict__ W, float* out, int iters, unsigned wmask) {
    __shared__ f32x4 lds[64 * 64];            // 64 rows x 64 lanes
    const int lane = threadIdx.x & 63;
    f32x16 acc[12];
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float b[16];
    for (int r = 0; r < 16; ++r) b[r] = 1.0f + lane * 1e-3f + r;
    const f32x4* p = W + lane;
    if (MODE == 3) { for (int i = threadIdx.x; i < 64 * 64; i += 256) lds[i] = W[i]; __syncthreads(); }
    const f32x4 cst = {1.f, 2.f, 3.f, 4.f};
    float side = 0.f;
    unsigned row = 0;
    if (MODE == 2) {
        f32x4 g0[12], g1[12];
        for (int i = 0; i < 12; ++i) g0[i] = p[((row + i) & wmask) * 64];
        for (int it = 0; it < iters * 2; ++it) {          // two k-groups per iteration
#pragma unroll
            for (int i = 0; i < 12; ++i) g1[i] = p[((row + 12 + i) & wmask) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 12; nt += 2) {
                acc[nt] = MF(g0[nt].x, b[0], acc[nt]); acc[nt + 1] = MF(g0[nt + 1].x, b[0], acc[nt + 1]);
                acc[nt] = MF(g0[nt].y, b[1], acc[nt]); acc[nt + 1] = MF(g0[nt + 1].y, b[1], acc[nt + 1]);
                acc[nt] = MF(g0[nt].z, b[2], acc[nt]); acc[nt + 1] = MF(g0[nt + 1].z, b[2], acc[nt + 1]);
                acc[nt] = MF(g0[nt].w, b[3], acc[nt]); acc[nt + 1] = MF(g0[nt + 1].w, b[3], acc[nt + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 12; ++i) g0[i] = p[((row + 24 + i) & wmask) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 12; nt += 2) {
                acc[nt] = MF(g1[nt].x, b[4], acc[nt]); acc[nt + 1] = MF(g1[nt + 1].x, b[4], acc[nt + 1]);
                acc[nt] = MF(g1[nt].y, b[5], acc[nt]); acc[nt + 1] = MF(g1[nt + 1].y, b[5], acc[nt + 1]);
                acc[nt] = MF(g1[nt].z, b[6], acc[nt]); acc[nt + 1] = MF(g1[nt + 1].z, b[6], acc[nt + 1]);
                acc[nt] = MF(g1[nt].w, b[7], acc[nt]); acc[nt + 1] = MF(g1[nt + 1].w, b[7], acc[nt + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            row += 24;
        }
    } else if (MODE == 4 || MODE == 5) {
        f32x4 g0[6], g1[6];
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
        const unsigned lo = lane * 16;
        auto ld = [&](unsigned r) -> f32x4 {
            if (MODE == 5) {
                auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, lo, (r & wmask) * 1024u, 0);
                return __builtin_bit_cast(f32x4, v);
            }
            return p[(r & wmask) * 64];
        };
        for (int i = 0; i < 6; ++i) g0[i] = ld(row + i);
        for (int it = 0; it < iters * 4; ++it) {          // two half-groups per iteration (12 rows)
#pragma unroll
            for (int i = 0; i < 6; ++i) g1[i] = ld(row + 6 + i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 6; nt += 2) {
                acc[nt] = MF(g0[nt].x, b[0], acc[nt]); acc[nt + 1] = MF(g0[nt + 1].x, b[0], acc[nt + 1]);
                acc[nt] = MF(g0[nt].y, b[1], acc[nt]); acc[nt + 1] = MF(g0[nt + 1].y, b[1], acc[nt + 1]);
                acc[nt] = MF(g0[nt].z, b[2], acc[nt]); acc[nt + 1] = MF(g0[nt + 1].z, b[2], acc[nt + 1]);
                acc[nt] = MF(g0[nt].w, b[3], acc[nt]); acc[nt + 1] = MF(g0[nt + 1].w, b[3], acc[nt + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 6; ++i) g0[i] = ld(row + 12 + i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 6; nt += 2) {
                acc[6 + nt] = MF(g1[nt].x, b[0], acc[6 + nt]); acc[7 + nt] = MF(g1[nt + 1].x, b[0], acc[7 + nt]);
                acc[6 + nt] = MF(g1[nt].y, b[1], acc[6 + nt]); acc[7 + nt] = MF(g1[nt + 1].y, b[1], acc[7 + nt]);
                acc[6 + nt] = MF(g1[nt].z, b[2], acc[6 + nt]); acc[7 + nt] = MF(g1[nt + 1].z, b[2], acc[7 + nt]);
                acc[6 + nt] = MF(g1[nt].w, b[3], acc[6 + nt]); acc[7 + nt] = MF(g1[nt + 1].w, b[3], acc[7 + nt]);
            }
            __builtin_amdgcn_sched_barrier(0);
            row += 12;
        }
    } else {
        f32x4 ring[RING];
        for (int i = 0; i < RING; ++i) ring[i] = MODE == 3 ? lds[(i & 63) * 64 + lane] : p[i * 64];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int nt = 0; nt < 12; nt += 2) {
                    const int i = g * 12 + nt, i1 = i + 1;
                    f32x4 a0 = ring[i % RING], a1 = ring[i1 % RING];
                    if (MODE == 1) { side += a0.x + a1.x; a0 = cst; a1 = cst; }
                    acc[nt] = MF(a0.x, b[4 * g + 0], acc[nt]); acc[nt + 1] = MF(a1.x, b[4 * g + 0], acc[nt + 1]);
                    acc[nt] = MF(a0.y, b[4 * g + 1], acc[nt]); acc[nt + 1] = MF(a1.y, b[4 * g + 1], acc[nt + 1]);
                    acc[nt] = MF(a0.z, b[4 * g + 2], acc[nt]); acc[nt + 1] = MF(a1.z, b[4 * g + 2], acc[nt + 1]);
                    acc[nt] = MF(a0.w, b[4 * g + 3], acc[nt]); acc[nt + 1] = MF(a1.w, b[4 * g + 3], acc[nt + 1]);
                    if (MODE == 3) {
                        ring[i % RING] = lds[((row + i + RING) & 63) * 64 + lane];
                        ring[i1 % RING] = lds[((row + i1 + RING) & 63) * 64 + lane];
                    } else {
                        ring[i % RING] = p[((row + i + RING) & wmask) * 64];
                        ring[i1 % RING] = p[((row + i1 + RING) & wmask) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            row += 48;
        }
    }
    float s = side;
    for (int i = 0; i < 12; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int RING>
void run(const char* name, const f32x4* W, float* out, unsigned wrows) {
    const int iters = 1000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, RING>), dim3(blocks), dim3(256), 0, 0, W, out, 10, wrows - 1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, RING>), dim3(blocks), dim3(256), 0, 0, W, out, iters, wrows - 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 4 * iters * 192.0;
    printf("%-58s %8.3f ms  %6.1f TF (%.1f%%)\n", name, ms, mfma * 4096 / ms / 1e9, mfma * 4096 / ms / 1e9 / 157.3 * 100);
}

int main() {
    const long maxrows = 8192;
    f32x4* W; float* out;
    hipMalloc(&W, maxrows * 64 * sizeof(f32x4) + (1 << 20)); hipMalloc(&out, 256 * 8 * 256 * 4);
    std::vector<float> h(maxrows * 256, 0.5f);
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 8>("ring 8, loads -> MFMA A operand (8 MB stream)", W, out, 8192);
    run<0, 4>("ring 4", W, out, 8192);
    run<0, 16>("ring 16", W, out, 8192);
    run<1, 8>("ring 8, loads consumed by VALU only, MFMA on constants", W, out, 8192);
    run<2, 8>("k-group double buffer: 12 loads, 1 wait, 48 MFMA b2b", W, out, 8192);
    run<3, 8>("ring 8 fed from LDS (ds_read_b128)", W, out, 8192);
    run<4, 8>("half-group batches: 6 loads, 1 wait, 24 MFMA b2b", W, out, 8192);
    run<5, 8>("same with buffer_load (SGPR descriptor, 32-bit voffset)", W, out, 8192);
    return 0;
}
