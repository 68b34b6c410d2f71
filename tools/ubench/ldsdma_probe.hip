// Probe of the LDS-DMA addressing rules the bf16x3 weight ring relies on (gfx950):
//   buffer_load_dwordx4 ... offen lds : LDS dest = M0 + inst_offset + lane*16 ?  M0 above 64 KiB ?
//   out-of-range buffer reads land zeros in LDS ?
// hipcc --offload-arch=gfx950 -O3 -o ldsdma_probe ldsdma_probe.hip && ./ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe(const unsigned* src, unsigned n_bytes, unsigned* out, unsigned lds_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 40960; i += 256) smem[i] = 0xdeadbeefu;      // 160 KiB
    __syncthreads();
    i32x4 rs;
    const unsigned long long a = (unsigned long long)src;
    rs.x = (int)(unsigned)a; rs.y = (int)(unsigned)(a >> 32); rs.z = (int)n_bytes; rs.w = 0x00020000;
    const unsigned voff = lane * 16u + wave * 2048u;
    const unsigned soff = 8192u * blockIdx.x;
    const unsigned m0v = (unsigned)(size_t)smem + lds_off + wave * 2048u;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(voff), "s"(rs), "s"(soff), "s"(m0v) : "memory");
    __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0) expcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    for (int i = tid; i < 40960; i += 256) out[(size_t)blockIdx.x * 40960 + i] = smem[i];
}

int main() {
    const unsigned n_words = 2 * 2048 + 512;       // 2 full batches + a quarter: the rest must read as zero
    std::vector<unsigned> h(n_words);
    for (unsigned i = 0; i < n_words; ++i) h[i] = 0x10000000u + i;
    unsigned *d, *o;
    hipMalloc(&d, 3 * 8192); hipMalloc(&o, 3 * 40960 * 4);
    hipMemcpy(d, h.data(), n_words * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    int bad_total = 0;
    for (unsigned lds_off : {0u, 49152u, 65536u + 4096u, 155648u}) {
        hipLaunchKernelGGL(probe, dim3(3), dim3(256), 163840, 0, d, n_words * 4, o, lds_off);
        std::vector<unsigned> r(3 * 40960);
        hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        for (int b = 0; b < 3; ++b)
            for (int i = 0; i < 40960; ++i) {
                unsigned exp = 0xdeadbeefu;
                const int w0 = (int)lds_off / 4;
                if (i >= w0 && i < w0 + 2048) {
                    const unsigned g = b * 2048 + (i - w0);
                    exp = g < n_words ? h[g] : 0u;
                }
                if (r[b * 40960 + i] != exp) { if (first < 0) first = b * 40960 + i; ++bad; }
            }
        printf("lds_off %6u: %s (%d mismatches, first at %d: got %08x)\n", lds_off, bad ? "MISMATCH" : "ok", bad, first,
               first >= 0 ? r[first] : 0);
        bad_total += bad;
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(e));
    return bad_total != 0;
}
