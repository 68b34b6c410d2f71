// tr_read_probe.hip -- what ds_read_b64_tr_b16 delivers on gfx950 (pins the lane/element mapping the bf16x3
// weight-gradient kernel relies on).  LDS is filled with 16-bit tags = element index; every lane reads 8 bytes at
// address lane*8 (so lane L holds elements 4L..4L+3 in a plain read) and prints what the transposing read returns.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out, int stride_bytes) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    // lane address: row r = (lane % 16) / 4 ... generic: base + lane*8 for mode 0; mode by stride: key row = lane/4 within 16, col quad = lane%4
    const int lane = threadIdx.x;
    unsigned addr;
    if (stride_bytes == 0) addr = lane * 8;
    else addr = ((lane & 15) >> 2) * stride_bytes + (lane & 3) * 8 + (lane >> 4) * 4 * stride_bytes;   // 16-lane group g: keys 4g..4g+3
    unsigned lo, hi;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + (unsigned)(size_t)lds) : "memory");
    lo = (unsigned)v; hi = (unsigned)(v >> 32);
    out[lane * 2] = lo; out[lane * 2 + 1] = hi;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 8);
    for (int stride : {0, 32, 64}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        unsigned h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("stride %d bytes: lane -> 4 x 16-bit tags (element indices)\n", stride);
        for (int l = 0; l < 64; ++l)
            printf("  lane %2d: %4u %4u %4u %4u%s", l, h[2*l] & 0xffff, h[2*l] >> 16, h[2*l+1] & 0xffff, h[2*l+1] >> 16, (l % 4 == 3) ? "\n" : " |");
    }
    return 0;
}
