// v_permlane32_swap_b32 semantics on gfx950 (used by fwd3_kernel's encoding dumps to build 16-byte stores from
// the two lane halves): prints r[0] / r[1] of __builtin_amdgcn_permlane32_swap(x, y) at lanes 0, 1, 32, 33 for
// x = lane, y = 100 + lane.   hipcc --offload-arch=gfx950 -O3 permlane_probe.hip -o permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned x = threadIdx.x, y = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512);
    k<<<1, 64>>>(d);
    unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 32, 33}) printf("lane %2d: r0 = %3u  r1 = %3u\n", l, h[l], h[64 + l]);
    return 0;
}
