// gnr_chain.h -- pieces shared by the register-chained kernels: dump stores, the counted vmcnt wait, the 32-sample
// chunk's CCM dump / compositing / ReLU bit words (bf16x3 kernels, gnr_chain3.h; the fp32 kernels use the 16-sample
// forms of gnr_chain16.h).  Round 2's fp32 chain on v_mfma_f32_32x32x2_f32 (one wave per SIMD) lived here until round 4.
#pragma once
#include "gnr_device.h"

namespace gnr {

// Training dumps are written once and read by a later kernel: nontemporal stores keep them from
// displacing the weight stream in the XCD's L2 (measured on the bf16x3 training forward: -18 %).
template <class T>
__device__ __forceinline__ void dump_store(T* p, T v) {
#ifdef GNR_NODUMP_TIMING            // timing experiment only (tools/ablate_fwd3.sh): results are incomplete
    (void)p; (void)v;
#elif defined(GNR_TEMPORAL_DUMP_TIMING)
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}


// One explicit wait per batch of a weight stream: loads return in order, so "at most N younger VMEM operations still
// outstanding" (N = the batch just requested + the dump stores issued since the consumed batch was requested)
// guarantees the whole consumed batch has landed; hipcc then drops its own per-row waits.
// gfx9 s_waitcnt encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14].
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// Chunk-channel-major ("CCM") dump of a register tile set: element (chunk c, channel n, sample j)
// lives at c*32*C + n*32 + j.  Register r of tile t is one channel per lane-half, so every store /
// load instruction moves two full 128-byte segments (32 samples x 4 B per half-wave), and the
// wgrad kernel can read a lane's 16 consecutive samples of one channel as 4 float4.
template <int NT>
__device__ __forceinline__ void dump(const f32x16 (&acc)[NT_H], float* __restrict__ dst, int C,
                                     long chunk, int j, int h) {
    float* base = dst + chunk * (CHUNK * (long)C) + (4 * h) * CHUNK + j;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dump_store(base + (32 * t + (r & 3) + 8 * (r >> 2)) * CHUNK, acc[t][r]);
}

// Chunk-local alpha compositing (CalcRayColor, utils/model_utils.py:498-534) of the 32 samples a wave
// owns: alpha_i = 1 - exp(-relu(sigma_raw_i) delta_i); T_i = exclusive product of (1 - alpha + 1e-10)
// WITHIN the chunk; w_i = alpha_i T_i.  Writes the chunk's weighted feature sum (288 floats), its
// total transmittance, sum w and sum w z; combine_kernel applies the cross-chunk prefix products.
__device__ __forceinline__ void composite_chunk(const f32x16 (&feat)[NT_H], float sigma_raw, float delta, float z0,
                                                const StreamWs& ws, long chunk, long row, int lane, bool keep_wl) {
    const int j = lane & 31, h = lane >> 5;
    const float sigma = fmaxf(sigma_raw, 0.0f);
    const float alpha = 1.0f - expf(-sigma * delta);
    const float x = (1.0f - alpha) + 1e-10f;
    float incl = x;                                  // inclusive prefix product over the 32 samples
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const float o = __shfl_up(incl, d, 32);
        if (j >= d) incl *= o;
    }
    float excl = __shfl_up(incl, 1, 32);
    if (j == 0) excl = 1.0f;
    const float wl = alpha * excl;
    const float ptot = __shfl(incl, 31, 32);
    const float accw = half_sum32(wl);
    const float dsum = half_sum32(wl * z0);
    if (lane == 0) *(f32x4*)(ws.part_sc + chunk * 4) = f32x4{ptot, accw, dsum, 0.0f};
    if (keep_wl && h == 0) ws.wl[row] = wl;
    float* pf = ws.part_feat + chunk * FEAT_PAD + 4 * h;
#pragma unroll
    for (int t = 0; t < NT_F; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 v;
            v.x = half_sum32(wl * feat[t][4 * rq + 0]);
            v.y = half_sum32(wl * feat[t][4 * rq + 1]);
            v.z = half_sum32(wl * feat[t][4 * rq + 2]);
            v.w = half_sum32(wl * feat[t][4 * rq + 3]);
            if (j == 0) *(f32x4*)(pf + 32 * t + 8 * rq) = v;
        }
}

// ReLU masks as bits: lane l keeps the signs of its own registers, tile pair (2w, 2w+1) -> word w,
// bit 16*(t&1) + r.  Stored lane-major ([word][64 lanes]) so a wave moves 256 contiguous bytes per word.
constexpr int RELU_WORDS = NT_H / 2;     // 6
__host__ __device__ constexpr size_t relu_bits_offset(int layer, long n_chunks, long chunk) {
    return ((size_t)layer * n_chunks + chunk) * RELU_WORDS * 64;
}

template <int NT>
__device__ __forceinline__ void load_relu_bits(unsigned (&mk)[RELU_WORDS], const unsigned* __restrict__ src, int lane) {
#pragma unroll
    for (int w = 0; w < NT / 2; ++w) mk[w] = src[w * 64 + lane];
}

}  // namespace gnr
