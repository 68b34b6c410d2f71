// gnr_chain.h -- register-chained dense layers on v_mfma_f32_32x32x2_f32 (shared by fwd and bwd).
//
// A wavefront owns 32 samples (MFMA columns).  Activations live in the C/D register layout, which
// is the B-operand layout of the next layer in the packer's k-order, so a layer is:
//   acc[nt] (+)= sum over input tiles t, registers r:  A-fragment(weights) x hin[t][r]
// with the weights streamed from the pre-packed, L2-resident fragment array.
#pragma once
#include "gnr_device.h"

namespace gnr {

// Weight-stream prefetch depth: RING float4 rows (1 KiB each per wave) stay in flight while the
// MFMAs of earlier rows issue; one row feeds 4 MFMAs = 256 cycles, so RING=8 covers ~2000 cycles of
// L2/MALL latency.  sched_barrier(0) after every row keeps hipcc from re-serialising the stream
// (left alone it re-uses one register quad and waits vmcnt(0) per row).
constexpr int RING = 8;

// ---- one dense layer: acc[nt] += sum over the channels held in hin[0..NT_IN) --------------------
template <int NT_IN, int NT_OUT>
__device__ __forceinline__ void mm_h(const f32x16 (&hin)[NT_H], f32x16 (&acc)[NT_H],
                                     const f32x4* __restrict__ P, int lane) {
    constexpr int NROW = NT_IN * 4 * NT_OUT;        // (k-group, n-tile) rows, k-group outer
    const f32x4* Pl = P + lane;
    f32x4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = Pl[i * 64];
#pragma unroll
    for (int t = 0; t < NT_IN; ++t) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
#pragma unroll
            for (int nt = 0; nt < NT_OUT; ++nt) {
                const int i = (t * 4 + rq) * NT_OUT + nt;
                acc[nt] = mfma32(ring[i % RING].x, hin[t][4 * rq + 0], acc[nt]);
                acc[nt] = mfma32(ring[i % RING].y, hin[t][4 * rq + 1], acc[nt]);
                acc[nt] = mfma32(ring[i % RING].z, hin[t][4 * rq + 2], acc[nt]);
                acc[nt] = mfma32(ring[i % RING].w, hin[t][4 * rq + 3], acc[nt]);
                if (i + RING < NROW) ring[i % RING] = Pl[(i + RING) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// ---- the 64-channel positional-encoding slab (32 k-steps), read back from LDS -------------------
template <int NT_OUT>
__device__ __forceinline__ void mm_enc(const float* enc_col, f32x16 (&acc)[NT_H],
                                       const f32x4* __restrict__ P, int lane) {
    constexpr int NROW = (ENC_STEPS / 4) * NT_OUT;
    const f32x4* Pl = P + lane;
    f32x4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = Pl[i * 64];
#pragma unroll
    for (int sg = 0; sg < ENC_STEPS / 4; ++sg) {
        const float e0 = enc_col[(4 * sg + 0) * 256];
        const float e1 = enc_col[(4 * sg + 1) * 256];
        const float e2 = enc_col[(4 * sg + 2) * 256];
        const float e3 = enc_col[(4 * sg + 3) * 256];
#pragma unroll
        for (int nt = 0; nt < NT_OUT; ++nt) {
            const int i = sg * NT_OUT + nt;
            acc[nt] = mfma32(ring[i % RING].x, e0, acc[nt]);
            acc[nt] = mfma32(ring[i % RING].y, e1, acc[nt]);
            acc[nt] = mfma32(ring[i % RING].z, e2, acc[nt]);
            acc[nt] = mfma32(ring[i % RING].w, e3, acc[nt]);
            if (i + RING < NROW) ring[i % RING] = Pl[(i + RING) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NT>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NT_H], const float* __restrict__ bias, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const f32x4 b4 = *(const f32x4*)(bias + 32 * t + 8 * rq + 4 * h);
            acc[t][4 * rq + 0] = b4.x;
            acc[t][4 * rq + 1] = b4.y;
            acc[t][4 * rq + 2] = b4.z;
            acc[t][4 * rq + 3] = b4.w;
        }
    }
}

template <int NT>
__device__ __forceinline__ void relu(f32x16 (&acc)[NT_H]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = fmaxf(acc[t][r], 0.0f);
}

// row-major [M][C] dump of a register tile set (16 B per lane; used by the training forward)
template <int NT>
__device__ __forceinline__ void dump(const f32x16 (&acc)[NT_H], float* __restrict__ dst, int C,
                                     long row, int h) {
    float* base = dst + row * C + 4 * h;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 v = {acc[t][4 * rq], acc[t][4 * rq + 1], acc[t][4 * rq + 2], acc[t][4 * rq + 3]};
            *(f32x4*)(base + 32 * t + 8 * rq) = v;
        }
}

}  // namespace gnr
