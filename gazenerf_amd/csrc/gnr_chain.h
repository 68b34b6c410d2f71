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
// If `dump_base` is non-null the layer also writes its INPUT registers (the previous layer's output)
// to HBM in the CCM layout, one or two 4-byte stores per weight row, so that the activation dump of
// the training forward / the dY dump of the backward trickles out under the MFMAs instead of hitting
// HBM as a chip-wide burst at every layer boundary.  dump_base = dst + chunk*32*C + 4h*32 + j.
template <int NT_IN, int NT_OUT, bool DUMP = false>
__device__ __forceinline__ void mm_h(const f32x16 (&hin)[NT_H], f32x16 (&acc)[NT_H],
                                     const f32x4* __restrict__ P, int lane, float* __restrict__ dump_base = nullptr) {
    constexpr int NROW = NT_IN * 4 * NT_OUT;        // (k-group, n-tile) rows, k-group outer
    constexpr int NREG = NT_IN * 16;                // input registers to dump
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
                if (DUMP) {
#pragma unroll
                    for (int q = (i * NREG) / NROW; q < ((i + 1) * NREG) / NROW; ++q) {
                        const int dt = q >> 4, dr = q & 15;
                        dump_base[(32 * dt + (dr & 3) + 8 * (dr >> 2)) * CHUNK] = hin[dt][dr];
                    }
                }
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
        for (int r = 0; r < 16; ++r) base[(32 * t + (r & 3) + 8 * (r >> 2)) * CHUNK] = acc[t][r];
}

__device__ __forceinline__ float* dump_ptr(float* dst, int C, long chunk, int j, int h) {
    return dst + chunk * (CHUNK * (long)C) + (4 * h) * CHUNK + j;
}

// ReLU masks as bits: lane l keeps the signs of its own registers, tile pair (2w, 2w+1) -> word w,
// bit 16*(t&1) + r.  Stored lane-major ([word][64 lanes]) so a wave moves 256 contiguous bytes per word.
constexpr int RELU_WORDS = NT_H / 2;     // 6
__host__ __device__ constexpr size_t relu_bits_offset(int layer, long n_chunks, long chunk) {
    return ((size_t)layer * n_chunks + chunk) * RELU_WORDS * 64;
}

template <int NT>
__device__ __forceinline__ void store_relu_bits(const f32x16 (&acc)[NT_H], unsigned* __restrict__ dst, int lane) {
#pragma unroll
    for (int w = 0; w < NT / 2; ++w) {
        unsigned bits = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) bits |= (acc[2 * w + q][r] > 0.0f ? 1u : 0u) << (16 * q + r);
        dst[w * 64 + lane] = bits;
    }
}

template <int NT>
__device__ __forceinline__ void load_relu_bits(unsigned (&mk)[RELU_WORDS], const unsigned* __restrict__ src, int lane) {
#pragma unroll
    for (int w = 0; w < NT / 2; ++w) mk[w] = src[w * 64 + lane];
}

// zero the gradient where the forward activation was clamped
template <int NT>
__device__ __forceinline__ void apply_relu_bits(f32x16 (&acc)[NT_H], const unsigned (&mk)[RELU_WORDS]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[t][r] = ((mk[t >> 1] >> (16 * (t & 1) + r)) & 1u) ? acc[t][r] : 0.0f;
}

}  // namespace gnr
