// gnr_chain.h -- register-chained dense layers on v_mfma_f32_32x32x2_f32 (shared by fwd and bwd).
//
// A wavefront owns 32 samples (MFMA columns).  Activations live in the C/D register layout, which
// is the B-operand layout of the next layer in the packer's k-order, so a layer is:
//   acc[nt] (+)= sum over input tiles t, registers r:  A-fragment(weights) x hin[t][r]
// with the weights streamed from the pre-packed, L2-resident fragment array.
#pragma once
#include "gnr_device.h"

namespace gnr {

// Weight stream.  The packer lays every layer's A-fragment rows (1 KiB per wave: 64 lanes x float4)
// in EXECUTION order, layer after layer and stream after stream, so a wave reads one linear sequence
// of rows for its whole life.  RING rows stay in flight: row i is consumed by 4 MFMAs (256 cycles)
// and its slot is immediately refilled with row i+RING, also across layer boundaries -- a layer
// never starts with an exposed load latency.  Every layer has a multiple of RING rows, so the ring
// phase is 0 at every layer start and all register indices stay compile-time constants.
// sched_barrier(0) after every row keeps hipcc from re-serialising the stream (left alone it
// re-uses one register quad and waits vmcnt(0) per row).
constexpr int RING = 8;

struct WStream {
    const f32x4* p;          // next row of this lane (base + lane)
    f32x4 ring[RING];
};

__device__ __forceinline__ void wstream_init(WStream& w, const float* packed, int lane) {
    w.p = (const f32x4*)packed + lane;
#pragma unroll
    for (int i = 0; i < RING; ++i) w.ring[i] = w.p[i * 64];
}

// ---- one dense layer: acc[nt] (+)= sum over the channels held in hin[0..NT_IN) -------------------
// ZERO: the first MFMA of every output tile takes an inline-zero C operand (no accumulator init).
// DUMP: the layer also writes its INPUT registers (the previous layer's output) to HBM in the CCM
// layout, one or two 4-byte stores per weight row, so that the activation dump of the training
// forward / the dY dump of the backward trickles out under the MFMAs instead of hitting HBM as a
// chip-wide burst at every layer boundary.  dump_base = dst + chunk*32*C + 4h*32 + j.
// EPI: per-output-tile epilogue (bias + activation / ReLU mask ...).  It is software-pipelined into
// the LAST k-group: right after the final MFMAs of tile nt are issued, the epilogue of tile nt-1
// (whose accumulator completed one row earlier) runs on the VALU underneath them, so the layer
// boundary exposes one tile's epilogue instead of twelve with the matrix pipe idle.
struct NoEpi {
    __device__ __forceinline__ void operator()(int) const {}
};

template <int NT_IN, int NT_OUT, bool ZERO, bool DUMP = false, class Epi = NoEpi>
__device__ __forceinline__ void mm_h(const f32x16 (&hin)[NT_H], f32x16 (&acc)[NT_H], WStream& w,
                                     float* __restrict__ dump_base = nullptr, Epi epi = Epi()) {
    constexpr int NROW = NT_IN * 4 * NT_OUT;        // (k-group, n-tile) rows, k-group outer
    constexpr int NREG = NT_IN * 16;                // input registers to dump
    static_assert(NROW % RING == 0, "layer rows must keep the ring phase");
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NT_IN; ++t) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
#pragma unroll
            for (int nt = 0; nt < NT_OUT; nt += 2) {
                const int i = (t * 4 + rq) * NT_OUT + nt;
                const bool z = ZERO && t == 0 && rq == 0;
                if (nt + 1 < NT_OUT) {
                    // two output tiles interleaved: consecutive MFMAs never share an accumulator
                    const int i1 = i + 1;
                    acc[nt] = mfma32(w.ring[i % RING].x, hin[t][4 * rq + 0], z ? zero : acc[nt]);
                    acc[nt + 1] = mfma32(w.ring[i1 % RING].x, hin[t][4 * rq + 0], z ? zero : acc[nt + 1]);
                    acc[nt] = mfma32(w.ring[i % RING].y, hin[t][4 * rq + 1], acc[nt]);
                    acc[nt + 1] = mfma32(w.ring[i1 % RING].y, hin[t][4 * rq + 1], acc[nt + 1]);
                    acc[nt] = mfma32(w.ring[i % RING].z, hin[t][4 * rq + 2], acc[nt]);
                    acc[nt + 1] = mfma32(w.ring[i1 % RING].z, hin[t][4 * rq + 2], acc[nt + 1]);
                    acc[nt] = mfma32(w.ring[i % RING].w, hin[t][4 * rq + 3], acc[nt]);
                    acc[nt + 1] = mfma32(w.ring[i1 % RING].w, hin[t][4 * rq + 3], acc[nt + 1]);
                    w.ring[i % RING] = w.p[(i + RING) * 64];
                    w.ring[i1 % RING] = w.p[(i1 + RING) * 64];
                } else {
                    acc[nt] = mfma32(w.ring[i % RING].x, hin[t][4 * rq + 0], z ? zero : acc[nt]);
                    acc[nt] = mfma32(w.ring[i % RING].y, hin[t][4 * rq + 1], acc[nt]);
                    acc[nt] = mfma32(w.ring[i % RING].z, hin[t][4 * rq + 2], acc[nt]);
                    acc[nt] = mfma32(w.ring[i % RING].w, hin[t][4 * rq + 3], acc[nt]);
                    w.ring[i % RING] = w.p[(i + RING) * 64];
                }
                if (DUMP) {
#pragma unroll
                    for (int ii = i; ii < i + 2 && ii < (t * 4 + rq + 1) * NT_OUT; ++ii)
#pragma unroll
                        for (int q = (ii * NREG) / NROW; q < ((ii + 1) * NREG) / NROW; ++q) {
                            const int dt = q >> 4, dr = q & 15;
                            dump_base[(32 * dt + (dr & 3) + 8 * (dr >> 2)) * CHUNK] = hin[dt][dr];
                        }
                }
                // epilogue of the PREVIOUS pair (complete since the last iteration) under this pair's MFMAs
                if (t == NT_IN - 1 && rq == 3 && nt >= 2) {
                    epi(nt - 2);
                    epi(nt - 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    constexpr int NT_LAST = (NT_OUT - 1) & ~1;
    epi(NT_LAST);
    if (NT_LAST + 1 < NT_OUT) epi(NT_LAST + 1);
    w.p += NROW * 64;
}

// ---- the 64-channel positional-encoding slab (32 k-steps), read back from LDS; always zero-starts
template <int NT_OUT>
__device__ __forceinline__ void mm_enc(const float* enc_col, f32x16 (&acc)[NT_H], WStream& w) {
    constexpr int NROW = (ENC_STEPS / 4) * NT_OUT;
    static_assert(NROW % RING == 0, "layer rows must keep the ring phase");
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int sg = 0; sg < ENC_STEPS / 4; ++sg) {
        const float e0 = enc_col[(4 * sg + 0) * 256];
        const float e1 = enc_col[(4 * sg + 1) * 256];
        const float e2 = enc_col[(4 * sg + 2) * 256];
        const float e3 = enc_col[(4 * sg + 3) * 256];
#pragma unroll
        for (int nt = 0; nt < NT_OUT; ++nt) {
            const int i = sg * NT_OUT + nt;
            acc[nt] = mfma32(w.ring[i % RING].x, e0, sg == 0 ? zero : acc[nt]);
            acc[nt] = mfma32(w.ring[i % RING].y, e1, acc[nt]);
            acc[nt] = mfma32(w.ring[i % RING].z, e2, acc[nt]);
            acc[nt] = mfma32(w.ring[i % RING].w, e3, acc[nt]);
            w.ring[i % RING] = w.p[(i + RING) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    w.p += NROW * 64;
}

// epilogue of ONE output tile: acc += bias (from the wave's LDS bias table), optional ReLU;
// returns the 16 sign bits of the result (for the training forward's ReLU bit words)
template <bool RELU>
__device__ __forceinline__ unsigned bias_act_tile(f32x16& acc, const float* bias_lds_tile, int h) {
    unsigned bits = 0;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        const f32x4 b4 = *(const f32x4*)(bias_lds_tile + 8 * rq + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = acc[4 * rq + e] + b4[e];
            acc[4 * rq + e] = RELU ? (v > 0.0f ? v : 0.0f) : v;
            bits |= (v > 0.0f ? 1u : 0u) << (4 * rq + e);
        }
    }
    return bits;
}

template <int NT, bool RELU>
__device__ __forceinline__ void bias_act(f32x16 (&acc)[NT_H], const float* bias_lds, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t) bias_act_tile<RELU>(acc[t], bias_lds + 32 * t, h);
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

// zero the gradient of one tile where the forward activation was clamped
__device__ __forceinline__ void apply_relu_bits_tile(f32x16& acc, unsigned word, int t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = ((word >> (16 * (t & 1) + r)) & 1u) ? acc[r] : 0.0f;
}

}  // namespace gnr
