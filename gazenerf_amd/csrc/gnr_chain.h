// gnr_chain.h -- register-chained dense layers on v_mfma_f32_32x32x2_f32 (shared by fwd and bwd).
//
// A wavefront owns 32 samples (MFMA columns).  Activations live in the C/D register layout, which
// is the B-operand layout of the next layer in the packer's k-order, so a layer is:
//   acc[nt] (+)= sum over input tiles t, registers r:  A-fragment(weights) x hin[t][r]
// with the weights streamed from the pre-packed, L2-resident fragment array.
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


// Weight stream.  The packer lays every layer's A-fragment rows (1 KiB per wave: 64 lanes x float4)
// in EXECUTION order, layer after layer and stream after stream, so a wave reads one linear sequence
// of rows for its whole life.  Rows are fetched in BATCHES of 6 (24 MFMAs = 1536 matrix-pipe cycles)
// into two alternating register sets: while batch k's MFMAs issue back-to-back, batch k+1 is in
// flight -- also across layer boundaries, so a layer never starts with an exposed load latency.
// Why batches: on gfx950 every non-MFMA instruction between two 64-cycle fp32 MFMAs costs ~15-25
// matrix-pipe cycles (tools/ubench/mfma_stream2.hip: a per-row refill + s_waitcnt tops out at 88 % of
// peak regardless of prefetch depth or cache residency; 6-row batches with one wait reach 90 %, on
// buffer loads -- SGPR descriptor, 32-bit offsets -- 92 %).  Every layer has a multiple of 12 rows, so
// the batch parity is 0 at every layer start and all register indices stay compile-time constants.
// sched_barrier(0) fences keep hipcc from re-serialising the stream (left alone it re-uses one
// register quad and waits vmcnt(0) per row).
constexpr int WB = 6;            // rows per batch
constexpr int RING = 2 * WB;     // rows of padding the packed stream needs past its end

struct WStream {
    __amdgpu_buffer_rsrc_t rs;   // whole packed stream (wave-uniform descriptor)
    unsigned voff;               // lane * 16 + byte offset of the batch most recently requested
    f32x4 g[2][WB];
};

// row q (0..5) of the batch at w.voff: rows 0-3 use the 12-bit immediate, rows 4-5 the one constant
// soffset 4096 (a per-row SGPR offset makes hipcc materialise and spill ~1000 scalars)
template <int Q>
__device__ __forceinline__ f32x4 wrow(const WStream& w) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
        w.rs, w.voff + (unsigned)(Q & 3) * 1024u, Q < 4 ? 0 : 4096, 0));
}

__device__ __forceinline__ void wbatch(WStream& w, f32x4 (&g)[WB]) {
    g[0] = wrow<0>(w); g[1] = wrow<1>(w); g[2] = wrow<2>(w);
    g[3] = wrow<3>(w); g[4] = wrow<4>(w); g[5] = wrow<5>(w);
}

__device__ __forceinline__ void wstream_init(WStream& w, const float* packed, int lane) {
    w.rs = __builtin_amdgcn_make_buffer_rsrc((void*)packed, 0, 0x7ffffff0, 0x00020000);
    w.voff = (unsigned)lane * 16u;
    wbatch(w, w.g[0]);
}

// One explicit wait per batch: loads return in order, so "at most N younger VMEM operations still
// outstanding" (N = the batch just requested + the dump stores issued since the consumed batch was
// requested) guarantees the whole consumed batch has landed; hipcc then drops its own per-row waits.
// gfx9 s_waitcnt encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14].
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// EPI: per-output-tile epilogue (bias + activation / ReLU mask ...), software-pipelined into the LAST
// k-group: after a pair of rows has been issued, the epilogue of the tiles finished by the PREVIOUS
// pair runs on the VALU underneath them.
struct NoEpi {
    __device__ __forceinline__ void operator()(int) const {}
};

// ---- one dense layer: acc[nt] (+)= sum over the channels held in hin[0..NT_IN) -------------------
// ZERO: the first MFMA of every output tile takes an inline-zero C operand (no accumulator init).
// DUMP: the layer also writes its INPUT registers (the previous layer's output) to HBM in the CCM
// layout, spread over its rows, so that the activation dump of the training forward / the dY dump of
// the backward trickles out under the MFMAs instead of hitting HBM as a chip-wide burst at every
// layer boundary.  dump_base = dst + chunk*32*C + 4h*32 + j.
template <int NT_IN, int NT_OUT, bool ZERO, bool DUMP = false, class Epi = NoEpi>
__device__ __forceinline__ void mm_h(const f32x16 (&hin)[NT_H], f32x16 (&acc)[NT_H], WStream& w,
                                     float* __restrict__ dump_base = nullptr, Epi epi = Epi()) {
    constexpr int NROW = NT_IN * 4 * NT_OUT;        // (k-group, n-tile) rows, k-group outer
    constexpr int NREG = NT_IN * 16;                // input registers to dump
    constexpr int NB = NROW / WB;
    constexpr int LAST0 = NROW - NT_OUT;            // first row of the last k-group
    static_assert(NROW % (2 * WB) == 0 && NB % 4 == 0, "layer rows must keep the batch parity");
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // two nested loops (hipcc does not fully unroll one long flat loop, and a rolled loop would index
    // the register arrays dynamically)
#pragma clang loop unroll(full)
    for (int kbo = 0; kbo < NB / 4; ++kbo)
#pragma clang loop unroll(full)
    for (int kbi = 0; kbi < 4; ++kbi) {
        const int kb = kbo * 4 + kbi;
        // next batch (runs on past the end of the layer into the next one)
        w.voff += WB * 1024u;
        wbatch(w, w.g[(kb + 1) & 1]);
        // inference: one wait for the whole consumed batch.  With dump stores in the stream (training
        // forward / backward) hipcc's own per-row counting is kept: an explicit count would also wait
        // for the slow HBM store acknowledgements.
        if (!DUMP) wait_vm<WB>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jp = 0; jp < WB; jp += 2) {
            const int i = kb * WB + jp, i1 = i + 1;
            const int sg = i / NT_OUT, nt = i % NT_OUT, sg1 = i1 / NT_OUT, nt1 = i1 % NT_OUT;
            const int t = sg >> 2, rq = sg & 3, t1 = sg1 >> 2, rq1 = sg1 & 3;
            const f32x4 a0 = w.g[kb & 1][jp], a1 = w.g[kb & 1][jp + 1];
            // two rows interleaved: consecutive MFMAs never share an accumulator
            acc[nt] = mfma32(a0.x, hin[t][4 * rq + 0], (ZERO && sg == 0) ? zero : acc[nt]);
            acc[nt1] = mfma32(a1.x, hin[t1][4 * rq1 + 0], (ZERO && sg1 == 0) ? zero : acc[nt1]);
            acc[nt] = mfma32(a0.y, hin[t][4 * rq + 1], acc[nt]);
            acc[nt1] = mfma32(a1.y, hin[t1][4 * rq1 + 1], acc[nt1]);
            acc[nt] = mfma32(a0.z, hin[t][4 * rq + 2], acc[nt]);
            acc[nt1] = mfma32(a1.z, hin[t1][4 * rq1 + 2], acc[nt1]);
            acc[nt] = mfma32(a0.w, hin[t][4 * rq + 3], acc[nt]);
            acc[nt1] = mfma32(a1.w, hin[t1][4 * rq1 + 3], acc[nt1]);
            if (DUMP) {
#pragma unroll
                for (int q = (i * NREG) / NROW; q < ((i + 2) * NREG) / NROW; ++q) {
                    const int dt = q >> 4, dr = q & 15;
                    dump_store(dump_base + (32 * dt + (dr & 3) + 8 * (dr >> 2)) * CHUNK, hin[dt][dr]);
                }
            }
            // epilogue of the tiles completed by the previous pair (rows i-2, i-1)
            if (i - 2 >= LAST0) epi(i - 2 - LAST0);
            if (i - 1 >= LAST0) epi(i - 1 - LAST0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (NT_OUT >= 2) epi(NT_OUT - 2);
    epi(NT_OUT - 1);
}

// ---- the 64-channel positional-encoding slab (32 k-steps), read back from LDS; always zero-starts
template <int NT_OUT, bool STORES_IN_FLIGHT>
__device__ __forceinline__ void mm_enc(const float* enc_col, f32x16 (&acc)[NT_H], WStream& w) {
    constexpr int NROW = (ENC_STEPS / 4) * NT_OUT;
    constexpr int NB = NROW / WB;
    static_assert(NROW % (2 * WB) == 0 && NT_OUT % 2 == 0 && NB % 4 == 0, "layer rows must keep the batch parity");
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float e[4] = {0, 0, 0, 0};
#pragma clang loop unroll(full)
    for (int kbo = 0; kbo < NB / 4; ++kbo)
#pragma clang loop unroll(full)
    for (int kbi = 0; kbi < 4; ++kbi) {
        const int kb = kbo * 4 + kbi;
        w.voff += WB * 1024u;
        wbatch(w, w.g[(kb + 1) & 1]);
        if ((kb * WB) % NT_OUT == 0) {              // a new k-group starts in this batch
            const int sg = (kb * WB) / NT_OUT;
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = enc_col[(4 * sg + c) * 256];
        }
        if (!STORES_IN_FLIGHT) wait_vm<WB>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jp = 0; jp < WB; jp += 2) {
            const int i = kb * WB + jp;
            const int sg = i / NT_OUT, nt = i % NT_OUT;      // NT_OUT even: the pair stays in one k-group
            const f32x4 a0 = w.g[kb & 1][jp], a1 = w.g[kb & 1][jp + 1];
            acc[nt] = mfma32(a0.x, e[0], sg == 0 ? zero : acc[nt]);
            acc[nt + 1] = mfma32(a1.x, e[0], sg == 0 ? zero : acc[nt + 1]);
            acc[nt] = mfma32(a0.y, e[1], acc[nt]);
            acc[nt + 1] = mfma32(a1.y, e[1], acc[nt + 1]);
            acc[nt] = mfma32(a0.z, e[2], acc[nt]);
            acc[nt + 1] = mfma32(a1.z, e[2], acc[nt + 1]);
            acc[nt] = mfma32(a0.w, e[3], acc[nt]);
            acc[nt + 1] = mfma32(a1.w, e[3], acc[nt + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
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
        for (int r = 0; r < 16; ++r) dump_store(base + (32 * t + (r & 3) + 8 * (r >> 2)) * CHUNK, acc[t][r]);
}

__device__ __forceinline__ float* dump_ptr(float* dst, int C, long chunk, int j, int h) {
    return dst + chunk * (CHUNK * (long)C) + (4 * h) * CHUNK + j;
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
__device__ __forceinline__ void store_relu_bits(const f32x16 (&acc)[NT_H], unsigned* __restrict__ dst, int lane) {
#pragma unroll
    for (int w = 0; w < NT / 2; ++w) {
        unsigned bits = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) bits |= (acc[2 * w + q][r] > 0.0f ? 1u : 0u) << (16 * q + r);
        dump_store(dst + w * 64 + lane, bits);
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
