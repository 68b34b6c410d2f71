// gnr_chain16.h -- the fp32 dense chain on v_mfma_f32_16x16x4_f32 with TWO waves per SIMD (round 3).
//
// Round 1/2 ran one wave per SIMD on 32-sample tiles (v_mfma_f32_32x32x2_f32; 192 + 192 activation registers).
// With one wave per SIMD nothing overlaps the wave's own non-MFMA instructions: every buffer load, LDS read or VALU
// instruction between two MFMAs stalls the matrix pipe (tools/ubench/mfma_stream2.hip: 88-92 % of peak for the bare
// weight-streaming loop, 0.84-0.88 for the real kernels).  tools/ubench/mfma_2w.hip measured the alternative:
//
//     one wave / SIMD, 32x32x2, 6-row buffer-load batches (the round-2 kernels' loop)      88.6-89.3 %
//     two waves / SIMD, 16x16x4, 4-row buffer-load batches                                 97.3 %
//     ... 8-row / 12-row batches                                                            96.1 / 97.1 %
//     ... fed by ds_read_b128 from LDS / by an LDS-DMA ring in a 512-thread workgroup       91.3 / 91.8 %
//     every VALU instruction among the MFMAs still costs ~4 matrix-pipe cycles (13 before)
//
// So: a wave owns 16 samples (the 16 columns of a 16x16x4 tile; same FLOPs per cycle as 32x32x2), its activation
// sets are 96 + 96 registers, two waves share a SIMD (256 registers each) and keep each other's MFMA pipe busy while
// one of them issues loads / epilogue VALU.  Waves stay independent (no barriers, per-wave buffer loads): the
// LDS-fed variants measured slower than plain buffer loads.
//
// Register chain, as before: the C/D layout of a 16x16 tile (lane l: sample l&15, register e of tile t: channel
// 16 t + 4 (l>>4) + e) IS the B-operand layout of the next layer when k-step e of input tile t takes register e
// (lane group g = l>>4 supplies k = g): a layer's output feeds the next one with no data movement.
#pragma once
#include "gnr_chain.h"

namespace gnr {

constexpr int SUB = 16;                    // samples per wavefront
constexpr int NT16_H = H / 16;             // 24 tiles of 16 channels
constexpr int NT16_H2 = H2 / 16;           // 12
constexpr int NT16_F = FEAT_PAD / 16;      // 18
constexpr int NT16_E = ENC_PAD / 16;       // 4
constexpr int ENC16 = 16;                  // encoding values per lane (64 slots over 4 lane groups)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    // v_mfma_f32_16x16x4_f32: lane l holds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15];
    // D register e: row i = 4 (l>>4) + e, column j = l&15.
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// channel held by register e of tile t in lane group g (C/D layout == next layer's k-order)
__host__ __device__ constexpr int d16_channel(int t, int e, int g) { return 16 * t + 4 * g + e; }

// ---- positional encoding over four lane groups ------------------------------------------------------------
// The HBM formats (encoding dump, weight-gradient column map) keep round 1's 64 "slots" (slot = 2 step + h of
// enc_channel, gnr_internal.h); only the assignment of slots to lanes is new.  The 30 (frequency, axis) pairs
// p = 15 h + 3 fl + a are dealt to the groups as 7 + 7 + 8 + 8, sin and cos of a pair to the same lane; groups 0 / 1
// also carry the raw coordinates: g0 {x, z}, g1 {y, pad}.  Lane value idx (0..15) is k-step idx&3 of k-group idx>>2.
__host__ __device__ constexpr int enc16_pair_base(int g) { return g < 2 ? 7 * g : 14 + 8 * (g - 2); }
__host__ __device__ constexpr int enc16_slot(int idx, int g) {
    if (g < 2 && idx < 2) return idx == 0 ? g : 2 + g;             // x: 0, y: 1, z: 2, pad: 3
    const int k = g < 2 ? idx - 2 : idx;
    const int p = enc16_pair_base(g) + (k >> 1), sc = k & 1;
    const int h = p / 15, fl = (p % 15) / 3, a = p % 3;
    return 2 * (2 + 6 * fl + 3 * sc + a) + h;
}
// reference channel (utils/model_utils.py:272-280 order) of lane value idx of group g, or -1 for the pad
__host__ __device__ constexpr int enc16_channel(int idx, int g) {
    const int s = enc16_slot(idx, g);
    const int step = s >> 1, h = s & 1;
    if (step == 0) return h;
    if (step == 1) return h == 0 ? 2 : -1;
    const int i2 = step - 2, fl = i2 / 6, q = i2 % 6;
    return 3 + 6 * (5 * h + fl) + q;
}

// Embedder.forward (utils/model_utils.py:272-280) for lane group g: the 16 values of enc16_slot(., g).
// Arguments and sincosf are exactly round 1's (scale = 2^f exactly; arguments reach ~1.7e3 rad).
__device__ __forceinline__ void encode_point16(float px, float py, float pz, int g, float (&e)[ENC16]) {
    const int base = g < 2 ? 7 * g : 14 + 8 * (g - 2);
    float s[8], c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int p = min(base + q, 29);               // (groups 0 / 1 have 7 pairs: their q = 7 is discarded)
        const int h = p >= 15 ? 1 : 0, pp = p - 15 * h, fl = pp / 3, a = pp - 3 * fl;
        const float scale = (float)(1 << fl) * (h ? 32.0f : 1.0f);
        const float x = a == 0 ? px : (a == 1 ? py : pz);
        sincosf(x * scale, &s[q], &c[q]);
    }
    const bool lowg = g < 2;
    const float raw0 = g == 0 ? px : py, raw1 = g == 0 ? pz : 0.0f;
    e[0] = lowg ? raw0 : s[0];
    e[1] = lowg ? raw1 : c[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        e[2 * q] = lowg ? s[q - 1] : s[q];
        e[2 * q + 1] = lowg ? c[q - 1] : c[q];
    }
}

// old-format slot of lane value idx (runtime group): used for the encoding dump / its read-back
__device__ __forceinline__ int enc16_slot_rt(int idx, int g) {
    if (g < 2 && idx < 2) return idx == 0 ? g : 2 + g;
    const int k = g < 2 ? idx - 2 : idx;
    const int p = (g < 2 ? 7 * g : 14 + 8 * (g - 2)) + (k >> 1), sc = k & 1;
    const int h = p >= 15 ? 1 : 0, pp = p - 15 * h, fl = pp / 3, a = pp - 3 * fl;
    return 2 * (2 + 6 * fl + 3 * sc + a) + h;
}

// ---- weight stream: rows of 1 KiB (64 lanes x float4 = the A fragments of 4 MFMAs), batches of 4 ---------------
constexpr int WB16 = 4;
// dump stores per burst (1 = one store every NROW / NREG rows).  Round 3 (half-row stores): 1 was best.  Round 4, whole-line
// stores (S16): bursts of 4-12 measure 0.5-0.9 % faster on the training forward and 0.2 % on the dgrad chain
// (profiles/r4_dump_burst.txt: the store queue takes a burst of full lines without stalling the issue, and fewer, longer
// interruptions of the MFMA stream cost less than many short ones); 8 is the default.
constexpr int DUMP_BURST = 8;
struct WStream16 {
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff;               // lane * 16 (constant)
    unsigned soff;               // byte offset of the batch most recently requested: wave-uniform, so the per-batch
                                 // increment is a scalar instruction (as a VGPR offset it was one v_add per 16 MFMAs)
    f32x4 g[2][WB16];
};

template <int Q>
__device__ __forceinline__ f32x4 wrow16(const WStream16& w) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.voff + (unsigned)Q * 1024u, (int)w.soff, 0));
}
__device__ __forceinline__ void wbatch16(WStream16& w, f32x4 (&g)[WB16]) {
    g[0] = wrow16<0>(w); g[1] = wrow16<1>(w); g[2] = wrow16<2>(w); g[3] = wrow16<3>(w);
}
// (plain C: the fully unrolled chain turns the running offset into per-batch scalar constants; hipcc keeps ~300 of
// them live and spills those through v_writelane / v_readlane -- ~600 VALU instructions per kernel against the 1150
// v_add of a VGPR offset.  Making the value opaque with an asm, even an empty one, costs thousands of VGPR spills.)
__device__ __forceinline__ void wadvance16(WStream16& w) { w.soff += WB16 * 1024u; }
__device__ __forceinline__ void wstream16_init(WStream16& w, const float* packed, int lane) {
    w.rs = __builtin_amdgcn_make_buffer_rsrc((void*)packed, 0, 0x7ffffff0, 0x00020000);
    w.voff = (unsigned)lane * 16u;
    w.soff = 0;
    wbatch16(w, w.g[0]);
}

// ---- de-phasing the two waves of a SIMD ----------------------------------------------------------------------------
// Workgroups of equal length that start together stay in step: the two waves of a SIMD would run their non-MFMA phases
// (ray set-up, encoding, bias-table fill, compositing) at the same time, with the matrix pipe idle.  The workgroups of
// the FIRST round that land in an odd wave slot sleep ~25 us once; every later workgroup inherits the offset of the slot
// it takes over.  (HW_ID bits 3:0 = wave slot within the SIMD.)
// GNR_SOFTSTART=N (timing experiment, round 6; results unchanged): the first-round workgroup with linear index i additionally
// sleeps i N / 64 periods of ~3.4 us before it starts -- the chip's matrix load then RAMPS over 512 N / 64 periods (N = 16: ~0.44 ms)
// instead of stepping up within a few microseconds.  Question behind it (profiles/r4_forward_clock_experiments.txt): is the clock
// the training forward loses after a backward (2.16-2.25 GHz for a whole 21 ms launch at cfg4's size) a reaction to the STEP?
#ifndef GNR_SOFTSTART
#define GNR_SOFTSTART 0
#endif
__device__ __forceinline__ void dephase_first_round(unsigned linear_block) {
    if (GNR_SOFTSTART > 0 && linear_block < 512u) {
        const int n = (int)(linear_block * (unsigned)GNR_SOFTSTART / 64u);
#pragma unroll 1
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    }
    if (linear_block < 512u) {                          // 2 workgroups x 256 CUs
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if (hw & 1u) {
#pragma unroll 1
            for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127);      // 8 x 127 x 64 cycles ~ 27 us at 2.4 GHz
        }
    }
}

// ---- dump destination: buffer stores with a wave-uniform descriptor ---------------------------------------------
// Layer dumps (activations h0..h7, y0, y1 and every dY) use the "S16" layout (round 4): a tensor of C channels is a
// sequence of 16-sample sub-chunks of 16 C floats; inside a sub-chunk channel n = 16 t + 4 g + e sits in the 64-byte row
// rho(n) = 16 t + 4 e + g (e and g swapped).  The wave owns the 16 samples j of one sub-chunk and, in register e of tile
// t, channel 16 t + 4 g + e in lane group g: ONE store instruction (fixed t, e) then writes rows 16 t + 4 e + 0..3 =
// 256 CONTIGUOUS bytes, two whole 128-byte lines.  Rounds 1-3 used chunk-channel-major rows of 32 samples (c*32*C + n*32 +
// j): a 16-sample wave could only write 64-byte HALF rows, four per instruction, the other half arriving from the partner
// wave at another time.  Measured (tools/ubench/store_pattern.hip, profiles/r4_store_pattern.txt): that pattern drains at
// 3.3 TB/s against 5.4 for whole lines and is COUNTED 1.10-1.17 x by WRITE_SIZE (which is exact, 1.001 x, on whole-line
// patterns) -- half-line writes that miss their partner in L2 go out as two masked writes of the line; in the real kernels
// 1.32 x (44.9 GB for 33.9 GB of dumps per launch).  The consumers (gnr_wgrad.hip) fetch 16-byte pieces = 4 consecutive
// samples of one channel, which are contiguous in either layout: only their lane / piece offsets change.
// With the descriptor based at the sub-chunk, every dump of the wave uses the SAME lane offset (16 g + j) 4 bytes and a
// compile-time constant (t 1024 + e 256 bytes) split over the 12-bit immediate and one scalar: no 64-bit address
// arithmetic on the VALU (global_store needed a v_add_co / v_addc pair every few stores -- each VALU instruction among
// the MFMAs costs ~4 matrix-pipe cycles), and one VGPR instead of a pointer pair per destination.
// (s16_row(n): gnr_internal.h)
struct Dump16 {
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff;
};
__device__ __forceinline__ unsigned dump_lane_off16(int j, int g) { return (unsigned)(16 * g + j) * 4u; }
__device__ __forceinline__ Dump16 dump_dst16(float* dst, int C, long sub, unsigned lane_off) {
    Dump16 d;
    d.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + sub * (SUB * (long)C)), 0, 0x7ffffff0, 0x00020000);
    d.voff = lane_off;
    return d;
}
// byte offset (compile-time) -> nontemporal dword store
template <class T>
__device__ __forceinline__ void dump_store16_bytes(const Dump16& d, unsigned byte, T v) {
    static_assert(sizeof(T) == 4, "dword dumps");
#ifdef GNR_NODUMP_TIMING            // timing experiment only: results are incomplete
    (void)d; (void)byte; (void)v;
#else
    // scalar part: multiples of 1 KiB (the tile); immediate: the register within the tile (e 256 bytes; four values).  With
    // the low 12 bits as the immediate hipcc CSEs "lane offset + immediate" over all layers and keeps the sums in VGPRs
    // (every store with its own address register, none with the offset field): registers the chain does not have; a
    // 2 KiB split (eight immediates) already spills 2 / 15 VGPRs in the forward / dgrad kernels.
    const unsigned lo = byte & 1023u, hi = byte & ~1023u;
#ifdef GNR_TEMPORAL_DUMP_TIMING
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), d.rs, d.voff + lo, (int)hi, 0);
#else
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), d.rs, d.voff + lo, (int)hi, 2);   // aux 2 = nt
#endif
#endif
}
// register e of tile t (S16 layout)
template <class T>
__device__ __forceinline__ void dump_store16(const Dump16& d, int t, int e, T v) {
    dump_store16_bytes(d, (unsigned)(t * 1024 + e * 256), v);
}

// The composited features (act_feat) keep the chunk-channel-major layout of rounds 1-3 -- element (chunk c, channel n,
// sample jj) at c*32*C + n*32 + jj: their one reader, comp_bwd_kernel, streams a sample per lane at 0.97 of the measured
// HBM rate on whole rows, and they are 7 % of the dumped bytes.
__device__ __forceinline__ Dump16 dump_dst16_ccm(float* dst, int C, long sub, int j, int g) {
    Dump16 d;
    d.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + (sub >> 1) * (CHUNK * (long)C)), 0, 0x7ffffff0, 0x00020000);
    d.voff = (unsigned)((4 * g) * CHUNK + 16 * (int)(sub & 1) + j) * 4u;
    return d;
}

// ---- one dense layer: acc[nt] (+)= sum over the channels held in hin[0..NT_IN) --------------------------------
// Rows are (k-group = input tile, n-tile), k-group outer; the four rows of a batch are interleaved so that
// consecutive MFMAs never share an accumulator.  INIT: the first MFMA of every output tile takes its C operand from
// init(nt) (the bias tile, or zeros) instead of the accumulator.  DUMP: the layer also writes its INPUT registers
// (the previous layer's output) to HBM in the CCM layout, spread over its rows.  epi(nt) runs for every finished
// output tile inside the tail of the loop, one batch behind the MFMAs that completed it.
struct ZeroInit16 {
    __device__ __forceinline__ f32x4 operator()(int) const { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
};

template <int NT_IN, int NT_OUT, bool INIT, bool DUMP, class Init, class Epi>
__device__ __forceinline__ void mm16_h(const f32x4 (&hin)[NT16_H], f32x4 (&acc)[NT16_H], WStream16& w,
                                       const Dump16& dump_dst, Init init, Epi epi) {
    constexpr int NROW = NT_IN * NT_OUT;
    constexpr int NREG = NT_IN * 4;
    constexpr int NB = NROW / WB16;
    constexpr int LAST0 = NROW - NT_OUT;
    static_assert(NROW % (2 * WB16) == 0 && NB % 2 == 0 && NT_OUT >= WB16, "layer rows must keep the batch parity");
#pragma clang loop unroll(full)
    for (int kbo = 0; kbo < NB / 2; ++kbo)
#pragma clang loop unroll(full)
    for (int kbi = 0; kbi < 2; ++kbi) {
        const int kb = kbo * 2 + kbi;
        wadvance16(w);
        wbatch16(w, w.g[(kb + 1) & 1]);
        if (!DUMP) wait_vm<WB16>();
        __builtin_amdgcn_sched_barrier(0);
        const int i0 = kb * WB16;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int u = 0; u < WB16; ++u) {
                const int i = i0 + u, kg = i / NT_OUT, nt = i % NT_OUT;
                const f32x4 a = w.g[kb & 1][u];
                acc[nt] = mfma16(a[e], hin[kg][e], (INIT && kg == 0 && e == 0) ? init(nt) : acc[nt]);
            }
        if (DUMP) {
            // stores due by the end of this batch, issued in bursts of DUMP_BURST (the inputs are complete from the start)
            constexpr int S = DUMP_BURST;
            const int t0 = (i0 * NREG) / NROW, t1 = ((i0 + WB16) * NREG) / NROW;
            const int q0 = kb == 0 ? 0 : ((t0 + S - 1) / S) * S, q1 = ((t1 + S - 1) / S) * S;
#pragma unroll
            for (int q = q0; q < (q1 < NREG ? q1 : NREG); ++q)
                dump_store16(dump_dst, q >> 2, q & 3, hin[q >> 2][q & 3]);
        }
        // epilogue of the tiles completed by the previous batch
#pragma unroll
        for (int u = 0; u < WB16; ++u)
            if (i0 - WB16 + u >= LAST0) epi(i0 - WB16 + u - LAST0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < WB16; ++u) epi(NT_OUT - WB16 + u);
}

// ---- the 64-slot positional-encoding slab (4 k-groups), B operand read back from LDS ----------------------------
template <int NT_OUT, bool STORES_IN_FLIGHT, class Init>
__device__ __forceinline__ void mm16_enc(const float* enc_col, f32x4 (&acc)[NT16_H], WStream16& w, Init init) {
    constexpr int NROW = NT16_E * NT_OUT;
    constexpr int NB = NROW / WB16;
    static_assert(NROW % (2 * WB16) == 0 && NT_OUT % WB16 == 0, "layer rows must keep the batch parity");
    float ev[4] = {0, 0, 0, 0};
#pragma clang loop unroll(full)
    for (int kb = 0; kb < NB; ++kb) {
        wadvance16(w);
        wbatch16(w, w.g[(kb + 1) & 1]);
        const int i0 = kb * WB16, kg = i0 / NT_OUT;
        if (i0 % NT_OUT == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) ev[c] = enc_col[(4 * kg + c) * 256];
        }
        if (!STORES_IN_FLIGHT) wait_vm<WB16>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int u = 0; u < WB16; ++u) {
                const int nt = (i0 + u) % NT_OUT;
                const f32x4 a = w.g[kb & 1][u];
                acc[nt] = mfma16(a[e], ev[e], (kg == 0 && e == 0) ? init(nt) : acc[nt]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// d: dump_dst16_ccm (chunk-channel-major rows of 32 samples: channel 16 t + 4 g + e at row offset (16 t + e) 128 bytes
// from the lane's)
template <int NT>
__device__ __forceinline__ void dump16_ccm(const f32x4 (&acc)[NT16_H], const Dump16& d) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) dump_store16_bytes(d, (unsigned)((16 * t + e) * CHUNK * 4), acc[t][e]);
}

// ReLU sign bits: lane l keeps the signs of its own registers, tiles 8 w .. 8 w + 7 -> word w, bit 31 - (4 (t & 7) + e)
// (the forward pushes them in with shift-or).
// [layer][sub-chunk][3 words][64 lanes]: the same bytes per 32 samples as the round-1 layout.
constexpr int RELU16_WORDS = NT16_H / 8;     // 3
__host__ __device__ constexpr size_t relu16_offset(int layer, long n_sub, long sub) {
    return ((size_t)layer * n_sub + sub) * RELU16_WORDS * 64;
}

// sum over the 16 lanes of a DPP row (all lanes receive it): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// Sub-chunk-local alpha compositing (CalcRayColor, utils/model_utils.py:498-534) of the 16 samples a wave owns
// (every lane group holds the same 16 scalars): writes the weighted feature sum (288 floats), the sub-chunk's
// transmittance, sum w and sum w z; combine_kernel applies the cross-sub-chunk prefix products.
__device__ __forceinline__ void composite_sub(const f32x4 (&feat)[NT16_H], float sigma_raw, float delta, float z0,
                                              const StreamWs& ws, long sub, long row, int lane, bool keep_wl) {
    const int j = lane & 15, g = lane >> 4;
    const float sigma = fmaxf(sigma_raw, 0.0f);
    const float alpha = 1.0f - expf(-sigma * delta);
    const float x = (1.0f - alpha) + 1e-10f;
    float incl = x;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const float o = __shfl_up(incl, d, 16);
        if (j >= d) incl *= o;
    }
    float excl = __shfl_up(incl, 1, 16);
    if (j == 0) excl = 1.0f;
    const float wl = alpha * excl;
    const float ptot = __shfl(incl, 15, 16);
    const float accw = row_sum16(wl);
    const float dsum = row_sum16(wl * z0);
    if (lane == 0) *(f32x4*)(ws.part_sc + sub * 4) = f32x4{ptot, accw, dsum, 0.0f};
    if (keep_wl && g == 0) ws.wl[row] = wl;
    float* pf = ws.part_feat + sub * FEAT_PAD + 4 * g;
#pragma unroll
    for (int t = 0; t < NT16_F; ++t) {
        f32x4 v;
        v.x = row_sum16(wl * feat[t][0]);
        v.y = row_sum16(wl * feat[t][1]);
        v.z = row_sum16(wl * feat[t][2]);
        v.w = row_sum16(wl * feat[t][3]);
        if (j == 0) *(f32x4*)(pf + 16 * t) = v;
    }
}

}  // namespace gnr
