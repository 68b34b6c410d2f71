// gnr_chain3.h -- the "bf16x3" building blocks shared by gnr_fwd3.hip and gnr_bwd3.hip (gfx950).
//
// Every fp32 operand x is split x = hi + lo (+ ~2^-17 |x|), hi = bf16(x), lo = bf16(x - hi), and
//     a * b  ~=  a_hi b_hi + a_lo b_hi + a_hi b_lo          (fp32 accumulate in the MFMA)
// i.e. three v_mfma_f32_32x32x16_bf16 (32 matrix-pipe cycles, 16 k each) replace eight
// v_mfma_f32_32x32x2_f32 (64 cycles each): 5.3x the fp32-MFMA rate at ~16 mantissa bits per operand.
#pragma once
#include "gnr_chain.h"

namespace gnr {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
    // A: lane l holds A[i = l&31][k = 8(l>>5) + 0..7]; B: B[k = 8(l>>5) + 0..7][j = l&31]; C/D as f32 32x32
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// round-to-nearest-even fp32 -> bf16 (finite inputs)
__host__ __device__ __forceinline__ unsigned bf16_rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__host__ __device__ __forceinline__ float bf16_to_f32(unsigned b) { return __builtin_bit_cast(float, b << 16); }

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// split a pair of fp32 values into packed {hi(a), hi(b)} and {lo(a), lo(b)}: v_cvt_pk_bf16_f32 (RNE),
// shift/mask back to fp32, one packed subtract, v_cvt_pk_bf16_f32 -- 5 VALU instructions per pair
__device__ __forceinline__ unsigned hi_pair(float a, float b) {                 // 1 VALU
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned lo_pair(float a, float b, unsigned hi) {    // 4 VALU
    const f32x2 v = {a, b};
    const f32x2 hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    // (round 5: two scalar v_sub_f32 instead of the packed subtract -- a packed f32 op beside MFMAs costs more than its issue slot --
    // were measured: 5 VALU per pair instead of 4, bwd3_chain_kernel 12.67 -> 14.11 ms, 904.5 -> 925.2 J per step: profiles/r5_x3_energy.txt)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, bf16x2));
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    hi = hi_pair(a, b);
    lo = lo_pair(a, b, hi);
}

// k-order of a K=16 bf16 step s = 2t + u over an activation tile held in the C/D layout: lane-half h
// supplies, as element q (0..7), register r = 8u + q of tile t.
__host__ __device__ inline int dlayout3_channel(int step16, int h, int q) {
    const int t = step16 >> 1, r = 8 * (step16 & 1) + q;
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// ---------------------------------------------------------------------------------------------
// weight ring: the workgroup's four waves consume the SAME rows, so the stream goes through LDS once per
// workgroup instead of four times through the vector L1.  (The first version, every wave loading every row into
// a 48-register ring, ran at the same speed -- what costs is issuing memory instructions, tools/ubench/feed_cost.hip
// -- but needed 107 spilled registers; the ring in LDS leaves 3, a quarter of the L1 traffic, and room for the
// training variants' extra state.)  LDS-DMA (buffer_load_dwordx4 ... lds) fills a ring of NSLOT batches of
// RB_ROWS rows; wave w fetches row w of each batch (2 KiB = 2 pieces).
//
// Per batch k ("phase"), every wave:
//   s_waitcnt vmcnt(2 (DEPTH-1))   its own share of batch k+1 has landed
//   s_barrier                      -> batch k+1 is complete and visible; every wave has issued (hence
//                                     fetched the operands of) all MFMAs of batch k-1
//   MFMAs of rows 0,1 of batch k, one ds_read of rows 2,3 behind each of the first four, the first piece of batch
//       k+DEPTH+1 (into the slot batch k-1 occupied, NSLOT = DEPTH + 2) behind the second;
//   MFMAs of rows 2,3, with the reads of rows 0,1 of batch k+1 and the second piece      (ring_layer: issue budget)
// The DMA is inline asm (hipcc would otherwise put a vmcnt(0) in front of every LDS read that might
// alias it); hipcc's own vmcnt bookkeeping stays correct because extra outstanding operations only make
// its counted waits stricter, and ours count only operations issued after the ones we wait for.
// Addressing rules (LDS dest = M0 + inst_offset + lane*16, M0 beyond 64 KiB, zero fill past
// num_records) are pinned by tools/ubench/ldsdma_probe.hip.
// ---------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Timing experiments only (results are wrong with any bit set): build with -DGNR_ABLATE=<bits>.
//   1 no LDS-DMA requests   2 no barriers   4 no activation conversion   8 no ring reads   16 no vmcnt waits
//   32 no activation dumps (training forward)   64 vmcnt waits ignore the dump stores (as in inference)
//   128 NOT a timing experiment -- the demonstration behind the bf16x3 gradient gate (tests/test_parity_gpu.py NOISE_GATE,
//       profiles/r5_grad_gate_dropped_term.txt): the W_lo x a_hi cross term is dropped from ONE layer, the 192 -> 384 one
//       (RGB_layer_1^T of bwd3_chain_kernel; the forward has no layer of that shape)
#ifndef GNR_ABLATE
#define GNR_ABLATE 0
#endif
constexpr int ABL = GNR_ABLATE;

constexpr int RB_ROWS = 4;                             // rows per ring batch (one per wave)
constexpr int NSLOT = 6;
constexpr int DEPTH = NSLOT - 2;                       // batches in flight beyond the one made visible
constexpr unsigned BATCH_BYTES = RB_ROWS * 2048u;      // 8 KiB
constexpr unsigned RING_BYTES = NSLOT * BATCH_BYTES;   // 48 KiB at LDS offset 0
static_assert(RB_ROWS == WAVES_PER_WG, "one row of each batch per wave");

// Every layer starts on ring slot 0 (its phase count is a multiple of NSLOT; the one that is not -- RGB2 with 27
// phases -- appends three empty phases, ring_layer<.., SKIP>), so after unrolling every LDS offset below is an
// immediate and the only ring arithmetic left is one scalar add of the stream offset per phase.  (Round 1 carried
// rd / wr / wrap tests in SGPRs: ~10 scalar instructions per 12 MFMAs of a wave that can issue one instruction per
// four cycles -- see the issue budget at ring_layer.)
struct WRing {
    i32x4 rs;                // buffer descriptor of the packed stream (num_records = exact bytes)
    unsigned voff;           // lane*16 + wave*2048
    unsigned soff;           // stream offset of the next batch to request
    unsigned m0v[NSLOT];     // LDS address (M0) of this wave's row in slot s
    const char* lane_base;   // ring + lane*16
    u32x4 g[2][2][2];        // [pair][row][hi/lo]: rows 0,1 and rows 2,3 of the current batch
};

// One 1 KiB piece (PIECE = 0 / 1) of the wave's row of the batch at stream offset soff -> ring slot `slot`.
// M0 is a reserved register: hipcc never keeps a value in it across instructions (it re-materialises M0 right before
// each of its own uses), so the asm overwrites it without saving (the generated code is checked for foreign M0 uses by
// tools/resource_usage.sh).  Measured (tools/ablate_fwd3.sh, tools/ubench/feed_cost.hip): a piece blocks the issuing
// wave for 40-90 cycles, more than the 32-cycle shadow of one MFMA.
template <int PIECE>
__device__ __forceinline__ void ring_issue_piece(const WRing& w, int slot, unsigned soff) {
    if (ABL & 1) return;
    if (PIECE == 0)
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                     :: "v"(w.voff), "s"(w.rs), "s"(soff), "s"(w.m0v[slot]) : "memory");
    else
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds"
                     :: "v"(w.voff), "s"(w.rs), "s"(soff), "s"(w.m0v[slot]) : "memory");
}

// quarter i (0..3) of a pair of rows: row i>>1, hi / lo half i&1
__device__ __forceinline__ void ring_read_quarter(const WRing& w, int slot, int pair, int i, u32x4 (&g)[2][2]) {
    if (ABL & 8) return;
    g[i >> 1][i & 1] = *(const u32x4*)(w.lane_base + slot * (int)BATCH_BYTES + pair * 4096 + i * 1024);
}

__device__ __forceinline__ void ring_init(WRing& w, const float* packed, unsigned stream_bytes, char* ring, int lane,
                                          unsigned wave) {
    const unsigned long long a = (unsigned long long)packed;
    w.rs.x = (int)(unsigned)a;
    w.rs.y = (int)(unsigned)(a >> 32);
    w.rs.z = (int)stream_bytes;
    w.rs.w = 0x00020000;
    w.voff = (unsigned)lane * 16u + wave * 2048u;
    w.soff = 0;
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) w.m0v[sl] = (unsigned)(size_t)ring + wave * 2048u + sl * BATCH_BYTES;
    w.lane_base = ring + lane * 16;
    if (ABL & 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) w.g[q >> 2][(q >> 1) & 1][q & 1] = u32x4{(unsigned)lane, 1u, 2u, 3u};
    }
#pragma unroll
    for (int k = 0; k <= DEPTH; ++k) {
        ring_issue_piece<0>(w, k, w.soff);
        ring_issue_piece<1>(w, k, w.soff);
        w.soff += BATCH_BYTES;
    }
}

// first rows into registers: call after ring_init, with no other barrier in between
// s_barrier is IntrNoMem for the compiler: it may move LDS loads across it.  The ring reads that follow a
// barrier must stay behind it, so every barrier here is followed by a compiler-only memory fence.
__device__ __forceinline__ void ring_barrier() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void ring_start(WRing& w) {
    wait_vm<2 * DEPTH>();
    ring_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) ring_read_quarter(w, 0, 0, i, w.g[0]);
}

// s_waitcnt vmcnt(n) for an n that is a constant after unrolling
__device__ __forceinline__ void wait_vm_n(int n) {
    switch (n < 30 ? n : 30) {
#define GNR_W(k) case k: wait_vm<k>(); break;
        GNR_W(0) GNR_W(1) GNR_W(2) GNR_W(3) GNR_W(4) GNR_W(5) GNR_W(6) GNR_W(7) GNR_W(8) GNR_W(9) GNR_W(10)
        GNR_W(11) GNR_W(12) GNR_W(13) GNR_W(14) GNR_W(15) GNR_W(16) GNR_W(17) GNR_W(18) GNR_W(19) GNR_W(20)
        GNR_W(21) GNR_W(22) GNR_W(23) GNR_W(24) GNR_W(25) GNR_W(26) GNR_W(27) GNR_W(28) GNR_W(29) GNR_W(30)
#undef GNR_W
    }
}

// NP pairs of rows (NP even): pair(P, g, mid) issues the six MFMAs of rows 2P, 2P+1 from g[row][hi/lo] and calls
// mid(i) right after its i-th MFMA (i = 0..3).
// Issue budget: with one wave per SIMD a wave issues ONE instruction per four cycles, whatever its kind, so the 32
// matrix-pipe cycles of a v_mfma_f32_32x32x16_bf16 cover seven more instructions and nothing else: the phase is
// scheduled by hand.  mid(i) = one ring read of the NEXT pair's rows (a phase ahead of its use) and, behind the second
// MFMA of each pair, one LDS-DMA piece; the activation conversion of mm3_h is cut into stages of <= 5 VALU behind
// other MFMAs.  Round 1 issued both pieces and four reads back to back after the barrier and a whole conversion
// (19 VALU) in one gap: 7.0 instructions per MFMA, matrix pipe 59 % busy (PMC: profiles/r2_x3_pmc_fwd.txt).
// stores(ph) = global stores the pair functions of phase ph issue (training dumps), 0 outside [0, NP/2):
// they sit in the same in-order vmcnt queue as the LDS-DMA pieces, so the wait for "everything but the
// last DEPTH-1 batches" must allow them too -- counting fewer than were issued only makes the wait
// stricter, never unsafe.
// SKIP empty phases (barrier + DMA, no MFMA) follow when NP/2 is not a multiple of NSLOT: the three requests that would
// target the skipped slots re-request the batch at the unchanged stream offset (the vmcnt bookkeeping counts two
// pieces per phase), and the next layer / weight set starts on slot 0 again.
template <int NP, int SKIP, class PairFn, class StoresFn>
__device__ __forceinline__ void ring_layer(WRing& w, PairFn pair, StoresFn stores) {
    static_assert(NP % 2 == 0, "layers start and end on batch boundaries");
    constexpr int NPH = NP / 2;
    static_assert((NPH + SKIP) % NSLOT == 0 && NPH > DEPTH, "every layer starts on ring slot 0");
#pragma clang loop unroll(full)
    for (int ph = 0; ph < NPH + SKIP; ++ph) {
        int allow = 2 * (DEPTH - 1);
#pragma unroll
        for (int d = 1; d < DEPTH; ++d) allow += (ph - d >= 0 && ph - d < NPH && !(ABL & 64)) ? stores(ph - d) : 0;
        if (!(ABL & 16)) wait_vm_n(allow);
        if (!(ABL & 2)) ring_barrier();
        const int rd = ph % NSLOT, nrd = (ph + 1) % NSLOT, wr = (ph + DEPTH + 1) % NSLOT;
        const bool dummy = ph >= NPH - (DEPTH + 1) && ph < NPH - (DEPTH + 1) + SKIP;
        const unsigned soff = w.soff;
        if (!dummy) w.soff += BATCH_BYTES;
        if (ph < NPH) {
            pair(2 * ph, w.g[0], [&](int i) {
                ring_read_quarter(w, rd, 1, i, w.g[1]);
                if (i == 1) ring_issue_piece<0>(w, wr, soff);
            });
            pair(2 * ph + 1, w.g[1], [&](int i) {
                ring_read_quarter(w, nrd, 0, i, w.g[0]);
                if (i == 1) ring_issue_piece<1>(w, wr, soff);
            });
        } else {
            ring_issue_piece<0>(w, wr, soff);
            ring_issue_piece<1>(w, wr, soff);
#pragma unroll
            for (int i = 0; i < 4; ++i) ring_read_quarter(w, nrd, 0, i, w.g[0]);
        }
    }
}

// converted B operands of one 32-channel input tile: two K=16 steps, hi and lo
struct BTile {
    u32x4 h[2], l[2];
};

// Transform applied to every quad of previous-layer values -- registers r..r+3 (r % 4 == 0) of tile t,
// i.e. the four consecutive channels 32t + 8(r>>2) + 4h + 0..3 of this lane's sample -- right before the
// hi/lo split: activation / ReLU mask, sign-bit collection, dumps.  May modify v.
struct XfNone {
    __device__ __forceinline__ void operator()(int, int, f32x4&) const {}
};
struct XfRelu {
    __device__ __forceinline__ void operator()(int, int, f32x4& v) const {
        // one v_max_i32 on the bit pattern each (negative floats are negative integers; fmaxf and fmed3 both compile
        // to a canonicalising v_max plus the v_max in IEEE mode)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[e];       // (a copy: __builtin_bit_cast of an ext-vector ELEMENT reads element 0 with this hipcc)
            const int b = __builtin_bit_cast(int, x);
            v[e] = __builtin_bit_cast(float, b > 0 ? b : 0);
        }
    }
};

// Training dumps use the "QHL" layout: one 16-byte element {hi01, hi23, lo01, lo23} -- the bf16 (hi, lo) split of
// the four consecutive channels a lane holds, exactly the words the next layer's MFMA operand is built from -- per
// (chunk c, channel quad q, sample j) of a C-channel tensor, at byte c*32*C*4 + q*512 + (j ^ 4(q&3))*16, the two 8-byte
// halves swapped in quads with bit 2 set.  A lane's quad is ONE 16-byte store and a wave's store covers 1 KiB.  (The
// vector-memory path accepts roughly one wave instruction per ~20 cycles per CU whatever its width: dword stores of
// the same data -- the chunk-channel-major layout of the fp32 kernels -- cost 4x the instructions and, measured, +40 %
// on the training forward.)  The weight-gradient kernel reads this layout with transposing LDS reads and never
// converts anything (wgrad3_tr_kernel, gnr_wgrad.hip: the slot XOR and the half swap are its bank-conflict scheme).
// Round 1 dumped the fp32 quad and let the GEMM re-split every element in each of its three workgroup columns.
struct QDump {
    char* base;          // tensor + chunk*32*C*4 (wave-uniform: stays in SGPRs), or nullptr: no dump
    unsigned s0, s1;     // this lane's byte offset inside quad q = 8t + 2(r>>2) + h for (r>>2) even / odd:
                         // h*512 + (j ^ 4(q&3))*16 with q&3 = 2((r>>2)&1) + h
};
__device__ __forceinline__ QDump qdump(float* dst, int C, long chunk, int j, int h) {
    QDump q;
    q.base = dst ? (char*)(dst + chunk * (CHUNK * (long)C)) : nullptr;
    q.s0 = h * 512 + (j ^ (4 * h)) * 16;
    q.s1 = h * 512 + (j ^ (8 + 4 * h)) * 16;
    return q;
}

// kChain3DumpBranch (a per-translation-unit constant of the product that every includer declares BEFORE this header -- a C++ constant, not a
// preprocessor switch, so no -D on a compiler command line can flip it: ADVICE round 5): test qd.base at run time as well.  The branch splits the unrolled layer code
// into basic blocks, which pins the hand-made instruction order better than sched_barrier does (pure MFMAs still move
// across those before machine scheduling): measured on bwd3_chain_kernel 6.5 ms with the branch, 6.9 ms without; on
// fwd3_kernel<true> the same branch cost 190 spilled registers and 1.3 ms.  Compiler behaviour, re-measure on upgrades.
static_assert(kChain3DumpBranch || !kChain3DumpBranch, "declare `namespace gnr { constexpr bool kChain3DumpBranch = ...; }` before including gnr_chain3.h");

struct XfLateNone {
    __device__ __forceinline__ void operator()(int, int, const f32x4&) const {}
};

// DUMP is a template parameter on purpose: a run-time test of qd.base puts a branch into the unrolled layer code and
// each of those costs the register allocator ~200 spills.  late(t, r, v): side effects on the transformed values
// (sign-bit collection of the training forward), kept apart so that the staged version can issue it later.
template <bool WRITEBACK, bool DUMP, class Xf, class Late>
__device__ __forceinline__ void convert_quad(f32x16& src, int r, BTile& dst, int t, Xf& xf, Late& late, const QDump& qd) {
    f32x4 v = {src[r], src[r + 1], src[r + 2], src[r + 3]};
    xf(t, r, v);
    late(t, r, v);
    if (WRITEBACK) { src[r] = v.x; src[r + 1] = v.y; src[r + 2] = v.z; src[r + 3] = v.w; }
    unsigned h0, l0, h1, l1;
    split_pair(v.x, v.y, h0, l0);
    split_pair(v.z, v.w, h1, l1);
    if (DUMP && !(ABL & 32) && (!kChain3DumpBranch || qd.base)) {     // the training dump: the split itself (quads with bit 2 set: halves swapped)
        u32x4* p = (u32x4*)(qd.base + (8 * t + 2 * (r >> 2)) * 512 + (((r >> 2) & 1) ? qd.s1 : qd.s0));
        dump_store(p, (r >> 3) ? u32x4{l0, l1, h0, h1} : u32x4{h0, h1, l0, l1});
    }
    const int u = r >> 3, w = (r & 7) >> 1;
    dst.h[u][w] = h0; dst.h[u][w + 1] = h1;
    dst.l[u][w] = l0; dst.l[u][w + 1] = l1;
}

// The same conversion in stages of <= 5 VALU (inference), each behind another MFMA (ring_layer's issue budget):
// fetch (v_accvgpr_read when the source tile lives in AGPRs) | transform + hi(0,1) | lo(0,1) + hi(2,3) |
// lo(2,3) + dump + operand words | late(): the training forward's sign bits.
struct ConvQuad {
    f32x4 v;
    unsigned h0, h1, l0;
};
__device__ __forceinline__ void conv_fetch(const f32x16& src, int r, ConvQuad& c) {
    c.v = f32x4{src[r], src[r + 1], src[r + 2], src[r + 3]};
}
template <bool WRITEBACK, class Xf>
__device__ __forceinline__ void conv_xf(f32x16& src, int r, int t, Xf& xf, ConvQuad& c) {
    xf(t, r, c.v);
    if (WRITEBACK) { src[r] = c.v.x; src[r + 1] = c.v.y; src[r + 2] = c.v.z; src[r + 3] = c.v.w; }
    c.h0 = hi_pair(c.v.x, c.v.y);
}
__device__ __forceinline__ void conv_mid(ConvQuad& c) {
    c.l0 = lo_pair(c.v.x, c.v.y, c.h0);
    c.h1 = hi_pair(c.v.z, c.v.w);
}
template <bool DUMP>
__device__ __forceinline__ void conv_finish(const ConvQuad& c, int r, BTile& dst, int t, const QDump& qd) {
    const unsigned l1 = lo_pair(c.v.z, c.v.w, c.h1);
    if (DUMP && !(ABL & 32) && (!kChain3DumpBranch || qd.base)) {
        u32x4* p = (u32x4*)(qd.base + (8 * t + 2 * (r >> 2)) * 512 + (((r >> 2) & 1) ? qd.s1 : qd.s0));
        dump_store(p, (r >> 3) ? u32x4{c.l0, l1, c.h0, c.h1} : u32x4{c.h0, c.h1, c.l0, l1});
    }
    const int u = r >> 3, w = (r & 7) >> 1;
    dst.h[u][w] = c.h0; dst.h[u][w + 1] = c.h1;
    dst.l[u][w] = c.l0; dst.l[u][w + 1] = l1;
}

// accumulator tile nt starts from its bias: lane (j, h) register r <-> channel 32nt + (r&3) + 8(r>>2) + 4h
__device__ __forceinline__ void bias_init(f32x16& acc, const float* bias, int nt, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *(const f32x4*)(bias + 32 * nt + 8 * q + 4 * h);
        acc[4 * q + 0] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
    }
}

// ---- one dense layer from the previous layer's accumulators ---------------------------------------
// prev[t] goes through xf (bias is already inside the accumulators) and is split tile by tile underneath
// the MFMAs of the tile before it.  INIT: how acc starts -- from out_bias, from zero (inline C operand),
// or continuing (L5 after its encoding part).  WRITEBACK keeps xf's result in prev (a second layer
// reading the same input then uses XfNone).
enum { INIT_NONE = 0, INIT_BIAS = 1, INIT_ZERO = 2 };

// DUMPS = global stores a conversion issues (0, or 1 when qd dumps the quad: the vmcnt bookkeeping needs the count).
// SKIP: see ring_layer.
template <int NT_IN, int NT_OUT, int INIT, bool WRITEBACK, int DUMPS, int SKIP = 0, class Xf, class Late = XfLateNone>
__device__ __forceinline__ void mm3_h(f32x16 (&prev)[NT_H], f32x16 (&acc)[NT_H], const float* out_bias, int h, WRing& w,
                                      Xf xf, const QDump qd = QDump{nullptr, 0, 0}, Late late = Late()) {
    constexpr bool DUMP = DUMPS != 0;
    constexpr bool DROP_LO_HI = (ABL & 128) && NT_IN == NT_H2 && NT_OUT == NT_H;
    constexpr int PPT = NT_OUT;                 // row pairs per input tile (2 K-steps x NT_OUT rows / 2)
    constexpr int NP = NT_IN * PPT;
    static_assert(PPT >= 2, "at most two conversions per pair");
    // quads of the next input tile converted before pair pt of the current one: its 4 register quads spread over the
    // PPT pairs; q_lo(pt + 1) - q_lo(pt) = the conversions (xf calls) issued inside pair pt: 0, 1 or (PPT = 2) 2
    auto q_lo = [](int pt) { return (4 * pt) / PPT; };
    auto conv_in_pair = [&](int P) {
        const int t = P / PPT, pt = P % PPT;
        return t + 1 < NT_IN ? q_lo(pt + 1) - q_lo(pt) : 0;
    };
    auto stores = [&](int ph) { return DUMPS * (conv_in_pair(2 * ph) + conv_in_pair(2 * ph + 1)); };
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    BTile cur, nxt;
    if (INIT == INIT_BIAS) { bias_init(acc[0], out_bias, 0, h); bias_init(acc[1 % NT_OUT], out_bias, 1 % NT_OUT, h); }
#pragma unroll
    for (int r = 0; r < 16; r += 4) convert_quad<WRITEBACK, DUMP>(prev[0], r, cur, 0, xf, late, qd);
    ring_layer<NP, SKIP>(w, [&](int P, const u32x4 (&g)[2][2], auto mid) {
        const int t = P / PPT, pt = P % PPT;
        const int i0 = 2 * pt, i1 = i0 + 1;
        const int u0 = i0 / NT_OUT, n0 = i0 % NT_OUT, u1 = i1 / NT_OUT, n1 = i1 % NT_OUT;
        const int qa = q_lo(pt), nq = (t + 1 < NT_IN && !(ABL & 4)) ? q_lo(pt + 1) - qa : 0;
        const int tn = t + 1 < NT_IN ? t + 1 : t;       // (index kept in range where nq == 0)
        ConvQuad cq;
        const bool z0 = INIT == INIT_ZERO && t == 0 && u0 == 0, z1 = INIT == INIT_ZERO && t == 0 && u1 == 0;
        acc[n0] = mfma_bf(g[0][0], cur.h[u0], z0 ? zero : acc[n0]);
        mid(0);
        if (nq == 1) conv_fetch(prev[tn], 4 * qa, cq);
        __builtin_amdgcn_sched_barrier(0);
        acc[n1] = mfma_bf(g[1][0], cur.h[u1], z1 ? zero : acc[n1]);
        mid(1);
        __builtin_amdgcn_sched_barrier(0);
        if (nq == 2) {
            convert_quad<WRITEBACK, DUMP>(prev[tn], 4 * qa, nxt, tn, xf, late, qd);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!DROP_LO_HI) acc[n0] = mfma_bf(g[0][1], cur.h[u0], acc[n0]);
        mid(2);
        if (nq == 1) conv_xf<WRITEBACK>(prev[tn], 4 * qa, tn, xf, cq);
        __builtin_amdgcn_sched_barrier(0);
        if (!DROP_LO_HI) acc[n1] = mfma_bf(g[1][1], cur.h[u1], acc[n1]);
        mid(3);
        if (nq == 1) conv_mid(cq);
        __builtin_amdgcn_sched_barrier(0);
        if (nq == 2) {
            convert_quad<WRITEBACK, DUMP>(prev[tn], 4 * (qa + 1), nxt, tn, xf, late, qd);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc[n0] = mfma_bf(g[0][0], cur.l[u0], acc[n0]);
        if (nq == 1) conv_finish<DUMP>(cq, 4 * qa, nxt, tn, qd);
        __builtin_amdgcn_sched_barrier(0);
        acc[n1] = mfma_bf(g[1][0], cur.l[u1], acc[n1]);
        if (nq == 1) late(tn, 4 * qa, cq.v);
        // biases of the tiles the NEXT pair opens
        if (INIT == INIT_BIAS && t == 0) {
#pragma unroll
            for (int i = i0 + 2; i < i0 + 4; ++i)
                if (i >= 2 && i < NT_OUT) bias_init(acc[i], out_bias, i, h);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (pt == PPT - 1 && t + 1 < NT_IN) cur = nxt;
    }, stores);
}

// ---- the 64-slot positional encoding (4 K=16 steps, pre-split in LDS) ------------------------------
template <int NT_OUT>
__device__ __forceinline__ void mm3_enc(const unsigned* enc_col, f32x16 (&acc)[NT_H], const float* out_bias, int h,
                                        WRing& w) {
    constexpr int NP = 4 * NT_OUT / 2;
    static_assert(NT_OUT % 2 == 0, "a row pair stays inside one K step");
    // LDS column layout: word index = tile*16 + (hi? 0 : 8) + u*4 + w, stride 256 threads
    u32x4 bh[4], bl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bh[s][c] = enc_col[((s >> 1) * 16 + (s & 1) * 4 + c) * 256];
            bl[s][c] = enc_col[((s >> 1) * 16 + 8 + (s & 1) * 4 + c) * 256];
        }
    bias_init(acc[0], out_bias, 0, h);
    bias_init(acc[1], out_bias, 1, h);
    ring_layer<NP, 0>(w, [&](int P, const u32x4 (&g)[2][2], auto mid) {
        const int i0 = 2 * P, s = i0 / NT_OUT, n0 = i0 % NT_OUT, n1 = n0 + 1;
        if (s == 0 && n0 + 3 < NT_OUT) {
            bias_init(acc[n0 + 2], out_bias, n0 + 2, h);
            bias_init(acc[n0 + 3], out_bias, n0 + 3, h);
        }
        acc[n0] = mfma_bf(g[0][0], bh[s], acc[n0]);
        mid(0);
        __builtin_amdgcn_sched_barrier(0);
        acc[n1] = mfma_bf(g[1][0], bh[s], acc[n1]);
        mid(1);
        __builtin_amdgcn_sched_barrier(0);
        acc[n0] = mfma_bf(g[0][1], bh[s], acc[n0]);
        mid(2);
        __builtin_amdgcn_sched_barrier(0);
        acc[n1] = mfma_bf(g[1][1], bh[s], acc[n1]);
        mid(3);
        __builtin_amdgcn_sched_barrier(0);
        acc[n0] = mfma_bf(g[0][0], bl[s], acc[n0]);
        acc[n1] = mfma_bf(g[1][0], bl[s], acc[n1]);
        __builtin_amdgcn_sched_barrier(0);
    }, [](int) { return 0; });
}

}  // namespace gnr
