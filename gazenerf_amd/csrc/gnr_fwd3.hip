// gnr_fwd3.hip -- inference forward with the dense layers on bf16 MFMA via a 3-term split ("bf16x3").
//
// Every fp32 operand x is split x = hi + lo (+ ~2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi), and
//     a * b  ~=  a_hi b_hi + a_lo b_hi + a_hi b_lo          (fp32 accumulate in the MFMA)
// i.e. three v_mfma_f32_32x32x16_bf16 (32 matrix-pipe cycles each, 16 k per instruction) instead of
// eight v_mfma_f32_32x32x2_f32 (64 cycles each): 5.3x the fp32-MFMA rate at ~16 mantissa bits per
// operand.  Measured against the reference fixtures (emulation in tools, kernels in
// tests/test_parity_gpu.py): feature map 2e-7 .. 6e-6, bg_alpha <= 3.1e-5 even with the x50 opaque
// density head -- the same order as the fp32 path's own rounding noise and well inside the 1e-4
// contract, whereas plain bf16 is off by 1.5e-2.  SURVEY.md section 7 lists this split as the
// sanctioned "later, measured optimisation"; the fp32 path stays the default.
//
// Structure (differences from gnr_fwd.hip):
//   * a layer's output stays in registers as RAW fp32 accumulators; the NEXT layer converts one
//     32-channel input tile at a time (bias + ReLU + hi/lo split + bf16 pack: 16 registers), spread
//     between its own MFMA batches -- the conversion never stalls the matrix pipe as a block, and only
//     one converted tile (+ the one being built) is live, which frees ~170 registers;
//   * those registers hold a deeper weight prefetch: bf16x3 consumes 2 KiB of weights (hi + lo rows)
//     per 96 matrix-pipe cycles, 5x the fp32 kernel's rate;
//   * the density head is still an fp32 VALU dot (folded into the conversion of h7); ray geometry,
//     encoding (fp32 sincosf, then split), compositing and combine are shared with the fp32 path.
#include "gnr_chain.h"

namespace gnr {
int fail(const char* fmt, ...);

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
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    const f32x2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, bf16x2));
}

// k-order of a K=16 bf16 step s = 2t + u over an activation tile held in the C/D layout: lane-half h
// supplies, as element q (0..7), register r = 8u + q of tile t.
__host__ __device__ inline int dlayout3_channel(int step16, int h, int q) {
    const int t = step16 >> 1, r = 8 * (step16 & 1) + q;
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// ---------------------------------------------------------------------------------------------
// packed weight stream: per row (K=16 step, n-tile): [64 lanes x 8 bf16 hi][64 lanes x 8 bf16 lo]
// = 2 KiB, rows in execution order; same total bytes as the fp32 stream (4 B per weight).
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int l3_enc_steps(int l) { return (l == L0 || l == L5) ? 4 : 0; }
__host__ __device__ constexpr int l3_h_steps(int l) { return l == L0 ? 0 : (l == LR2 ? 2 * NT_H2 : 2 * NT_H); }
__host__ __device__ constexpr size_t l3_rows(int l) { return (size_t)(l3_enc_steps(l) + l3_h_steps(l)) * layer_nt(l); }
__host__ __device__ constexpr size_t l3_row_offset(int l) {
    size_t o = 0;
    for (int i = 0; i < l; ++i) o += l3_rows(i);
    return o;
}
constexpr size_t ROWS3 = l3_row_offset(N_CHAIN);          // 2652 rows = 5.43 MB per MLP
static_assert(ROWS3 * 512 == PACKED_FLOATS, "bf16x3 stream must fit the fp32 stream's workspace slot");

struct Pack3Params {
    const float* w[N_CHAIN];
    int ld[N_CHAIN], n_out[N_CHAIN], hcol[N_CHAIN], kh[N_CHAIN];
    unsigned short* packed;
};

__global__ void pack3_kernel(const Pack3Params pp) {
    const size_t total = ROWS3 * 64 * 8;                     // (row, lane, q) triples
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / 512;
        const int lane = (int)((e % 512) / 8), q = (int)(e % 8);
        int l = 0;
        size_t off = 0;
        while (l + 1 < N_CHAIN && row >= off + l3_rows(l)) { off += l3_rows(l); ++l; }
        const int nt_n = layer_nt(l);
        const int s = (int)((row - off) / nt_n), nt = (int)((row - off) % nt_n);
        const int h = lane >> 5, n = 32 * nt + (lane & 31);
        int col = -1;
        if (s < l3_enc_steps(l)) {
            col = enc_channel(8 * s + q, h);
        } else {
            const int k = dlayout3_channel(s - l3_enc_steps(l), h, q);
            if (k < pp.kh[l]) col = pp.hcol[l] + k;
        }
        float v = 0.0f;
        if (n < pp.n_out[l] && col >= 0) v = pp.w[l][(size_t)n * pp.ld[l] + col];
        const unsigned hi = bf16_rne(v);
        const unsigned lo = bf16_rne(v - bf16_to_f32(hi));
        unsigned short* r = pp.packed + row * 1024;          // 2 KiB = 1024 shorts
        r[lane * 8 + q] = (unsigned short)hi;
        r[512 + lane * 8 + q] = (unsigned short)lo;
    }
}

// ---------------------------------------------------------------------------------------------
// weight ring: the workgroup's four waves consume the SAME rows, so the stream goes through LDS once
// per workgroup instead of four times through the vector L1 (which at 2 KiB per 96 matrix-pipe cycles
// per wave is past the 64 B/clk the texture path delivers).  LDS-DMA (buffer_load_dwordx4 ... lds)
// fills a ring of NSLOT batches of RB_ROWS rows; wave w fetches row w of each batch (2 KiB = 2 pieces).
//
// Per batch k ("phase"), every wave:
//   s_waitcnt vmcnt(2 (DEPTH-1))   its own share of batch k+1 has landed
//   s_barrier                      -> batch k+1 is complete and visible; every wave has issued (hence
//                                     fetched the operands of) all MFMAs of batch k-1
//   request batch k+DEPTH+1        into the slot batch k-1 occupied            (NSLOT = DEPTH + 2)
//   ds_read rows 2,3 of batch k;  MFMAs of rows 0,1;  ds_read rows 0,1 of batch k+1;  MFMAs of rows 2,3
// The DMA is inline asm (hipcc would otherwise put a vmcnt(0) in front of every LDS read that might
// alias it); hipcc's own vmcnt bookkeeping stays correct because extra outstanding operations only make
// its counted waits stricter, and ours count only operations issued after the ones we wait for.
// Addressing rules (LDS dest = M0 + inst_offset + lane*16, M0 beyond 64 KiB, zero fill past
// num_records) are pinned by tools/ubench/ldsdma_probe.hip.
// ---------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Timing experiments only (results are wrong with any bit set): build with -DGNR_ABLATE=<bits>.
//   1 no LDS-DMA requests   2 no barriers   4 no activation conversion   8 no ring reads   16 no vmcnt waits
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

struct WRing {
    i32x4 rs;                // buffer descriptor of the packed stream (num_records = exact bytes)
    unsigned voff;           // lane*16 + wave*2048
    unsigned soff;           // stream offset of the next batch to request
    unsigned wr;             // LDS address (M0) of this wave's row in the slot to fill next
    unsigned wr_end;         // wr wraps here
    unsigned rd;             // ring offset of the batch being consumed
    const char* lane_base;   // ring + lane*16
    u32x4 g[2][2][2];        // [pair][row][hi/lo]: rows 0,1 and rows 2,3 of the current batch
};

struct RingTicket {
    unsigned soff, wr;
};

// Measured (tools/ablate_fwd3.sh): a piece blocks the issuing wave for ~94 cycles, and that is not
// contention between the four waves -- spreading their requests over the phase (one wave per MFMA group)
// made the kernel 12 % slower, because the barrier then waits for whichever wave is stalled.
__device__ __forceinline__ void ring_issue(const WRing& w, const RingTicket& t) {
    if (ABL & 1) return;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(w.voff), "s"(w.rs), "s"(t.soff), "s"(t.wr)
        : "memory");
}

__device__ __forceinline__ RingTicket ring_advance(WRing& w) {
    const RingTicket t = {w.soff, w.wr};
    w.soff += BATCH_BYTES;
    w.wr += BATCH_BYTES;
    if (w.wr == w.wr_end) w.wr -= RING_BYTES;
    return t;
}

__device__ __forceinline__ void ring_request(WRing& w) { ring_issue(w, ring_advance(w)); }

__device__ __forceinline__ void ring_read_pair(const WRing& w, unsigned slot_off, int pair, u32x4 (&g)[2][2]) {
    if (ABL & 8) return;
    const char* p = w.lane_base + slot_off + pair * 4096;
    g[0][0] = *(const u32x4*)(p);
    g[0][1] = *(const u32x4*)(p + 1024);
    g[1][0] = *(const u32x4*)(p + 2048);
    g[1][1] = *(const u32x4*)(p + 3072);
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
    w.wr = (unsigned)(size_t)ring + wave * 2048u;
    w.wr_end = w.wr + RING_BYTES;
    w.rd = 0;
    w.lane_base = ring + lane * 16;
    if (ABL & 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) w.g[q >> 2][(q >> 1) & 1][q & 1] = u32x4{(unsigned)lane, 1u, 2u, 3u};
    }
    if (ABL & 4) {}
#pragma unroll
    for (int k = 0; k <= DEPTH; ++k) ring_request(w);
}

// first rows into registers: call after ring_init, with no other barrier in between
__device__ __forceinline__ void ring_start(WRing& w) {
    wait_vm<2 * DEPTH>();
    __builtin_amdgcn_s_barrier();
    ring_read_pair(w, 0, 0, w.g[0]);
}

// NP pairs of rows (NP even): pair(P, g) issues the six MFMAs of rows 2P, 2P+1 from g[row][hi/lo]
template <int NP, class PairFn>
__device__ __forceinline__ void ring_layer(WRing& w, PairFn pair) {
    static_assert(NP % 2 == 0, "layers start and end on batch boundaries");
#pragma clang loop unroll(full)
    for (int ph = 0; ph < NP / 2; ++ph) {
        if (!(ABL & 16)) wait_vm<2 * (DEPTH - 1)>();
        if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
        ring_request(w);
        ring_read_pair(w, w.rd, 1, w.g[1]);
        pair(2 * ph, w.g[0]);
        const unsigned nrd = (w.rd + BATCH_BYTES == RING_BYTES) ? 0u : w.rd + BATCH_BYTES;
        ring_read_pair(w, nrd, 0, w.g[0]);
        pair(2 * ph + 1, w.g[1]);
        w.rd = nrd;
    }
}

// converted B operands of one 32-channel input tile: two K=16 steps, hi and lo
struct BTile {
    u32x4 h[2], l[2];
};

struct NoSide {
    __device__ __forceinline__ void operator()(int, int, float, float) const {}
};

// optional ReLU + split of register pair (r, r+1) of accumulator tile `src` (bias already inside)
template <bool RELU, class Side>
__device__ __forceinline__ void convert_pair(const f32x16& src, int r, BTile& dst, int t, Side side) {
    float a = src[r], b = src[r + 1];
    if (RELU) { a = fmaxf(a, 0.0f); b = fmaxf(b, 0.0f); }
    side(t, r, a, b);
    unsigned hi, lo;
    split_pair(a, b, hi, lo);
    const int u = r >> 3, w = (r & 7) >> 1;
    dst.h[u][w] = hi;
    dst.l[u][w] = lo;
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
// prev[t] (bias inside; ReLU here if RELU_IN) is split tile by tile underneath the MFMAs of the tile
// before it.  INIT: acc starts from out_bias (else it continues, e.g. L5 after its encoding part).
template <int NT_IN, int NT_OUT, bool INIT, bool RELU_IN, class Side = NoSide>
__device__ __forceinline__ void mm3_h(const f32x16 (&prev)[NT_H], f32x16 (&acc)[NT_H], const float* out_bias, int h,
                                      WRing& w, Side side = Side()) {
    constexpr int PPT = NT_OUT;                 // row pairs per input tile (2 K-steps x NT_OUT rows / 2)
    constexpr int NP = NT_IN * PPT;
    BTile cur, nxt;
    if (INIT) { bias_init(acc[0], out_bias, 0, h); bias_init(acc[1 % NT_OUT], out_bias, 1 % NT_OUT, h); }
#pragma unroll
    for (int r = 0; r < 16; r += 2) convert_pair<RELU_IN>(prev[0], r, cur, 0, side);
    ring_layer<NP>(w, [&](int P, const u32x4 (&g)[2][2]) {
        const int t = P / PPT, pt = P % PPT;
        const int i0 = 2 * pt, i1 = i0 + 1;
        const int u0 = i0 / NT_OUT, n0 = i0 % NT_OUT, u1 = i1 / NT_OUT, n1 = i1 % NT_OUT;
        // biases of the tiles the NEXT pair opens
        if (INIT && t == 0) {
#pragma unroll
            for (int i = i0 + 2; i < i0 + 4; ++i)
                if (i >= 2 && i < NT_OUT) bias_init(acc[i], out_bias, i, h);
        }
        acc[n0] = mfma_bf(g[0][0], cur.h[u0], acc[n0]);
        acc[n1] = mfma_bf(g[1][0], cur.h[u1], acc[n1]);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < NT_IN && !(ABL & 4)) {
#pragma unroll
            for (int pr = ((2 * pt) * 8) / (2 * PPT); pr < ((2 * pt + 1) * 8) / (2 * PPT); ++pr)
                convert_pair<RELU_IN>(prev[t + 1], 2 * pr, nxt, t + 1, side);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[n0] = mfma_bf(g[0][1], cur.h[u0], acc[n0]);
        acc[n1] = mfma_bf(g[1][1], cur.h[u1], acc[n1]);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < NT_IN && !(ABL & 4)) {
#pragma unroll
            for (int pr = ((2 * pt + 1) * 8) / (2 * PPT); pr < ((2 * pt + 2) * 8) / (2 * PPT); ++pr)
                convert_pair<RELU_IN>(prev[t + 1], 2 * pr, nxt, t + 1, side);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[n0] = mfma_bf(g[0][0], cur.l[u0], acc[n0]);
        acc[n1] = mfma_bf(g[1][0], cur.l[u1], acc[n1]);
        __builtin_amdgcn_sched_barrier(0);
        if (pt == PPT - 1 && t + 1 < NT_IN) cur = nxt;
    });
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
    ring_layer<NP>(w, [&](int P, const u32x4 (&g)[2][2]) {
        const int i0 = 2 * P, s = i0 / NT_OUT, n0 = i0 % NT_OUT, n1 = n0 + 1;
        if (s == 0 && n0 + 3 < NT_OUT) {
            bias_init(acc[n0 + 2], out_bias, n0 + 2, h);
            bias_init(acc[n0 + 3], out_bias, n0 + 3, h);
        }
        acc[n0] = mfma_bf(g[0][0], bh[s], acc[n0]);
        acc[n1] = mfma_bf(g[1][0], bh[s], acc[n1]);
        acc[n0] = mfma_bf(g[0][1], bh[s], acc[n0]);
        acc[n1] = mfma_bf(g[1][1], bh[s], acc[n1]);
        acc[n0] = mfma_bf(g[0][0], bl[s], acc[n0]);
        acc[n1] = mfma_bf(g[1][0], bl[s], acc[n1]);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// LDS: [ring 48 KiB][encoding 32 x 256 words][per-wave bias table (N_CHAIN layers + density row) x 4]
constexpr int BIAS_ROWS = N_CHAIN + 1;
constexpr size_t FWD3_LDS_BYTES = RING_BYTES + (size_t)(ENC_STEPS * 256 + WAVES_PER_WG * BIAS_ROWS * H) * sizeof(float);
static_assert(FWD3_LDS_BYTES <= 160 * 1024, "LDS budget");

__global__ __launch_bounds__(256, 1) void fwd3_kernel(const FwdParams fp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* ring = (char*)smem;
    unsigned* enc_lds = (unsigned*)(smem + RING_BYTES / 4);          // [32 words][256 threads]
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* bias_lds = smem + RING_BYTES / 4 + ENC_STEPS * 256 + wave * (BIAS_ROWS * H);
    const int j = lane & 31, h = lane >> 5;
    const GnrProblem& p = fp.prob;

    WRing w;
    ring_init(w, fp.ws[0].packed, (unsigned)(fp.n_streams * ROWS3 * 2048), ring, lane, wave);

    // every wave stays in the barrier protocol: past the end it recomputes the last chunk and stores nothing
    const long chunk_raw = (long)blockIdx.x * WAVES_PER_WG + wave;
    const bool live = chunk_raw < fp.n_chunks;
    const long chunk = live ? chunk_raw : fp.n_chunks - 1;
    const int cpr = fp.chunks_per_ray;
    const long ray_g = chunk / cpr;
    const int c_in = (int)(chunk - ray_g * cpr);
    const int b = (int)(ray_g / p.n_rays);
    const int ray = (int)(ray_g - (long)b * p.n_rays);
    const int i = c_in * CHUNK + j;
    const bool valid = i < p.n_samples;
    const long row = chunk * CHUNK + j;

    const Ray r = make_ray(p, b, ray);
    const int ic = valid ? i : p.n_samples - 1;
    const float z0 = sample_edge(p, r.oz, ray_g, ic);
    const float z1 = sample_edge(p, r.oz, ray_g, ic + 1);
    const float delta = valid ? __fmul_rn(__fsub_rn(z1, z0), r.l) : 0.0f;
    const float px = __fadd_rn(r.ox, __fmul_rn(__fmul_rn(r.dx, r.l), z0));
    const float py = __fadd_rn(r.oy, __fmul_rn(__fmul_rn(r.dy, r.l), z0));
    const float pz = __fadd_rn(r.oz, __fmul_rn(__fmul_rn(r.dz, r.l), z0));
    if (fp.want_wl && h == 0 && live) fp.zval[row] = z0;

    // encoding: fp32 sincosf, then hi/lo split into two pre-converted B tiles kept in LDS
    unsigned* enc_col = enc_lds + tid;
    {
        float e[ENC_STEPS];
        encode_point(px, py, pz, h, e);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) {
                    unsigned hi, lo;
                    split_pair(e[16 * T + 8 * u + 2 * wd], e[16 * T + 8 * u + 2 * wd + 1], hi, lo);
                    enc_col[(T * 16 + u * 4 + wd) * 256] = hi;
                    enc_col[(T * 16 + 8 + u * 4 + wd) * 256] = lo;
                }
    }
    ring_start(w);

    f32x16 A[NT_H], Bv[NT_H];
#pragma unroll 1
    for (int s = 0; s < fp.n_streams; ++s) {
        const StreamWs& ws = fp.ws[s];
        {   // this wave's biases (latent codes folded in) + the density row -> LDS
            const long bstride = (long)p.batch * H;
            const float* bsrc = ws.bias + (long)b * H;
            for (int q = lane; q < N_CHAIN * (H / 4); q += 64) {
                const int l = q / (H / 4), c4 = q - l * (H / 4);
                *(f32x4*)(bias_lds + l * H + 4 * c4) = *(const f32x4*)(bsrc + l * bstride + 4 * c4);
            }
            for (int q = lane; q < H / 4; q += 64)
                *(f32x4*)(bias_lds + N_CHAIN * H + 4 * q) = *(const f32x4*)(ws.wsig + 4 * q);
        }
        auto bl = [&](int l) { return bias_lds + l * H; };

        mm3_enc<NT_H>(enc_col, A, bl(0), h, w);                                // L0 -> A (pre-activation)
#pragma unroll 1
        for (int rep = 0; rep < 2; ++rep) {                                    // L1..L4
            mm3_h<NT_H, NT_H, true, true>(A, Bv, bl(2 * rep + 1), h, w);
            mm3_h<NT_H, NT_H, true, true>(Bv, A, bl(2 * rep + 2), h, w);
        }
        mm3_enc<NT_H>(enc_col, Bv, bl(5), h, w);                               // L5: encoding part, then h4 part
        mm3_h<NT_H, NT_H, false, true>(A, Bv, bl(5), h, w);
        mm3_h<NT_H, NT_H, true, true>(Bv, A, bl(6), h, w);                     // L6
        mm3_h<NT_H, NT_H, true, true>(A, Bv, bl(7), h, w);                     // L7 (pre-activation h7 in Bv)
        // RGB0 consumes h7 = relu(Bv); the density head rides on the conversion (fp32 VALU dot)
        float sig = 0.0f;
        const float* wsg = bias_lds + N_CHAIN * H + 4 * h;
        mm3_h<NT_H, NT_H, true, true>(Bv, A, bl(LR0), h, w, [&](int t, int rr, float a, float bb) {
            const int ch = 32 * t + (rr & 3) + 8 * (rr >> 2);
            sig = fmaf(wsg[ch], a, sig);
            sig = fmaf(wsg[ch + 1], bb, sig);
        });
        sig += __shfl_xor(sig, 32);
        sig += ws.wsig[H];
        mm3_h<NT_H, NT_H2, true, false>(A, Bv, bl(LR1), h, w);                 // RGB1 from y0 (no activation)
        mm3_h<NT_H2, NT_F, true, true>(Bv, A, bl(LR2), h, w);                  // RGB2 from relu(y1)
        if (live) composite_chunk(A, sig, delta, z0, ws, chunk, row, lane, fp.want_wl != 0);
    }
    wait_vm<0>();       // no LDS-DMA may outlive the wave
}

void launch_prep3(const GnrProblem& p, int n_streams, const GnrWeights* const* wts, StreamWs* ws, hipStream_t stream) {
    const int vp = ENC_CH + p.shape_dims + p.gaze_dims;
    for (int s = 0; s < n_streams; ++s) {
        Pack3Params pp;
        for (int l = 0; l < N_CHAIN; ++l) {
            if (l <= 7) {
                pp.w[l] = wts[s]->fea_w[l];
                pp.ld[l] = (l == 0) ? vp : (l == 5 ? vp + H : H);
                pp.n_out[l] = H; pp.hcol[l] = (l == 5) ? vp : 0; pp.kh[l] = (l == 0) ? 0 : H;
            } else if (l == LR0) {
                pp.w[l] = wts[s]->rgb_w[0]; pp.ld[l] = H; pp.n_out[l] = H; pp.hcol[l] = 0; pp.kh[l] = H;
            } else if (l == LR1) {
                pp.w[l] = wts[s]->rgb_w[1]; pp.ld[l] = H + p.appea_dims; pp.n_out[l] = H2; pp.hcol[l] = 0; pp.kh[l] = H;
            } else {
                pp.w[l] = wts[s]->rgb_w[2]; pp.ld[l] = H2; pp.n_out[l] = p.feat_nc; pp.hcol[l] = 0; pp.kh[l] = H2;
            }
        }
        pp.packed = (unsigned short*)ws[s].packed;
        hipLaunchKernelGGL(pack3_kernel, dim3(1024), dim3(256), 0, stream, pp);
    }
}

void launch_fwd3(const FwdParams& fp, hipStream_t stream) {
    const unsigned grid = (unsigned)((fp.n_chunks + WAVES_PER_WG - 1) / WAVES_PER_WG);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)fwd3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FWD3_LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(fwd3_kernel, dim3(grid), dim3(256), FWD3_LDS_BYTES, stream, fp);
}

}  // namespace gnr
