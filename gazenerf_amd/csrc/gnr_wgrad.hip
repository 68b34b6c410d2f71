// gnr_wgrad.hip -- weight gradients of the per-sample FC layers (gfx950).
//
//   dW[n][k] = sum_s dY[s][n] * X[s][k]      (s over all M samples of the call)
//
// dY and X are the chunk-channel-major ("CCM") dumps of the dgrad chain / training forward:
// element (chunk c, channel n, sample j) at c*32*C + n*32 + j, so the [rows][32 samples]
// operand tile of one chunk is ONE contiguous block: the threads stream it with fully coalesced
// float4 loads.  The contraction index is the sample; its order is free, so lane-half
// h of v_mfma_f32_32x32x2_f32 takes samples 16h..16h+15 of the chunk and a lane reads its operand
// values for 4 consecutive MFMA steps with one ds_read_b128.  The LDS image keeps the 128-byte rows
// but XORs the 16-byte piece index with (row/2)%8, which makes both the ds_write_b128 of the staging
// pass and the ds_read_b128 of the MFMA pass bank-conflict free (MI355X LDS: 64 banks for b128,
// non-contiguous 16-lane groups).
//
// Two fp32 kernels share that operand scheme (wgrad3_tr_kernel is the bf16x3 one):
//   wgrad2w_kernel     the whole MLP (and the upsampler's 1032 x 516 product): ONE software-pipelined workgroup per CU, eight
//                      waves on 16x16x4 MFMAs, LDS-DMA ring, operand prefetch; 192 x 192, 96 x 192 and 192 x 64 tiles -- see
//                      its header below;
//   wgrad_kernel       the upsampler's narrow channels-first products: WN x WK waves of XN x XK MFMA tiles,
//                      i.e. a (32 XN WN) x (32 XK WK) workgroup tile chosen per shape by padded work
//                          128 x 128  (2x2 waves of 2x2)      192 x 64 / 64 x 192  (2x2 waves of 3x1 / 1x3)
//                           96 x  96  (3x1 waves of 1x3)
//                      (one 128 x 128 tiling for everything spent 9 % of the MLP's weight-gradient MFMAs on padding),
//                      chunks double-buffered through LDS (the next chunk's global loads are in flight during the
//                      current chunk's MFMAs), two workgroups per CU.
// The sample range is split (per image) over workgroups; the tiles of one split are placed on one XCD back-to-back so
// the operand rows they share hit that XCD's L2.  Per-split partial tiles are summed in a fixed order by
// wgrad_reduce_kernel (deterministic; the reference trains with cudnn.deterministic, train.py:57).
//
// Riding along: column sums of dY (bias gradients, per image) and vec^T X for a per-sample vector (density-head
// gradient).  wgrad_kernel takes them from the operand registers between the MFMAs, in every workgroup (GNR_WG_RIDERS
// below: with two workgroups per CU that is free, and a single round of workgroups ends with its slowest member);
// wgrad2w_kernel shares them between the waves that hold the same rows.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace gnr { constexpr bool kChain3DumpBranch = false; }      // see gnr_chain3.h
#include "gnr_canary.h"
#include "gnr_chain3.h"
#include "gnr_wgrad.h"

namespace gnr {

// Tuning switches (tools/ubench/wgrad_bench.hip builds the variants side by side):
//   GNR_WG_RIDERS  how wgrad_kernel computes the bias / density riders.  0 none (timing floor; wrong gradients)
//                  1 from the staging registers, in the workgroups that own them   2 from the operand registers inside
//                  the MFMA loop, in the workgroups that own them   3 the same in every workgroup (default: measured
//                  fastest -- a single round of workgroups ends with its slowest member, and in this two-workgroups-
//                  per-CU kernel a few VALU adds between the MFMAs cost nothing: 384^2 layer 2.65 / 2.77 / 2.92 /
//                  2.99 ms for modes 3 / 1 / 2 / 0)
//   GNR_PIPE_ABL   timing experiments on wgrad2w_kernel, see there
#ifndef GNR_WG_RIDERS
#define GNR_WG_RIDERS 3
#endif
constexpr int RM = GNR_WG_RIDERS;

constexpr int WG_TN = 128, WG_TK = 128;    // tile of the bf16x3 kernel (and the largest fp32 tile: 16384 floats)

struct WgradParams {
    const float* A;      // dY, CCM with C = lda
    const float* B;      // X,  CCM with C = ldb
    int lda, ldb;
    int n_valid, k_valid;
    int tiles_n, tiles_k;
    int batch, spi;               // splits per image
    long chunks_per_image;
    long chunks_per_split;
    float* partial;               // [batch*spi][tiles_n][tiles_k][TN][TK]
    float* colsum_part;           // [batch*spi][tiles_n*TN]
    const float* vec;             // [M] or NULL
    float* vec_part;              // [batch*spi][tiles_k*TK]
    // operand addressing (floats): element (image b, local chunk lc, channel n, sample j) at
    //   b*img + lc*chunk + n*row + j.  CCM dumps: row = 32, chunk = 32*ld, img = chunks_per_image*32*ld;
    //   channels-first images [B][C][P]: row = P, chunk = 32, img = C*P.
    long a_row, a_chunk, a_img, b_row, b_chunk, b_img;
    int linear_map;               // wgrad2w_kernel: workgroup id -> contiguous ranges of (split, tile) per XCD instead of split % 8
    // Round 4: the fp32 chain kernels dump in the S16 layout (gnr_chain16.h): 16-sample sub-chunks of 16 ld floats, channel n
    // in the 64-byte row s16_row(n).  A chunk of 32 samples is still one block of 32 ld floats (two sub-chunks), a 16-byte
    // piece (4 consecutive samples of one channel) still contiguous: piece p of channel n sits at byte
    // (p >> 2) 64 ld + s16_row(n) 64 + (p & 3) 16 of its chunk.  0: rows of 32 samples (the encoding dump, the bf16x3 path's
    // own formats, images).
    int a_s16, b_s16;
    unsigned long long* clk;      // shader-clock probe (gnr_internal.h) or nullptr
};

template <int N>
__device__ __forceinline__ void wait_vm_dma() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS position (in floats) of 16-byte piece `c` (0..7) of tile row `n`
__device__ __forceinline__ int swz(int n, int c) { return n * CHUNK + ((c ^ ((n >> 1) & 7)) << 2); }

// S16 source addressing of an LDS-DMA instruction (lane l fills slot l%8 of tile row 8 i + l/8 with piece p): byte offset
// inside the chunk, relative to the tile's first row (a multiple of 16 channels) = lane part + piece part.
//   tile row 8 i + r  ->  64-byte row 16 (i >> 1) + 4 (r & 3) + 2 (i & 1) + (r >> 2)       (s16_row, gnr_chain16.h)
__device__ __forceinline__ unsigned s16_lane_off(int r, int p, int ld) {
    return 256u * (unsigned)(r & 3) + 64u * (unsigned)(r >> 2) + (unsigned)(p >> 2) * (unsigned)(64 * ld) + 16u * (unsigned)(p & 3);
}
__device__ __forceinline__ unsigned s16_piece_off(unsigned i) { return (i >> 1) * 1024u + (i & 1u) * 128u; }

template <int WN, int WK, int XN, int XK, bool VEC>
__global__ __launch_bounds__(64 * WN * WK, 2) void wgrad_kernel(const WgradParams wp) {
    constexpr int THREADS = 64 * WN * WK;
    constexpr int TN = 32 * XN * WN, TK = 32 * XK * WK;
    constexpr int A_IT = TN * 8 / THREADS, B_IT = TK * 8 / THREADS;      // float4 pieces per thread per chunk
    static_assert(A_IT * THREADS == TN * 8 && B_IT * THREADS == TK * 8, "tile rows must spread evenly over the threads");
    constexpr int BUF = (TN + TK) * CHUNK;
    __shared__ __attribute__((aligned(16))) float lds[2][BUF];              // [buffer][A rows | B rows][32 samples]
    // XCD-aware placement: workgroup id -> (xcd, slot); an XCD runs the tiles of one split back-to-back
    const int tiles = wp.tiles_n * wp.tiles_k;
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int split = xcd + 8 * (slot / tiles);
    const int tile = slot % tiles;
    if (split >= wp.batch * wp.spi) return;
    const int tn = tile / wp.tiles_k, tk = tile - tn * wp.tiles_k;
    const ClkProbe clk0 = clk_begin();

    // Two workgroups share each SIMD's matrix pipe.  With equal priority they advance in lock-step
    // and reach their per-chunk barrier together, idling the pipe; a static priority for the wave
    // in the odd hardware slot (HW_ID.wave_id, scalar) lets one stream MFMAs while the other
    // stages/syncs, and the roles alternate by themselves.
    if (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) __builtin_amdgcn_s_setprio(1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WK, wk = wave - wn * WK;
    const int li = lane & 31, lh = lane >> 5;
    const int b = split / wp.spi, sp = split - b * wp.spi;
    const long c0 = (long)b * wp.chunks_per_image + (long)sp * wp.chunks_per_split;
    long c1 = c0 + wp.chunks_per_split;
    const long cmax = (long)(b + 1) * wp.chunks_per_image;
    if (c1 > cmax) c1 = cmax;
    const bool do_cs = tk == 0, do_vs = VEC && tn == 0;     // workgroup-uniform

    // staging: float4 piece f = tid + THREADS q: tile row f/8, piece f%8 == tid%8 (rows clamped into the
    // buffer: rows/cols beyond it only feed outputs that are dropped)
    const int pc = tid & 7;
    const float* ga[A_IT];
    const float* gb[B_IT];
    int lposa[A_IT], lposb[B_IT];
#pragma unroll
    for (int q = 0; q < A_IT; ++q) {
        const int r = (tid + THREADS * q) >> 3;
        int n = tn * TN + r; if (n >= wp.lda) n = wp.lda - 1;
        const long ea = wp.a_s16 ? (long)(pc >> 2) * (16 * wp.lda) + (long)s16_row(n) * 16 + 4 * (pc & 3) : (long)n * wp.a_row + 4 * pc;
        ga[q] = wp.A + (long)b * wp.a_img + ea - (long)b * wp.chunks_per_image * wp.a_chunk;
        lposa[q] = swz(r, pc);
    }
#pragma unroll
    for (int q = 0; q < B_IT; ++q) {
        const int r = (tid + THREADS * q) >> 3;
        int k = tk * TK + r; if (k >= wp.ldb) k = wp.ldb - 1;
        const long eb = wp.b_s16 ? (long)(pc >> 2) * (16 * wp.ldb) + (long)s16_row(k) * 16 + 4 * (pc & 3) : (long)k * wp.b_row + 4 * pc;
        gb[q] = wp.B + (long)b * wp.b_img + eb - (long)b * wp.chunks_per_image * wp.b_chunk;
        lposb[q] = TN * CHUNK + swz(r, pc);
    }
    const long strideA = wp.a_chunk, strideB = wp.b_chunk;
    const float* pv = VEC ? wp.vec + 4 * pc : nullptr;
    f32x4 ra[A_IT], rb[B_IT], rv;
    float cs[A_IT], vs[B_IT];
#pragma unroll
    for (int q = 0; q < A_IT; ++q) cs[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < B_IT; ++q) vs[q] = 0.0f;
    auto gload = [&](long c) {
#pragma unroll
        for (int q = 0; q < A_IT; ++q) ra[q] = *(const f32x4*)(ga[q] + c * strideA);
#pragma unroll
        for (int q = 0; q < B_IT; ++q) rb[q] = *(const f32x4*)(gb[q] + c * strideB);
        if (RM == 1 && do_vs) rv = *(const f32x4*)(pv + c * CHUNK);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < A_IT; ++q) *(f32x4*)&lds[buf][lposa[q]] = ra[q];
#pragma unroll
        for (int q = 0; q < B_IT; ++q) *(f32x4*)&lds[buf][lposb[q]] = rb[q];
        // riders, from the staging registers (every chunk is staged exactly once)
        if (RM == 1 && do_cs) {
#pragma unroll
            for (int q = 0; q < A_IT; ++q) cs[q] += (ra[q].x + ra[q].y) + (ra[q].z + ra[q].w);
        }
        if (RM == 1 && do_vs) {
#pragma unroll
            for (int q = 0; q < B_IT; ++q)
                vs[q] += fmaf(rv.x, rb[q].x, rv.y * rb[q].y) + fmaf(rv.z, rb[q].z, rv.w * rb[q].w);
        }
    };

    f32x16 acc[XN][XK];
#pragma unroll
    for (int x = 0; x < XN; ++x)
#pragma unroll
        for (int y = 0; y < XK; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;

    // operand read positions of this lane: rows wn*32*XN + 32x + li (A) / wk*32*XK + 32y + li (B), pieces 4lh + g
    int apos[XN][4], bpos[XK][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int x = 0; x < XN; ++x) apos[x][g] = swz(wn * 32 * XN + 32 * x + li, 4 * lh + g);
#pragma unroll
        for (int y = 0; y < XK; ++y) bpos[y][g] = TN * CHUNK + swz(wk * 32 * XK + 32 * y + li, 4 * lh + g);
    }

    if (c0 < c1) {
        gload(c0);
        lstore(0);
    }
    __syncthreads();
    // in-loop riders (RM 2/3): from the operand registers of the waves that own the rows
    float csl[XN], vsl[XK];
#pragma unroll
    for (int x = 0; x < XN; ++x) csl[x] = 0.0f;
#pragma unroll
    for (int y = 0; y < XK; ++y) vsl[y] = 0.0f;
    auto loop = [&](auto tag) {
        constexpr bool RIDE = decltype(tag)::value;
        const bool my_cs = RIDE && wk == 0 && (RM == 3 || do_cs);
        const bool my_vs = RIDE && VEC && wn == 0 && (RM == 3 || do_vs);
        for (long c = c0; c < c1; ++c) {
            const int buf = (int)((c - c0) & 1);
            if (c + 1 < c1) gload(c + 1);
            f32x4 v4[4];
            if (RIDE && VEC) {
#pragma unroll
                for (int g = 0; g < 4; ++g) v4[g] = *(const f32x4*)(wp.vec + c * CHUNK + 16 * lh + 4 * g);
            }
            // all operand reads of the chunk first (one exposed LDS latency per chunk, not per 4 steps)
            f32x4 av[XN][4], bv[XK][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int x = 0; x < XN; ++x) av[x][g] = *(const f32x4*)&lds[buf][apos[x][g]];
#pragma unroll
                for (int y = 0; y < XK; ++y) bv[y][g] = *(const f32x4*)&lds[buf][bpos[y][g]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int x = 0; x < XN; ++x)
#pragma unroll
                        for (int y = 0; y < XK; ++y) acc[x][y] = mfma32(av[x][g][e], bv[y][g][e], acc[x][y]);
                    if (RIDE) {
                        if (my_cs) {
#pragma unroll
                            for (int x = 0; x < XN; ++x) csl[x] += av[x][g][e];
                        }
                        if (my_vs) {
#pragma unroll
                            for (int y = 0; y < XK; ++y) vsl[y] = fmaf(v4[g][e], bv[y][g][e], vsl[y]);
                        }
                    }
                }
            }
            if (c + 1 < c1) lstore(buf ^ 1);
            __syncthreads();
        }
    };
    if (RM == 3 || (RM == 2 && (do_cs || do_vs))) loop(std::true_type{});
    else loop(std::false_type{});

    clk_end(clk0, wp.clk);
    float* pt = wp.partial + (((long)split * wp.tiles_n + tn) * wp.tiles_k + tk) * (long)(TN * TK);
#pragma unroll
    for (int x = 0; x < XN; ++x)
#pragma unroll
        for (int y = 0; y < XK; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = wn * 32 * XN + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int jx = wk * 32 * XK + y * 32 + li;
                pt[i * TK + jx] = acc[x][y][r];
            }
    if (RM >= 2) {
        if (do_cs && wk == 0) {
#pragma unroll
            for (int x = 0; x < XN; ++x) {
                const float t = csl[x] + __shfl_xor(csl[x], 32);
                if (lh == 0) wp.colsum_part[(long)split * (wp.tiles_n * TN) + tn * TN + wn * 32 * XN + 32 * x + li] = t;
            }
        }
        if (do_vs && wn == 0) {
#pragma unroll
            for (int y = 0; y < XK; ++y) {
                const float t = vsl[y] + __shfl_xor(vsl[y], 32);
                if (lh == 0) wp.vec_part[(long)split * (wp.tiles_k * TK) + tk * TK + wk * 32 * XK + 32 * y + li] = t;
            }
        }
    }
    // riders: the eight threads tid%8 = 0..7 of a row hold its eight 4-sample pieces
    if (RM == 1 && do_cs) {
#pragma unroll
        for (int q = 0; q < A_IT; ++q) {
            float t = cs[q];
            t += __shfl_xor(t, 1);
            t += __shfl_xor(t, 2);
            t += __shfl_xor(t, 4);
            if (pc == 0) wp.colsum_part[(long)split * (wp.tiles_n * TN) + tn * TN + ((tid + THREADS * q) >> 3)] = t;
        }
    }
    if (RM == 1 && do_vs) {
#pragma unroll
        for (int q = 0; q < B_IT; ++q) {
            float t = vs[q];
            t += __shfl_xor(t, 1);
            t += __shfl_xor(t, 2);
            t += __shfl_xor(t, 4);
            if (pc == 0) wp.vec_part[(long)split * (wp.tiles_k * TK) + tk * TK + ((tid + THREADS * q) >> 3)] = t;
        }
    }
}

#ifndef GNR_PIPE_ABL
#define GNR_PIPE_ABL 0      // timing experiments on wgrad2w_kernel (wrong results): 1 no riders, 2 no DMA requests in the loop, 4 no per-chunk wait + barrier
#endif
constexpr int PABL = GNR_PIPE_ABL;

// ---------------------------------------------------------------------------------------------
// wgrad2w_kernel (round 3): ONE software-pipelined workgroup per CU on a 192 x 192 tile -- operands of chunk c + 2 arrive by
// LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction, XOR swizzle applied to the source address) into a ring of
// three LDS buffers, the operands of the next slot are read into a second register set underneath the current slot's MFMAs, one
// s_waitcnt vmcnt(0) + s_barrier per chunk -- with EIGHT waves (two per SIMD) of 96 x 48 on v_mfma_f32_16x16x4_f32.  (Round 2's
// wgrad_pipe_kernel ran the same pipeline with four waves of 96 x 96 on 32x32x2; round 5 moved its last shape, the 64 encoding
// columns, here and removed it: HISTORY.md 3.3.)  Why two waves: with one wave per SIMD every ds_read / LDS-DMA / rider
// instruction between two MFMAs stalls the matrix pipe (~13 cycles each; ablations: reads + loop 2.8 %, DMA issue 1.4 %, riders 1.6 %);
// with a second wave on the SIMD those issue beside the other wave's MFMAs (tools/ubench/mfma_2w.hip) and a VALU
// instruction costs ~4.  A wave's registers: 72 accumulators (6 x 3 tiles of 16 x 16) + 2 x 36 operand registers.
//
// Contraction order (free): lane group g = l >> 4 of a 16x16x4 MFMA supplies one sample; group g takes the 8 samples
// 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3 of the chunk, as two 16-byte pieces (slot t: piece g + 4 ((t + rot) & 1)),
// component e of a piece at MFMA step 4 t + e.  A lane's operand for 4 steps of one tile is ONE ds_read_b128 of the
// swizzled [row][32 samples] image.  Pieces g and g + 4 (not 2 g, 2 g + 1): ds_read_b128 executes in four phases of 16
// NON-contiguous lanes -- {0-3, 12-15, 20-23, 24-27}, {4-7, 8-11, 16-19, 28-31} and the same + 32 -- so a phase mixes two
// lane groups on different row quartets; with the image's XOR swizzle (piece ^ (row/2)%8, built for 32-row tiles) pieces
// 2 g of one group and 2 g + 2 of the next collide on half the rows (PMC: 4 conflict cycles per read, 6 % of the
// kernel), pieces g and g + 1 never do.
// Riders: the 4 tiles_k waves-columns that hold the same dY rows share the 8 steps: holder hc takes component hc & 3 of
// slot 0 (rot = hc >> 2), or of both slots when there are only 4 holders (tiles_k = 1); the loop body is instantiated
// for the 4 components and picked once per wave.  The density-head dot (one layer in twelve) is taken by the waves of
// the first row block alone, through a 0 / 1 scale factor instead of a branch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma16w(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// XA = 16-row tiles per wave in the row direction: 6 = the 192 x 192 workgroup tile; 3 (round 5) = a 96 x 192 tile of 48 x 48
// waves for RGB_layer_2's 66 rows beyond its first 192 (until then a 96 x 96 tile of wgrad_kernel: 0.87 ms at 0.58 of the matrix
// pipe per 2 M samples).  Its 36 DMA pieces per chunk do not divide by the eight waves: the first wave of every SIMD takes
// five, the second four -- the same split as the de-phasing (PH) of the loop, so every count stays a compile-time constant.
// XB = 16-column tiles per wave: 3 = 192 columns; 1 (round 5) = a 192 x 64 tile of 96 x 16 waves for the two 384 x 64 encoding-column
// products per weight set (until then wgrad_pipe_kernel<3, 1, false, 2>, one wave per SIMD: 0.87 ms at 0.78 of the matrix pipe).
// NB = LDS ring buffers.  3: chunk k + 2 is requested during chunk k (one chunk period = 3.8 us of lead on the 192 x 192 tile).  The
// 192 x 64 tile's chunk period is 1.3 us and its traffic 10.7 bytes per CU and clock: with one 32 KiB chunk in flight per CU the
// kernel waited on HBM latency (0.795 ms = 4.65 TB/s); NB = 4 keeps two in flight (chunk k + 3 requested during chunk k).
template <bool VEC, bool CS2, int XA = 6, int XB = 3, int NB = 3>
__global__ __launch_bounds__(512, 1) void wgrad2w_kernel(const WgradParams wp) {
    static_assert(!VEC || NB == 3, "the density-vector loads share the in-order vmcnt queue: counted for a ring of three");
    constexpr int TN = 32 * XA, TK = 64 * XB, WKG = 4;            // wave grid 2 (rows) x 4 (columns): (16 XA) x (16 XB) per wave
    constexpr int WR = 16 * XA, WC = 16 * XB;                     // rows / columns per wave
    constexpr int BUF_BYTES = (TN + TK) * CHUNK * 4;              // 48 KiB (XA = 3: 36 KiB)
    constexpr int NP_ALL = BUF_BYTES / 1024;                      // DMA pieces of 1 KiB per chunk
    constexpr int NP0 = (NP_ALL + 7) / 8, NP1 = (NP_ALL + 3) / 8; // ... of a wave with wave < 4 / wave >= 4 (piece 8 j + wave < NP_ALL)
    constexpr int NPIECE = NP0;
    static_assert(NP_ALL % 4 == 0 && (TN / 8) % 4 == 0, "the four waves of a phase group take pieces of the same operand");
    __shared__ __attribute__((aligned(1024))) char lds[NB * BUF_BYTES];
    const int tiles = wp.tiles_n * wp.tiles_k;
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    int split, tile;
    if (wp.linear_map) {
        // image operands: any number of (split, tile) pairs, an XCD takes a contiguous range of them (the tiles of a split
        // share its operand rows in that XCD's L2)
        const int total = wp.batch * wp.spi * tiles, per_xcd = (total + 7) >> 3;
        const int gi = xcd * per_xcd + slot;
        if (slot >= per_xcd || gi >= total) return;
        split = gi / tiles;
        tile = gi - split * tiles;
    } else {
        split = xcd + 8 * (slot / tiles);
        tile = slot % tiles;
        if (split >= wp.batch * wp.spi) return;
    }
    const int tn = tile / wp.tiles_k, tk = tile - tn * wp.tiles_k;
    const ClkProbe clk0 = clk_begin();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WKG, wk = wave % WKG;
    const int li = lane & 15, lg = lane >> 4;
    const int b = split / wp.spi, sp = split - b * wp.spi;
    const long c0 = (long)b * wp.chunks_per_image + (long)sp * wp.chunks_per_split;
    long c1 = c0 + wp.chunks_per_split;
    const long cmax = (long)(b + 1) * wp.chunks_per_image;
    if (c1 > cmax) c1 = cmax;
    const int nchunks = (int)(c1 - c0);

    // Operand addressing.  Chunk-channel-major dumps: chunk k of the split is one block of ld rows x 128 bytes, a DMA
    // piece (8 rows) is 1 KiB of it.  Channels-first images [B][C][P] (the upsampler's weight gradients): row r of the
    // tile is P floats long, chunk k = its pixels 32 k .. 32 k + 31; the LDS image is the same [row][32] -- only the
    // lane's source offset (row stride P) and the two scalar strides change.  Rows >= n_valid read zeros through the
    // descriptor's bound in both layouts.
    const bool img = wp.a_row != CHUNK;
    const unsigned kstride_a = img ? 128u : (unsigned)(wp.lda * CHUNK * 4), kstride_b = img ? 128u : (unsigned)(wp.ldb * CHUNK * 4);
    const unsigned pstride_a = img ? (unsigned)(8 * wp.a_row * 4) : 1024u, pstride_b = img ? (unsigned)(8 * wp.b_row * 4) : 1024u;
    auto desc = [&](const float* base, long ld, long tile_row0, unsigned chunk_bytes, int valid, long row, long img_stride, int s16) {
        unsigned long long a;
        long bytes;
        if (s16) {         // fp32 chain dumps: the tile's first row sits tile_row0 * 16 floats into the chunk
            a = (unsigned long long)(base + c0 * (CHUNK * ld) + tile_row0 * 16);
            bytes = (long)nchunks * chunk_bytes - tile_row0 * 64;
        } else if (img) {
            const long start = ((long)sp * wp.chunks_per_split) * CHUNK;              // first pixel of the split
            a = (unsigned long long)(base + (long)b * img_stride + tile_row0 * row + start);
            bytes = ((long)valid - tile_row0) * row * 4 - start * 4;
        } else {
            a = (unsigned long long)(base + c0 * (CHUNK * ld) + tile_row0 * CHUNK);
            bytes = (long)nchunks * chunk_bytes - tile_row0 * CHUNK * 4;
        }
        if (bytes < 0) bytes = 0;
        if (bytes > 0xFFFFFFFFL) bytes = 0xFFFFFFFFL;
        i32x4 r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
        r.z = __builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
        r.w = 0x00020000;
        return r;
    };
    const i32x4 rsa = desc(wp.A, wp.lda, (long)tn * TN, kstride_a, wp.n_valid, wp.a_row, wp.a_img, wp.a_s16);
    const i32x4 rsb = desc(wp.B, wp.ldb, (long)tk * TK, kstride_b, wp.k_valid, wp.b_row, wp.b_img, wp.b_s16);
    // lane l fills slot l%8 of row 8i + l/8 with source piece (l%8) ^ ((4i + l/16) % 8): odd i =
    // the piece index ^ 4
    const int pe = (lane & 7) ^ (lane >> 4);
    const unsigned vpiece = (unsigned)pe * 16u;
    const unsigned voff_even_a = wp.a_s16 ? s16_lane_off(lane >> 3, pe, wp.lda) : (unsigned)(lane >> 3) * (img ? (unsigned)(wp.a_row * 4) : 128u) + vpiece;
    const unsigned voff_even_b = wp.b_s16 ? s16_lane_off(lane >> 3, pe, wp.ldb) : (unsigned)(lane >> 3) * (img ? (unsigned)(wp.b_row * 4) : 128u) + vpiece;
    const unsigned voff_odd_a = wp.a_s16 ? s16_lane_off(lane >> 3, pe ^ 4, wp.lda) : voff_even_a ^ 64u;
    const unsigned voff_odd_b = wp.b_s16 ? s16_lane_off(lane >> 3, pe ^ 4, wp.ldb) : voff_even_b ^ 64u;
    const unsigned lds0 = (unsigned)(size_t)&lds[0];
    constexpr int PA = TN / 8;                                    // A pieces per chunk (24), then TK / 8 B pieces
    // scalar source offset of this wave's piece j inside a chunk (piece i = 8 rows of the tile)
    const int wave_hi = wave >= 4 ? 4 : 0;                        // pieces 8 j + wave_hi .. + 3 belong to this wave's phase group
    unsigned poff[NPIECE];
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
        const bool isa = 8 * j + wave_hi < PA;
        const unsigned i = isa ? (unsigned)(8 * j + wave) : (unsigned)(8 * j + wave) - PA;
        poff[j] = (isa ? wp.a_s16 : wp.b_s16) ? s16_piece_off(i) : i * (isa ? pstride_a : pstride_b);
    }
    // piece j of this wave's share of chunk c0 + k, into ring buffer `buf`: global piece index 8 j + wave.  isa (an A piece?) is
    // the caller's: a compile-time constant inside the loop (j and the phase group are), a uniform branch in the prologue
    auto dma_piece = [&](int k, int buf, int j, bool isa) {
        const unsigned gp = (unsigned)(8 * j + wave);
        const unsigned i = isa ? gp : gp - PA;
        const unsigned l = lds0 + (unsigned)buf * BUF_BYTES + (isa ? 0u : (unsigned)(TN * CHUNK * 4)) + i * 1024u;
        const unsigned so = (unsigned)k * (isa ? kstride_a : kstride_b) + poff[j];
        unsigned keep;
        if (isa)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"((i & 1) ? voff_odd_a : voff_even_a), "s"(rsa), "s"(l), "s"(so) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"((i & 1) ? voff_odd_b : voff_even_b), "s"(rsb), "s"(l), "s"(so) : "memory");
    };
    auto dma_chunk = [&](int k, int buf) {                        // prologue only
#pragma unroll
        for (int j = 0; j < NPIECE; ++j)
            if (8 * j + wave_hi < NP_ALL) {
                if (8 * j + wave_hi < PA) dma_piece(k, buf, j, true);
                else dma_piece(k, buf, j, false);
            }
    };

    // rider shares: holders of the same dY rows = the (tk, wk) waves
    const int hc = tk * WKG + wk;                                 // 0 .. 4 tiles_k - 1
    const int rot = CS2 ? 0 : ((hc >> 2) & 1);                    // which piece is "slot 0"
    // the 8 (slot, component) steps of a chunk go to the first 8 holders; with three column tiles (image operands only)
    // holders 8 .. 11 contribute zeros
    const float cscale = (CS2 || hc < 8) ? 1.0f : 0.0f;
    // operand read offsets (bytes within a ring buffer) of slot t: row wn*96 + 16 x + li (A) / wk*48 + 16 y + li (B),
    // piece 2 lg + ((t + rot) & 1).  The swizzle term depends on (row/2)%8 = (li/2)%8 only: tile x is a constant
    // + 2048 bytes.
    int apos[2], bpos[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        apos[t] = 4 * swz(wn * WR + li, lg + 4 * ((t + rot) & 1));
        bpos[t] = 4 * (TN * CHUNK + swz(wk * WC + li, lg + 4 * ((t + rot) & 1)));
    }
    float csl[XA], vsl[XB];
#pragma unroll
    for (int x = 0; x < XA; ++x) csl[x] = 0.0f;
#pragma unroll
    for (int y = 0; y < XB; ++y) vsl[y] = 0.0f;
    f32x4 acc[XA][XB];
#pragma unroll
    for (int x = 0; x < XA; ++x)
#pragma unroll
        for (int y = 0; y < XB; ++y) acc[x][y] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float vscale = (VEC && wn == 0 && tn == 0) ? 1.0f : 0.0f;    // the density dot is taken by the first row block

    f32x4 opa[2][XA], opb[2][XB], vv[2], vnext;
    auto read_ops = [&](int buf, int t, f32x4 (&a)[XA], f32x4 (&bb)[XB]) {
        const char* pa = lds + buf * BUF_BYTES + apos[t];
        const char* pb = lds + buf * BUF_BYTES + bpos[t];
#pragma unroll
        for (int x = 0; x < XA; ++x) a[x] = *(const f32x4*)(pa + x * (16 * CHUNK * 4));
#pragma unroll
        for (int y = 0; y < XB; ++y) bb[y] = *(const f32x4*)(pb + y * (16 * CHUNK * 4));
    };
    // density vector: the samples of slot t of this lane group, scaled by 0 / 1
    auto load_vec = [&](int k, int t) -> f32x4 {
        const long c = c0 + (k < nchunks ? k : nchunks - 1);
        const f32x4 v = *(const f32x4*)(wp.vec + c * CHUNK + 4 * lg + 16 * ((t + rot) & 1));
        return f32x4{v.x * vscale, v.y * vscale, v.z * vscale, v.w * vscale};
    };

    // prologue: chunks 0 and 1 in flight; slot 0 of chunk 0 into operand set 0
#pragma unroll
    for (int c = 0; c < NB - 1; ++c) dma_chunk(c, c);
    if (NP0 == NP1 || wave < 4) wait_vm_dma<(NB - 2) * NP0>();     // everything but the last NB - 2 chunks' pieces
    else wait_vm_dma<(NB - 2) * NP1>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_ops(0, 0, opa[0], opb[0]);
    if (VEC) vv[0] = load_vec(0, 0);

    // One chunk = two slots of four MFMA steps (18 MFMAs each).  Operand set t holds slot t; while slot t is multiplied
    // the other set receives the next slot (the same chunk's slot 1, or slot 0 of chunk k+1 after that chunk's barrier).
    // PH de-phases the two waves of a SIMD: they pass every barrier together, so with the same code both would issue
    // their LDS reads and DMA requests at the same moments and neither would have an MFMA to issue meanwhile
    // (tools/ubench/mfma_2w.hip: two in-phase waves 92 %, two independent ones 97 %).  The second wave of each SIMD
    // (PH = 1) issues the next slot's reads after MFMA step 1 instead of before step 0 and its DMA pieces a step later.
    auto chunk = [&](int k, int b0, int b1, int bl, auto etag, auto phtag) {      // bl: the buffer chunk k - 1 just left = chunk k + NB - 1's
        constexpr int E = decltype(etag)::value;
        constexpr int PH = decltype(phtag)::value;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            auto next_reads = [&]() {
                if (t == 0) {
                    read_ops(b0, 1, opa[1], opb[1]);
                    // both density-vector loads of the coming slots go out HERE, a slot ahead of the LDS-DMA pieces of
                    // t = 1: the compiler's counted wait for them (it cannot see the asm requests in the same in-order
                    // vmcnt queue) then never waits on a piece that has just been issued
                    if (VEC && PH == 0) { vv[1] = load_vec(k, 1); vnext = load_vec(k + 1, 0); }
                } else {
                    read_ops(b1, 0, opa[0], opb[0]);
                    if (VEC && PH == 0) vv[0] = vnext;
                }
            };
            if (t == 1 && !(PABL & 4)) {
                // chunk k+1 has landed (NB = 3: it is the only one in flight; NB = 4: chunk k+2's pieces may still be out -- vmcnt
                // retires in order)
                wait_vm_dma<(NB - 3) * (PH ? NP1 : NP0)>();
                __builtin_amdgcn_s_barrier();                          // ... in every wave; ring buffer bl is free
                asm volatile("" ::: "memory");
            }
            if (PH == 0) next_reads();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int x = 0; x < XA; ++x)
#pragma unroll
                    for (int y = 0; y < XB; ++y) acc[x][y] = mfma16w(opa[t][x][e], opb[t][y][e], acc[x][y]);
                if (e == E && (t == 0 || CS2) && !(PABL & 1)) {
#pragma unroll
                    for (int x = 0; x < XA; ++x) csl[x] = fmaf(opa[t][x][e], cscale, csl[x]);
                }
                // the density dot belongs to the first wave row (wn == 0 == the PH = 0 waves: round 5 -- until then the second wave
                // of every SIMD ran the same FMAs against a zero factor)
                if (VEC && PH == 0) {
#pragma unroll
                    for (int y = 0; y < XB; ++y) vsl[y] = fmaf(vv[t][e], opb[t][y][e], vsl[y]);
                }
                if (PH == 1 && e == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    next_reads();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the request for chunk k+2 goes out two pieces per MFMA step of the second slot (after the barrier that
                // freed its ring buffer): steps 0-2 in the first wave of a SIMD, 1-3 in the second
                if (t == 1 && e >= PH && e < 3 + PH && !(PABL & 2)) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int jj = 2 * (e - PH); jj < 2 * (e - PH) + 2; ++jj)
                        if (jj < (PH ? NP1 : NP0)) dma_piece(k + NB - 1, bl, jj, 8 * jj + 4 * PH < PA);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto run = [&](auto etag, auto phtag) {
        int b0 = 0;
        for (int k = 0; k < nchunks; ++k) {
            const int b1 = b0 == NB - 1 ? 0 : b0 + 1, bl = b0 == 0 ? NB - 1 : b0 - 1;
            chunk(k, b0, b1, bl, etag, phtag);
            b0 = b1;
        }
    };
    auto run_e = [&](auto etag) {
        if (wave < 4) run(etag, std::integral_constant<int, 0>{});        // waves w and w + 4 share SIMD w % 4
        else run(etag, std::integral_constant<int, 1>{});
    };
    switch (hc & 3) {
        case 0: run_e(std::integral_constant<int, 0>{}); break;
        case 1: run_e(std::integral_constant<int, 1>{}); break;
        case 2: run_e(std::integral_constant<int, 2>{}); break;
        default: run_e(std::integral_constant<int, 3>{}); break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the zero-filled tail requests

    clk_end(clk0, wp.clk);
    float* pt = wp.partial + (((long)split * wp.tiles_n + tn) * wp.tiles_k + tk) * (long)(TN * TK);
#pragma unroll
    for (int x = 0; x < XA; ++x)
#pragma unroll
        for (int y = 0; y < XB; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = wn * WR + x * 16 + 4 * lg + r;      // D register r of a 16x16 tile: row 4 (l>>4) + r
                const int jx = wk * WC + y * 16 + li;
                pt[i * TK + jx] = acc[x][y][r];
            }
    // rider shares: colsum_part[split][hc][tiles_n*TN], vec_part[split][0][tiles_k*TK]
    {
        const int Qc = wp.tiles_k * WKG;
#pragma unroll
        for (int x = 0; x < XA; ++x) {
            float t = csl[x];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (lg == 0) wp.colsum_part[((long)split * Qc + hc) * (wp.tiles_n * TN) + tn * TN + wn * WR + 16 * x + li] = t;
        }
        if (VEC && wn == 0 && tn == 0) {
#pragma unroll
            for (int y = 0; y < XB; ++y) {
                float t = vsl[y];
                t += __shfl_xor(t, 16);
                t += __shfl_xor(t, 32);
                if (lg == 0) wp.vec_part[(long)split * (wp.tiles_k * TK) + tk * TK + wk * WC + 16 * y + li] = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad3_tr_kernel: the bf16x3 weight-gradient GEMM from PRE-SPLIT operands, with nothing but LDS-DMA, transposing
// LDS reads and MFMAs in its loop.
//
// The bf16x3 chain kernels hold every activation / gradient value as a (hi, lo) bf16 pair already -- the next layer's
// own B operand -- packed as the four consecutive channels of one sample.  They dump exactly that ("QHL" layout:
// element (chunk c, channel quad q, sample j) = 16 bytes {hi01, hi23, lo01, lo23} at c*32*C*4 + q*512 +
// (j ^ 4(q&3))*16, with the two 8-byte halves swapped ({lo01, lo23, hi01, hi23}) in quads with bit 2 set; same bytes
// and the same 16-byte stores as the fp32 quads of round 1).  The contraction of
// dW = dY^T X runs over samples, i.e. the MFMA operands are the TRANSPOSE of that packing: ds_read_b64_tr_b16
// (semantics pinned by tools/ubench/tr_read_probe.hip: in a 16-lane group lane L supplies the 8-byte address of key row
// L/4, column quad L%4 and receives column L of the four keys) delivers it: keys = 4 consecutive samples, columns = 16
// channels, two reads = the 8 samples x 1 channel a lane feeds to v_mfma_f32_32x32x16_bf16.  The sample slot is XORed
// with 4(q&3) in the dump so that the four quads of a 16-lane group hit four different 64-byte bank windows.
//
// Structure = round 2's wgrad_pipe_kernel: one workgroup per CU, 2x2 waves of 3 x XK tiles (192-row tiles), the operands of chunk
// c+1 read into a second register set under the MFMAs of chunk c, LDS-DMA ring of three buffers -- but with 1728
// matrix-pipe cycles per chunk instead of 9216 the requests run TWO chunks ahead (the buffer of chunk c is free as soon
// as its operands are in registers): 96 KiB in flight per CU.  HBM bounds the kernel: 3.2 GB per 384^2 layer at
// M = 1 M -> 0.51 ms at 6.3 TB/s against 0.37 ms of MFMA time.
// Riders: the bias column sums / density dot are unpacked from the operand registers (hi + lo) on one K-step out of
// 2 tiles_k (resp. 2 tiles_n), rotating over the waves that hold the same rows; wgrad_reduce_kernel adds the shares.
// ---------------------------------------------------------------------------------------------
typedef short tr_s16x4 __attribute__((ext_vector_type(4)));
#ifndef GNR_TR_ABL
#define GNR_TR_ABL 0        // timing experiments (wrong results): 1 no riders, 2 no DMA requests in the loop, 4 no operand reads in the loop
#endif
constexpr int TRABL = GNR_TR_ABL;

__device__ __forceinline__ unsigned long long tr_read(const char* p) {
    return __builtin_bit_cast(unsigned long long,
                              __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_s16x4*)p));
}

template <int XK, bool VEC>
__global__ __launch_bounds__(512, 1) void wgrad3_tr_kernel(const WgradParams wp) {
    constexpr int XN = 3, TN = 64 * XN, TK = 64 * XK;
    constexpr int QA = TN / 4, QB = TK / 4;                      // channel quads per operand tile
    constexpr int A_BYTES = QA * 512, BUF_BYTES = (QA + QB) * 512;
    constexpr int PPW = (QA + QB) / 8;                           // 1 KiB DMA pieces (2 quads) per loader wave per chunk
    static_assert(PPW * 8 == QA + QB, "pieces must split over the 4 loader waves");
    constexpr int VEC_BYTES = VEC ? 1024 : 0;                     // per ring slot: the chunk's 32 density values (+ slack)
    constexpr int NREQ = PPW + (VEC ? 1 : 0);                     // vm requests per loader wave per chunk
    __shared__ __attribute__((aligned(1024))) char lds[3 * BUF_BYTES + 3 * VEC_BYTES];
    const int tiles = wp.tiles_n * wp.tiles_k;
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int split = xcd + 8 * (slot / tiles);
    const int tile = slot % tiles;
    if (split >= wp.batch * wp.spi) return;
    const int tn = tile / wp.tiles_k, tk = tile - tn * wp.tiles_k;
    const ClkProbe clk0 = clk_begin();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = split / wp.spi, sp = split - b * wp.spi;
    const long c0 = (long)b * wp.chunks_per_image + (long)sp * wp.chunks_per_split;
    long c1 = c0 + wp.chunks_per_split;
    const long cmax = (long)(b + 1) * wp.chunks_per_image;
    if (c1 > cmax) c1 = cmax;
    const int nchunks = (int)(c1 - c0);

    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves
        // An LDS-DMA instruction blocks the issuing wave ~60-100 cycles: twelve per chunk inside the MFMA waves cost
        // 30 % of a 1728-cycle chunk (measured), so the requests get their own wave per SIMD (a few SALU + VMEM
        // instructions per chunk: no issue pressure on the MFMA wave next to it).
        const int lw = wave - 4;
        const unsigned chunk_a = (unsigned)(wp.lda * CHUNK * 4), chunk_b = (unsigned)(wp.ldb * CHUNK * 4);
        auto desc = [&](const float* base, long ld, long quad0, unsigned chunk_bytes) {
            const unsigned long long a = (unsigned long long)(base + c0 * (CHUNK * ld) + quad0 * 128);
            long bytes = (long)nchunks * chunk_bytes - quad0 * 512;
            if (bytes < 0) bytes = 0;
            i32x4 r;
            r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
            r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
            r.z = __builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
            r.w = 0x00020000;
            return r;
        };
        const i32x4 rsa = desc(wp.A, wp.lda, (long)tn * QA, chunk_a);
        const i32x4 rsb = desc(wp.B, wp.ldb, (long)tk * QB, chunk_b);
        i32x4 rsv = rsa;
        if (VEC) {
            const unsigned long long a = (unsigned long long)(wp.vec + c0 * CHUNK);
            rsv.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
            rsv.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
            rsv.z = __builtin_amdgcn_readfirstlane(nchunks * CHUNK * 4);
        }
        const unsigned lds0 = (unsigned)(size_t)&lds[0];
        const unsigned voff = (unsigned)lane * 16u;
        // chunk c0 + k into ring buffer `buf`: this wave's PPW pieces (linear image: a piece = 2 quads x 32 slots x 16
        // bytes) and, for the density rider, the chunk's 32 vector values (every loader wave issues them -- same bytes
        // to the same place -- so that all four count the same vmcnt); requests past the split read zeros
        auto dma_chunk = [&](int k, int buf) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
#pragma unroll
            for (int jj = 0; jj < PPW; ++jj) {
                const int j = lw + 4 * jj;
                const bool isa = j < QA / 2;
                const unsigned i = (unsigned)(isa ? j : j - QA / 2);
                const unsigned l = lds0 + (unsigned)buf * BUF_BYTES + (isa ? 0u : (unsigned)A_BYTES) + i * 1024u;
        const unsigned so = (unsigned)k * (isa ? chunk_a : chunk_b) + i * 1024u;
                if (isa)
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(voff), "s"(rsa), "s"(l), "s"(so) : "memory");
                else
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(voff), "s"(rsb), "s"(l), "s"(so) : "memory");
            }
            if (VEC) {
                const unsigned l = lds0 + 3u * BUF_BYTES + (unsigned)buf * 1024u, so = (unsigned)k * (CHUNK * 4);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(voff), "s"(rsv), "s"(l), "s"(so) : "memory");
            }
            asm volatile("s_mov_b32 m0, %0" :: "s"(keep));
        };
        dma_chunk(0, 0);
        dma_chunk(1, 1);
        wait_vm_dma<NREQ>();                                  // chunk 0 has landed
        __builtin_amdgcn_s_barrier();
        int b2 = 2;                                           // ring buffer of chunk k+2 == the one chunk k-1 just left
        for (int k = 0; k < nchunks; ++k) {
            if (!(TRABL & 2)) dma_chunk(k + 2, b2);
            if (TRABL & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else wait_vm_dma<NREQ>();                         // chunk k+1 has landed (requested a period ago)
            __builtin_amdgcn_s_barrier();
            b2 = b2 == 2 ? 0 : b2 + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves
    const int wn = wave >> 1, wk = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // transposing-read addresses: 16-lane group g = lane/16: k-half lh = g/2, channel half ch = g%2 (channels 16 ch ..
    // 16 ch + 15 of a 32-channel tile); inside the group key row r = (lane%16)/4, column quad c = lane%4.
    //   quad = tile_quad0 + 4 ch + c (tile_quad0 % 8 == 0 -> quad & 3 == c);  sample = 16 s + 8 lh + 4 u + r;
    //   byte = quad*512 + (sample ^ 4c)*16 + plane*8.   One register per (s, u); tiles and planes are immediates.
    // Quads with bit 2 set (the ch = 1 half of every tile) hold {lo, hi} instead of {hi, lo}: the two 16-lane groups
    // that a 32-lane LDS pass serves together then read different 8-byte halves of the same 16-byte bank slots.
    int aoff[2][4], boff[2][4];                                   // [plane][s, u]
    {
        const int ch = (lane >> 4) & 1, r = (lane & 15) >> 2, c = lane & 3;
#pragma unroll
        for (int su = 0; su < 4; ++su) {
            const int sample = 16 * (su >> 1) + 8 * lh + 4 * (su & 1) + r;
            const int sl = (sample ^ (4 * c)) * 16;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                aoff[pl][su] = (wn * 8 * XN + 4 * ch + c) * 512 + sl + 8 * (pl ^ ch);
                boff[pl][su] = A_BYTES + (wk * 8 * XK + 4 * ch + c) * 512 + sl + 8 * (pl ^ ch);
            }
        }
    }
    f32x16 acc[XN][XK];
#pragma unroll
    for (int x = 0; x < XN; ++x)
#pragma unroll
        for (int y = 0; y < XK; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
    // Operand registers: hi planes of the current K-step (ah, bh) and of the next one (nah, nbh), lo planes of the
    // current one (al, bl): 72 registers beside the 144 accumulators.
    u32x4 ah[XN], bh[XK], nah[XN], nbh[XK], al[XN], bl[XK];
    auto rd = [&](const char* base, int o0, int o1, int imm) {
        const unsigned long long p0 = tr_read(base + o0 + imm), p1 = tr_read(base + o1 + imm);
        return u32x4{(unsigned)p0, (unsigned)(p0 >> 32), (unsigned)p1, (unsigned)(p1 >> 32)};
    };
    auto read_planes = [&](int buf, int sst, int plane, u32x4 (&a)[XN], u32x4 (&bb)[XK]) {
        if (TRABL & 4) return;
        const char* pb = lds + buf * BUF_BYTES;
#pragma unroll
        for (int x = 0; x < XN; ++x) a[x] = rd(pb, aoff[plane][2 * sst], aoff[plane][2 * sst + 1], x * 4096);
#pragma unroll
        for (int y = 0; y < XK; ++y) bb[y] = rd(pb, boff[plane][2 * sst], boff[plane][2 * sst + 1], y * 4096);
    };
    auto mma = [&](const u32x4 (&a)[XN], const u32x4 (&bb)[XK]) {
#pragma unroll
        for (int x = 0; x < XN; ++x)
#pragma unroll
            for (int y = 0; y < XK; ++y) acc[x][y] = mfma_bf(a[x], bb[y], acc[x][y]);
    };
    if (TRABL & 4) {
#pragma unroll
        for (int x = 0; x < XN; ++x) ah[x] = nah[x] = al[x] = u32x4{(unsigned)lane, 1u, 2u, 3u};
#pragma unroll
        for (int y = 0; y < XK; ++y) bh[y] = nbh[y] = bl[y] = u32x4{(unsigned)lane, 5u, 6u, 7u};
    }
    // riders: hi + lo of the 8 samples a lane holds of one operand, summed / weighted by the density vector; this
    // wave's turn is one K-step out of 2 tiles_k (dY rows) resp. 2 tiles_n (X rows), rotating over the waves that hold
    // the same rows
    float csl[XN], vsl[XK];
#pragma unroll
    for (int x = 0; x < XN; ++x) csl[x] = 0.0f;
#pragma unroll
    for (int y = 0; y < XK; ++y) vsl[y] = 0.0f;
    const int Qc = wp.tiles_k * 2, qc = tk * 2 + wk, Qv = wp.tiles_n * 2, qv = tn * 2 + wn;
    auto usum = [](const u32x4& h, const u32x4& l) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            t += __builtin_bit_cast(float, h[w] << 16) + __builtin_bit_cast(float, h[w] & 0xffff0000u);
            t += __builtin_bit_cast(float, l[w] << 16) + __builtin_bit_cast(float, l[w] & 0xffff0000u);
        }
        return t;
    };
    auto udot = [](const u32x4& h, const u32x4& l, const f32x4& v0, const f32x4& v1) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float va = w < 2 ? v0[2 * w] : v1[2 * w - 4], vb = w < 2 ? v0[2 * w + 1] : v1[2 * w - 3];
            t = fmaf(va, __builtin_bit_cast(float, h[w] << 16) + __builtin_bit_cast(float, l[w] << 16), t);
            t = fmaf(vb, __builtin_bit_cast(float, h[w] & 0xffff0000u) + __builtin_bit_cast(float, l[w] & 0xffff0000u), t);
        }
        return t;
    };
    auto ride = [&](int k, int sst, int buf, const u32x4 (&hi_a)[XN], const u32x4 (&hi_b)[XK]) {     // lo planes: al, bl
        if (TRABL & 1) return;
        if ((2 * k + sst) % Qc == qc) {
#pragma unroll
            for (int x = 0; x < XN; ++x) csl[x] += usum(hi_a[x], al[x]);
        }
        if (VEC && (2 * k + sst) % Qv == qv) {
            const char* pv = lds + 3 * BUF_BYTES + buf * 1024 + (16 * sst + 8 * lh) * 4;
            const f32x4 v0 = *(const f32x4*)pv, v1 = *(const f32x4*)(pv + 16);
#pragma unroll
            for (int y = 0; y < XK; ++y) vsl[y] += udot(hi_b[y], bl[y], v0, v1);
        }
    };

    __builtin_amdgcn_s_barrier();            // the loaders' prologue barrier: chunk 0 is in buffer 0
    asm volatile("" ::: "memory");
    // a*b ~ ah*bh + al*bh + ah*bl per K-step.  Every group of XN*XK MFMAs has the LDS reads of a LATER group in front of
    // it, and the barrier that hands chunk k's buffer back to the loaders sits in front of the chunk's last group,
    // which runs from registers while the first reads of chunk k+1 are in flight.
    // The 2 (XN + XK) transposing reads of a segment are spread between its XN XK MFMAs (a burst of twelve reads in
    // front of nine MFMAs drained the matrix pipe: +35 % cycles, measured): one MFMA, then up to two reads.
    auto spread = [&]() {
#pragma unroll
        for (int i = 0; i < XN * XK; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
    };
    read_planes(0, 0, 0, ah, bh);
    int buf = 0;
    for (int k = 0; k < nchunks; ++k) {
        const int nbuf = buf == 2 ? 0 : buf + 1;
        __builtin_amdgcn_sched_barrier(0);
        read_planes(buf, 0, 1, al, bl);      // lo(k, 0) under ...
        mma(ah, bh);                         // step 0: hi x hi
        spread();
        __builtin_amdgcn_sched_barrier(0);
        read_planes(buf, 1, 0, nah, nbh);    // hi(k, 1) under ...
        mma(al, bh);                         //         lo x hi
        spread();
        __builtin_amdgcn_sched_barrier(0);
        ride(k, 0, buf, ah, bh);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah, bl);                         //         hi x lo   (the last use of al, bl of step 0 was above / is here)
        __builtin_amdgcn_sched_barrier(0);
        read_planes(buf, 1, 1, al, bl);      // lo(k, 1) under ...
        mma(nah, nbh);                       // step 1: hi x hi
        spread();
        __builtin_amdgcn_sched_barrier(0);
        mma(al, nbh);                        //         lo x hi
        __builtin_amdgcn_sched_barrier(0);
        ride(k, 1, buf, nah, nbh);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every read of chunk k has landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_planes(nbuf, 0, 0, ah, bh);      // hi(k+1, 0) (zeros past the end: never used) under ...
        mma(nah, bl);                        //         hi x lo
        spread();
        buf = nbuf;
    }
    clk_end(clk0, wp.clk);
    float* pt = wp.partial + (((long)split * wp.tiles_n + tn) * wp.tiles_k + tk) * (long)(TN * TK);
#pragma unroll
    for (int x = 0; x < XN; ++x)
#pragma unroll
        for (int y = 0; y < XK; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = wn * 32 * XN + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int jx = wk * 32 * XK + y * 32 + li;
                pt[i * TK + jx] = acc[x][y][r];
            }
    // rider shares: colsum_part[split][qc][tiles_n*TN], vec_part[split][qv][tiles_k*TK]
#pragma unroll
    for (int x = 0; x < XN; ++x) {
        const float t = csl[x] + __shfl_xor(csl[x], 32);
        if (lh == 0) wp.colsum_part[((long)split * Qc + qc) * (wp.tiles_n * TN) + tn * TN + wn * 32 * XN + 32 * x + li] = t;
    }
    if (VEC) {
#pragma unroll
        for (int y = 0; y < XK; ++y) {
            const float t = vsl[y] + __shfl_xor(vsl[y], 32);
            if (lh == 0) wp.vec_part[((long)split * Qv + qv) * (wp.tiles_k * TK) + tk * TK + wk * 32 * XK + 32 * y + li] = t;
        }
    }
}


// dW: thread (e, g) of a 256-thread block adds the splits sp = g, g + G, g + 2G, ... of output element e in that order
// (G = 4 groups x 64 elements, or 16 x 16 for small outputs with hundreds of splits), then the G partial sums are
// combined in LDS in a fixed order: deterministic, coalesced (a wave reads 64 or 16 consecutive floats of one
// partial tile row), and 4-16x the loads in flight of one thread per element.
template <int G>
__device__ __forceinline__ void reduce_dw(const WgradReduceParams& rp, float* sm, unsigned bid, unsigned nb) {
    constexpr int E = 256 / G;
    const long total = (long)rp.n_valid * rp.k_valid;
    const int tid = threadIdx.x, el = tid % E, g = tid / E;
    const long tsz = (long)rp.tn_rows * rp.tk_cols;
    const long sstride = (long)rp.tiles_n * rp.tiles_k * tsz;
    for (long e0 = (long)bid * E; e0 < total; e0 += (long)nb * E) {
        const long e = e0 + el;
        float acc = 0.0f;
        int n = 0, k = 0;
        if (e < total) {
            n = (int)(e / rp.k_valid); k = (int)(e % rp.k_valid);
            const int tn = n / rp.tn_rows, i = n % rp.tn_rows, tk = k / rp.tk_cols, j = k % rp.tk_cols;
            const float* src = rp.partial + ((long)tn * rp.tiles_k + tk) * tsz + i * rp.tk_cols + j;
            int sp = g;
            for (; sp + 3 * G < rp.splits; sp += 4 * G) {
                const float v0 = __builtin_nontemporal_load(src + sp * sstride);
                const float v1 = __builtin_nontemporal_load(src + (sp + G) * sstride);
                const float v2 = __builtin_nontemporal_load(src + (sp + 2 * G) * sstride);
                const float v3 = __builtin_nontemporal_load(src + (sp + 3 * G) * sstride);
                acc += v0; acc += v1; acc += v2; acc += v3;
            }
            for (; sp < rp.splits; sp += G) acc += __builtin_nontemporal_load(src + sp * sstride);
        }
        sm[g * E + el] = acc;
        __syncthreads();
        if (g == 0 && e < total) {
            float t = sm[el];
#pragma unroll
            for (int gg = 1; gg < G; ++gg) t += sm[gg * E + el];
            int col = k;
            if (rp.enc_map == 1) col = enc_channel(k >> 1, k & 1);                               // slot 2 s + h
            else if (rp.enc_map == 2) col = enc_channel(2 * (k >> 2) + (k & 1), (k >> 1) & 1);   // k' = 4 (s>>1) + 2 h + (s&1)
            if (col >= 0) rp.dW[(long)n * rp.ldw + rp.col_off + col] = t;
        }
        __syncthreads();
    }
}

// The same for four consecutive columns per thread (16-byte loads; k_valid and the tile width are multiples of 4 for
// every layer of the full-width network): per element the SAME order of additions as reduce_dw<4>, a quarter of the
// load instructions.  Round 2: the 384^2 reductions took 149 us each (37.7 MB of partials: 250 GB/s) with scalar loads.
__device__ __forceinline__ void reduce_dw_vec4(const WgradReduceParams& rp, f32x4* sm4, unsigned bid, unsigned nb) {
    constexpr int G = 4, E = 64;
    const long total4 = ((long)rp.n_valid * rp.k_valid) / 4;
    const int kq = rp.k_valid / 4;
    const int tid = threadIdx.x, el = tid % E, g = tid / E;
    const long tsz = (long)rp.tn_rows * rp.tk_cols;
    const long sstride = (long)rp.tiles_n * rp.tiles_k * tsz;
    for (long e0 = (long)bid * E; e0 < total4; e0 += (long)nb * E) {
        const long e = e0 + el;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        int n = 0, k = 0;
        if (e < total4) {
            n = (int)(e / kq); k = 4 * (int)(e % kq);
            const int tn = n / rp.tn_rows, i = n % rp.tn_rows, tk = k / rp.tk_cols, j = k % rp.tk_cols;
            const float* src = rp.partial + ((long)tn * rp.tiles_k + tk) * tsz + i * rp.tk_cols + j;
            int sp = g;
            for (; sp + 3 * G < rp.splits; sp += 4 * G) {
                const f32x4 v0 = __builtin_nontemporal_load((const f32x4*)(src + sp * sstride));
                const f32x4 v1 = __builtin_nontemporal_load((const f32x4*)(src + (sp + G) * sstride));
                const f32x4 v2 = __builtin_nontemporal_load((const f32x4*)(src + (sp + 2 * G) * sstride));
                const f32x4 v3 = __builtin_nontemporal_load((const f32x4*)(src + (sp + 3 * G) * sstride));
                acc += v0; acc += v1; acc += v2; acc += v3;
            }
            for (; sp < rp.splits; sp += G) acc += __builtin_nontemporal_load((const f32x4*)(src + sp * sstride));
        }
        sm4[g * E + el] = acc;
        __syncthreads();
        if (g == 0 && e < total4) {
            f32x4 t = sm4[el];
#pragma unroll
            for (int gg = 1; gg < G; ++gg) t += sm4[gg * E + el];
            if (rp.enc_map == 0 && ((rp.ldw | rp.col_off) & 3) == 0 && ((size_t)rp.dW & 15) == 0) {
                *(f32x4*)(rp.dW + (long)n * rp.ldw + rp.col_off + k) = t;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int kk = k + c;
                    int col = kk;
                    if (rp.enc_map == 1) col = enc_channel(kk >> 1, kk & 1);
                    else if (rp.enc_map == 2) col = enc_channel(2 * (kk >> 2) + (kk & 1), (kk >> 1) & 1);
                    if (col >= 0) rp.dW[(long)n * rp.ldw + rp.col_off + col] = t[c];
                }
            }
        }
        __syncthreads();
    }
}

// One reduction, run by blocks bid = 0 .. nb - 1 of 256 threads (a launch of its own, or a block range of the batched launch).
__device__ __forceinline__ void wgrad_reduce_body(const WgradReduceParams& rp, unsigned bid, unsigned nb, f32x4* sm4) {
    float* sm = (float*)sm4;
    const long gid = (long)bid * 256 + threadIdx.x, gsz = (long)nb * 256;
    if (rp.dW) {
        const long total = (long)rp.n_valid * rp.k_valid;
        if (total >= 16384 && (rp.k_valid & 3) == 0 && (rp.tk_cols & 3) == 0) reduce_dw_vec4(rp, sm4, bid, nb);
        else if (total >= 16384) reduce_dw<4>(rp, sm, bid, nb);
        else reduce_dw<16>(rp, sm, bid, nb);
    }
    // Rider sums: a few hundred shares per output, summed by ONE thread in share order (deterministic).  The loads are
    // issued 16 at a time and added in order: a serial chain of 256 dependent-latency loads took ~125 us per 384-row
    // layer -- more than the whole dW reduction above.
    auto ordered_sum = [](const float* src, long stride, int count) {
        float acc = 0.0f;
        int sp = 0;
        for (; sp + 16 <= count; sp += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[(long)(sp + u) * stride];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
        for (; sp < count; ++sp) acc += src[(long)sp * stride];
        return acc;
    };
    if (rp.colsum_out && rp.colsum_ld == 0) {
        // one sum over every image's shares (the splits of image b follow those of b-1): up to ~1000 per output, so a WAVE
        // per output -- lane l adds shares l, l + 64, ... in order, then the 64 lane sums meet in a fixed tree
        const long stride = (long)rp.tiles_n * rp.tn_rows;
        const int count = rp.batch * rp.spi * rp.cs_q, lane = threadIdx.x & 63;
        for (long n = gid >> 6; n < rp.n_valid; n += gsz >> 6) {
            float a = 0.0f;
            for (int sp = lane; sp < count; sp += 64) a += rp.colsum_part[(long)sp * stride + n];
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) a += __shfl_xor(a, sft);
            if (lane == 0) rp.colsum_out[n] = a;
        }
    } else if (rp.colsum_out)
        for (long e = gid; e < (long)rp.batch * rp.n_valid; e += gsz) {
            const int b = (int)(e / rp.n_valid), n = (int)(e % rp.n_valid);
            const long stride = (long)rp.tiles_n * rp.tn_rows;
            rp.colsum_out[(long)b * rp.colsum_ld + n] =
                ordered_sum(rp.colsum_part + (long)b * rp.spi * rp.cs_q * stride + n, stride, rp.spi * rp.cs_q);
        }
    if (rp.vec_out)
        for (long e = gid; e < rp.k_valid; e += gsz)
            rp.vec_out[e] = ordered_sum(rp.vec_part + e, (long)rp.tiles_k * rp.tk_cols, rp.splits * rp.vs_q);
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradReduceParams rp) {
    __shared__ f32x4 sm4[256];
    wgrad_reduce_body(rp, blockIdx.x, gridDim.x, sm4);
}

// Every queued reduction of a WgradDefer in one launch (gnr_wgrad.h): block b belongs to the job j with first[j] <= b <
// first[j + 1] and runs that job exactly as its own launch would have -- same block count, same order of additions.
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradReduceBatch rb) {
    __shared__ f32x4 sm4[256];
    int j = 0;
    while (j + 1 < rb.n && blockIdx.x >= rb.first[j + 1]) ++j;
    const unsigned bid = blockIdx.x - rb.first[j], nb = rb.first[j + 1] - rb.first[j];
    if (rb.kind[j] == 0) wgrad_reduce_body(rb.r[j], bid, nb, sm4);
    else wgrad16_reduce_body(rb.r16[j], bid);
}

void wgrad_defer_flush(WgradDefer* d, hipStream_t st) {
    if (d->batch.n > 0)
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(d->batch.first[d->batch.n]), dim3(256), 0, st, d->batch);
    d->batch.n = 0;
    d->cursor = 0;
}
float* wgrad_defer_take(WgradDefer* d, size_t floats, hipStream_t st) {
    floats = (floats + 63) & ~(size_t)63;
    const size_t gap = CANARY_BYTES / 4;                      // experimental builds (gnr_canary.h): a gap behind every sub-allocation
    if (floats + gap > d->arena_floats) return nullptr;
    if (d->batch.n >= WG_DEFER_MAX || d->cursor + floats + gap > d->arena_floats) wgrad_defer_flush(d, st);
    float* q = d->arena + d->cursor;
    d->cursor += floats + gap;
    if (gap) canary_note_now(q + floats, "the weight-gradient arena (wgrad_defer_take)", d->batch.n, st);
    return q;
}
void wgrad_defer_push(WgradDefer* d, const WgradReduceParams& rp, unsigned blocks) {
    const int j = d->batch.n++;
    d->batch.kind[j] = 0; d->batch.r[j] = rp; d->batch.first[j + 1] = d->batch.first[j] + blocks;
}
void wgrad_defer_push16(WgradDefer* d, const Wgrad16ReduceParams& rp, unsigned blocks) {
    const int j = d->batch.n++;
    d->batch.kind[j] = 1; d->batch.r16[j] = rp; d->batch.first[j + 1] = d->batch.first[j] + blocks;
}

constexpr int WG_MAX_BLOCKS = 1024;       // splits * tiles bound: ~2 rounds of 2 workgroups per CU
constexpr int WG_MAX_TILE = WG_TN * WG_TK;    // floats; every tile configuration stays below it
constexpr int WG_RIDER_ROWS = 192;            // largest tile edge

size_t wgrad_scratch_floats() {
    return (size_t)WG_MAX_BLOCKS * WG_MAX_TILE + 2 * (size_t)WG_MAX_BLOCKS * WG_RIDER_ROWS;
}
// Scratch for the queued reductions of one weight set / one upsampler backward (gnr_wgrad.h): a 384 x 384 layer's GEMM takes
// 256 tiles of 192 x 192 + its rider shares = 9.8 M floats, so four single-GEMM scratches (152 M floats, 610 MB) hold all 13-14
// GEMMs of a weight set at once; anything beyond is handled by an early flush.
size_t wgrad_arena_floats(int batch, int max_m, int max_k) {
    const size_t base = 4 * wgrad_scratch_floats();
    // one partial tile set per image at least (spi >= 1): padded output area x images, + the rider shares (<= 4 x 192 + 192 floats
    // per workgroup of >= 32 x 64 outputs: < 1/2 of the tiles) + the rounding of the grid to eight
    const size_t area = (size_t)(max_m + 191) * (size_t)(max_k + 191);
    const size_t big = (size_t)batch * area * 3 / 2 + (size_t)16 * 192 * 192;
    return big > base ? big : base;
}

// fp32 tile configurations: {rows, cols, relative cost per MFMA slot (LDS reads per MFMA, 3-wave workgroups)}
struct TileCfg { int tn, tk; float cost; };
// (the last three: the upsampler's narrow high-resolution layers -- 32 x 64 and 128 x 64 channels over 1.8 M / 0.46 M pixels --
// are HBM-bound; a tile no larger than the product stages no duplicate rows)
constexpr int N_TILE_CFGS = 7;
static const TileCfg kTileCfgs[N_TILE_CFGS] = {{128, 128, 1.00f}, {192, 64, 1.04f}, {64, 192, 1.04f}, {96, 96, 1.08f},
                                               {128, 64, 1.10f}, {64, 64, 1.15f}, {32, 64, 1.20f}};

static int choose_tile(int n_valid, int k_valid, bool vec) {
    if (vec) return 0;                                  // the rider variant exists for the 128 x 128 tile only
    int best = 0;
    float best_cost = 0.0f;
    for (int i = 0; i < N_TILE_CFGS; ++i) {
        const TileCfg& c = kTileCfgs[i];
        const float cost = (float)((n_valid + c.tn - 1) / c.tn) * (float)((k_valid + c.tk - 1) / c.tk) * c.tn * c.tk * c.cost;
        if (i == 0 || cost < best_cost) { best = i; best_cost = cost; }
    }
    return best;
}

// Host-only plan of one weight-gradient GEMM (split from the launch so that the scratch a GEMM really needs --
// partial tiles + rider shares -- can be computed where workspaces are SIZED, gnr_workspace_bytes / gnr_upsample_workspace_bytes,
// and checked there against wgrad_arena_floats() for every batch: tests/test_host_logic.py sweeps 1..512 images on the CPU).
struct WgradPlan {
    int pipe_xk, xa, xb, cfg, TN, TK, tiles_n, tiles_k, linear_map, splits, cs_q, vs_q;
    bool half_rows, img2w, two_wave, unsupported;
    long spi, chunks_per_split;
    unsigned blocks;
    size_t need, cs_need, vec_need;
};
static WgradPlan wgrad_plan(int lda, int n_valid, int ldb, int k_valid, int batch, long chunks_per_image, long pixels_per_image,
                            bool with_vec, bool bf16x3, bool small_tiles) {
    WgradPlan pl{};
    // chunk-channel-major fp32 operands of the MLP's shapes go to the pipelined one-workgroup-per-CU kernel
    // (192-row tiles; K = 64 for the encoding columns); everything else to the two-workgroups-per-CU kernel
    int pipe_xk = 0;
    if (!bf16x3 && pixels_per_image == 0 && n_valid <= 384 && (k_valid == 64 || k_valid == 192 || k_valid == 384) &&
        (!with_vec || (n_valid > 192 && k_valid == 384)))        // the density rider needs the 2 x 2 tile grid
        pipe_xk = k_valid == 64 ? 1 : 3;
    // small_tiles: a product far below the 192-row tile (the 66 rows RGB_layer_2 has beyond its first 192) takes a 96 x 192 tile
    // of the two-wave kernel (round 5; until then a 96 x 96 tile of wgrad_kernel) instead of a 192 x 192 tile that would be
    // two-thirds padding; other shapes keep wgrad_kernel's per-shape tiles
    const bool half_rows = small_tiles && !bf16x3 && !with_vec && pipe_xk == 3 && n_valid <= 96 && k_valid == 192;
    if (small_tiles && !bf16x3 && !with_vec && !half_rows) pipe_xk = 0;
    // bf16x3: pre-split QHL dumps of the chain kernels -> the transposing-read kernel (192-row tiles)
    if (bf16x3 && pixels_per_image == 0 && n_valid <= 384 && lda % 32 == 0 && ldb % 32 == 0 &&
        (k_valid == 64 || k_valid == 192 || k_valid == 384) && (!with_vec || (n_valid > 192 && k_valid == 384)))
        pipe_xk = k_valid == 64 ? 1 : 3;
    if (bf16x3 && !pipe_xk) { pl.unsupported = true; return pl; }      // (the launch reports it)
    // channels-first images (the upsampler's 1x1 convolutions): the same pipelined kernel through its image addressing
    // when the 192 x 192 tiles are reasonably full (measured against wgrad_kernel's per-shape tiles: DESIGN.md 3.5)
    bool img2w = false;
    if (!bf16x3 && pixels_per_image > 0 && !with_vec && pixels_per_image % CHUNK == 0) {
        const int tn = (n_valid + 191) / 192, tk = (k_valid + 191) / 192;
        const double fill = (double)n_valid * k_valid / ((double)tn * tk * 192.0 * 192.0);
        img2w = fill >= 0.45 && tk <= 3;
    }
    if (img2w) pipe_xk = 3;
    // the two-wave kernel's tile: (16-row tiles per wave, 16-column tiles per wave) -> (32 xa) x (64 xb) per workgroup
    int xa = half_rows ? 3 : 6, xb = pipe_xk;
    // Round 5: three of the upsampler's narrow products on instances of the two-wave kernel sized to them (ring of four: their
    // chunk periods are 0.9-1.7 us) instead of wgrad_kernel's 32x32x2 tiles, two workgroups per CU
    if (!(PABL & 8) && !img2w && !bf16x3 && pixels_per_image > 0 && !with_vec && pixels_per_image % CHUNK == 0) {
        if (n_valid % 128 == 0 && k_valid == 128) { xa = 4; xb = 2; }                                  // 256 x 128 (layer_2 at 64 channels)
        else if (n_valid == 128 && k_valid == 64) { xa = 4; xb = 1; }                                  // layer_1 at 64 channels
        else if (n_valid > 32 && n_valid <= 64 && k_valid > 64 && k_valid <= 192) { xa = 2; xb = 3; }  // 64 x 129 (feat_layers)
        if (xb) { img2w = true; pipe_xk = 3; }
    }
    const bool two_wave = !bf16x3 && pipe_xk != 0;        // (K = 64 never comes with the density rider: see above)
    const int cfg = pipe_xk ? 0 : choose_tile(n_valid, k_valid, with_vec);
    const int TN = two_wave ? 32 * xa : (pipe_xk ? 192 : kTileCfgs[cfg].tn), TK = two_wave ? 64 * xb : (pipe_xk ? 64 * pipe_xk : kTileCfgs[cfg].tk);
    pl.tiles_n = (n_valid + TN - 1) / TN;
    pl.tiles_k = (k_valid + TK - 1) / TK;
    const int tiles = pl.tiles_n * pl.tiles_k;
    // Split count: the kernel places split s on XCD s % 8 (32 CUs x 2 resident workgroups = 64 slots
    // per XCD; the pipelined kernel runs one workgroup per CU: 32 slots).  Every XCD must get the SAME number of
    // workgroups and fill whole rounds, otherwise the launch waits for one XCD's straggler round (113 splits instead
    // of 112 cost 40 %): splits = 8 * floor(slots / tiles), made divisible by the batch: ONE round of workgroups.
    // (Two rounds ran the GEMM no faster and doubled the partial tiles the reduce kernel has to sum.)
    // (the 128-thread 32 x 64 tile: four workgroups per CU)
    long splits_total = 8L * ((pipe_xk ? 32 : (cfg == 6 ? 128 : 64)) / tiles);
    if (splits_total < 8) splits_total = 8;
    long spi = splits_total / batch;
    if (img2w) {
        // (split, tile) pairs are spread over the XCDs as one contiguous range each (linear_map): any count that fills
        // the 256 one-workgroup CUs once
        pl.linear_map = 1;
        spi = 256 / ((long)batch * tiles);
    }
    if (spi < 1) spi = 1;
    if (spi > chunks_per_image) spi = chunks_per_image;
    while ((long)batch * spi * tiles > WG_MAX_BLOCKS && spi > 1) --spi;
    pl.chunks_per_split = (chunks_per_image + spi - 1) / spi;
    const int splits = batch * (int)spi;
    const unsigned blocks = img2w ? (unsigned)(8 * (((long)splits * tiles + 7) / 8)) : (unsigned)(8 * ((splits + 7) / 8) * tiles);
    // rider shares per split: the two-wave kernel 4 tiles_k column-sum shares, wgrad3_tr_kernel 2 tiles_k / 2 tiles_n, wgrad_kernel 1
    const int cs_q = two_wave ? 4 * pl.tiles_k : (pipe_xk ? 2 * pl.tiles_k : 1);
    const int vs_q = two_wave ? 1 : (pipe_xk ? 2 * pl.tiles_n : 1);
    // scratch of this GEMM: [partial tiles][column-sum shares][vector shares], each at its exact size (round 5: the shares had a
    // fixed 1024 x 192 floats each, sized for ONE round of workgroups -- more images than workgroup slots, e.g. 40 stacked maps
    // through the upsampler's 18-tile product or > 64 images through the MLP, wrote past it into the next GEMM's partial tiles:
    // wrong bias gradients, tests/test_upsample.py::test_hip_vs_oracle_live[258-16-32-32-40]).  Queued (gnr_wgrad.h) the GEMM takes
    // what it needs from the caller's arena; alone it owns `scratch` (wgrad_scratch_floats()) and must fit it.
    const size_t need = ((size_t)blocks * TN * TK + 63) & ~(size_t)63;
    const size_t cs_need = ((size_t)splits * cs_q * pl.tiles_n * TN + 63) & ~(size_t)63;
    const size_t vec_need = with_vec ? (((size_t)splits * vs_q * pl.tiles_k * TK + 63) & ~(size_t)63) : 0;
    pl.pipe_xk = pipe_xk; pl.half_rows = half_rows; pl.img2w = img2w; pl.xa = xa; pl.xb = xb; pl.two_wave = two_wave; pl.cfg = cfg;
    pl.TN = TN; pl.TK = TK; pl.spi = spi; pl.splits = splits; pl.blocks = blocks; pl.cs_q = cs_q; pl.vs_q = vs_q;
    pl.need = need; pl.cs_need = cs_need; pl.vec_need = vec_need;
    return pl;
}

size_t wgrad_need_floats(const WgradShape& g, int batch) {
    const WgradPlan pl = wgrad_plan(g.lda, g.n_valid, g.ldb, g.k_valid, batch, g.chunks_per_image, g.pixels_per_image, g.with_vec != 0,
                                    g.bf16x3 != 0, g.small_tiles != 0);
    // (+ the experimental build's gaps: behind each of the three sub-blocks and behind the sub-allocation itself)
    return pl.unsupported ? 0 : pl.need + pl.cs_need + pl.vec_need + 4 * (CANARY_BYTES / 4) + 64;
}
size_t wgrad_arena_floats_for(const WgradShape* gemms, int n, int batch, int max_m, int max_k) {
    size_t a = wgrad_arena_floats(batch, max_m, max_k);
    for (int i = 0; i < n; ++i) {
        const size_t need = wgrad_need_floats(gemms[i], batch);
        if (need > a) a = need;
    }
    return a;
}

// dW[n_valid x k_valid] (+ col_off, optional encoding-slot map) = A^T B over all chunks.
// Optional: colsum_out[b][n] = per-image column sums of A; vec_out[k] = vec^T B.
// n_crop x k_crop (<= n_valid x k_valid): the part of the product that is written out -- a network narrower than the
// kernels' 384 channels has zero channels beyond its width in the dumps; the GEMM runs on the kernels' shapes, the
// reduction writes the network's.
static void launch_wgrad_impl(const float* A, int lda, int n_valid, const float* B, int ldb, int k_valid, int batch,
                              long chunks_per_image, float* dW, int ldw, int col_off, int enc_map, float* colsum_out,
                              int colsum_ld, const float* vec, float* vec_out, float* scratch, hipStream_t stream,
                              bool bf16x3, long pixels_per_image, int n_crop, int k_crop, bool small_tiles = false,
                              WgradDefer* defer = nullptr) {
    WgradParams wp{};
    wp.A = A; wp.B = B; wp.lda = lda; wp.ldb = ldb; wp.n_valid = n_valid; wp.k_valid = k_valid;
    if (pixels_per_image > 0) {       // channels-first images [B][C][P]
        wp.a_row = pixels_per_image; wp.a_chunk = CHUNK; wp.a_img = (long)lda * pixels_per_image;
        wp.b_row = pixels_per_image; wp.b_chunk = CHUNK; wp.b_img = (long)ldb * pixels_per_image;
    } else {                          // chunk-channel-major dumps
        wp.a_row = CHUNK; wp.a_chunk = (long)CHUNK * lda; wp.a_img = chunks_per_image * wp.a_chunk;
        wp.b_row = CHUNK; wp.b_chunk = (long)CHUNK * ldb; wp.b_img = chunks_per_image * wp.b_chunk;
    }
    // fp32 chain dumps are S16 (gnr_chain16.h); the encoding dump (enc_map != 0) keeps rows of 32 samples
    wp.a_s16 = (!bf16x3 && pixels_per_image == 0) ? 1 : 0;
    wp.b_s16 = (wp.a_s16 && enc_map == 0) ? 1 : 0;
    wp.clk = clock_probe_slot(GNR_STAGE_WGRAD);
    const bool with_vec = vec_out != nullptr;
    const WgradPlan pl = wgrad_plan(lda, n_valid, ldb, k_valid, batch, chunks_per_image, pixels_per_image, with_vec, bf16x3, small_tiles);
    if (pl.unsupported) {
        // QHL dumps have no other reader; gnr_api.hip admits only the reference's layer widths, so this is a bug
        fprintf(stderr, "gnr: bf16x3 weight gradient asked for an unsupported shape (%d x %d, ld %d / %d)\n", n_valid, k_valid, lda, ldb);
        abort();
    }
    const int pipe_xk = pl.pipe_xk, xa = pl.xa, xb = pl.xb, cfg = pl.cfg, TN = pl.TN, TK = pl.TK, splits = pl.splits, cs_q = pl.cs_q, vs_q = pl.vs_q;
    const bool half_rows = pl.half_rows, two_wave = pl.two_wave;
    const long spi = pl.spi;
    const unsigned blocks = pl.blocks;
    const size_t need = pl.need, cs_need = pl.cs_need, vec_need = pl.vec_need;
    wp.tiles_n = pl.tiles_n; wp.tiles_k = pl.tiles_k; wp.linear_map = pl.linear_map;
    wp.batch = batch; wp.spi = (int)spi;
    wp.chunks_per_image = chunks_per_image;
    wp.chunks_per_split = pl.chunks_per_split;
    WgradDefer* const owner = defer;
    const size_t gap = CANARY_BYTES / 4;          // experimental builds: [partial tiles] gap [column-sum shares] gap [vector shares] gap
    if (defer) {
        float* q = wgrad_defer_take(defer, need + cs_need + vec_need + 3 * gap, stream);
        if (q) scratch = q;
        else {                                // an arena below one GEMM's need (callers size it with wgrad_arena_floats()) --
            wgrad_defer_flush(defer, stream); // run what is queued, then this GEMM owns `scratch` like a launch of its own
            defer = nullptr;
        }
    }
    if (!defer && need + cs_need + vec_need + 3 * gap > wgrad_scratch_floats()) {
        // continuing would write past the caller's scratch.  With a queue the GEMM is skipped and the queue's owner (gnr_bwd /
        // gnr_upsample_bwd) returns an error; arenas sized by wgrad_arena_floats(batch, ...) never get here
        if (owner) { owner->failed = true; return; }
        fprintf(stderr, "gnr: weight gradient %d x %d over %d image(s): %zu floats of split-K scratch needed, %zu available\n", n_valid, k_valid,
                batch, need + cs_need + vec_need, wgrad_scratch_floats());
        abort();
    }
    wp.partial = scratch;
    float* cs_part = scratch + need + gap;
    float* vec_part = cs_part + cs_need + gap;
    float* const cs_gap = cs_part + cs_need;          // (taken before the self-test below moves cs_part)
#if defined(GNR_CANARY) && GNR_CANARY == 2
    // the harness's own self-test (tools/session.sh <name> canary): the column-sum shares start 64 floats late, so their last 64
    // floats land in the gap behind them -- the round-5 overrun in miniature; every call with a bias gradient must then FAIL
    if (colsum_out) cs_part += 64;
#endif
    if (gap) {      // (the round-5 overrun: the column-sum shares of one GEMM running into what lay behind them)
        canary_note_now(scratch + need, "a weight-gradient GEMM's scratch: partial tiles", 0, stream);
        canary_note_now(cs_gap, "a weight-gradient GEMM's scratch: column-sum shares", 1, stream);
        canary_note_now(vec_part + vec_need, "a weight-gradient GEMM's scratch: vector shares", 2, stream);
    }
    wp.colsum_part = cs_part;
    wp.vec = with_vec ? vec : nullptr;
    wp.vec_part = vec_part;
    if (pipe_xk && bf16x3) {
        if (pipe_xk == 1) hipLaunchKernelGGL((wgrad3_tr_kernel<1, false>), dim3(blocks), dim3(512), 0, stream, wp);
        else if (wp.vec) hipLaunchKernelGGL((wgrad3_tr_kernel<3, true>), dim3(blocks), dim3(512), 0, stream, wp);
        else hipLaunchKernelGGL((wgrad3_tr_kernel<3, false>), dim3(blocks), dim3(512), 0, stream, wp);
    } else if (pipe_xk && two_wave) {
        // eight waves (two per SIMD) of 16x16x4 MFMAs: 192 x 192 tiles; round 5: 192 x 64 (encoding columns), 96 x 192 (half_rows)
        if (xa == 6 && xb == 1) hipLaunchKernelGGL((wgrad2w_kernel<false, true, 6, 1, 4>), dim3(blocks), dim3(512), 0, stream, wp);
        else if (xa == 4 && xb == 2) hipLaunchKernelGGL((wgrad2w_kernel<false, true, 4, 2, 4>), dim3(blocks), dim3(512), 0, stream, wp);
        else if (xa == 4 && xb == 1) hipLaunchKernelGGL((wgrad2w_kernel<false, true, 4, 1, 4>), dim3(blocks), dim3(512), 0, stream, wp);
        else if (xa == 2 && xb == 3) hipLaunchKernelGGL((wgrad2w_kernel<false, true, 2, 3, 4>), dim3(blocks), dim3(512), 0, stream, wp);
        else if (half_rows) hipLaunchKernelGGL((wgrad2w_kernel<false, true, 3>), dim3(blocks), dim3(512), 0, stream, wp);
        else if (wp.vec) hipLaunchKernelGGL((wgrad2w_kernel<true, false>), dim3(blocks), dim3(512), 0, stream, wp);
        else if (wp.tiles_k >= 2) hipLaunchKernelGGL((wgrad2w_kernel<false, false>), dim3(blocks), dim3(512), 0, stream, wp);
        else hipLaunchKernelGGL((wgrad2w_kernel<false, true>), dim3(blocks), dim3(512), 0, stream, wp);
    } else if (wp.vec) {
        hipLaunchKernelGGL((wgrad_kernel<2, 2, 2, 2, true>), dim3(blocks), dim3(256), 0, stream, wp);
    } else {
        switch (cfg) {
            case 0: hipLaunchKernelGGL((wgrad_kernel<2, 2, 2, 2, false>), dim3(blocks), dim3(256), 0, stream, wp); break;
            case 1: hipLaunchKernelGGL((wgrad_kernel<2, 2, 3, 1, false>), dim3(blocks), dim3(256), 0, stream, wp); break;
            case 2: hipLaunchKernelGGL((wgrad_kernel<2, 2, 1, 3, false>), dim3(blocks), dim3(256), 0, stream, wp); break;
            case 3: hipLaunchKernelGGL((wgrad_kernel<3, 1, 1, 3, false>), dim3(blocks), dim3(192), 0, stream, wp); break;
            case 4: hipLaunchKernelGGL((wgrad_kernel<2, 2, 2, 1, false>), dim3(blocks), dim3(256), 0, stream, wp); break;
            case 5: hipLaunchKernelGGL((wgrad_kernel<2, 2, 1, 1, false>), dim3(blocks), dim3(256), 0, stream, wp); break;
            default: hipLaunchKernelGGL((wgrad_kernel<1, 2, 1, 1, false>), dim3(blocks), dim3(128), 0, stream, wp); break;
        }
    }
    WgradReduceParams rp{};
    rp.partial = scratch; rp.splits = splits; rp.tiles_n = wp.tiles_n; rp.tiles_k = wp.tiles_k;
    rp.tn_rows = TN; rp.tk_cols = TK;
    rp.cs_q = cs_q; rp.vs_q = vs_q;
    rp.n_valid = n_crop; rp.k_valid = k_crop; rp.dW = dW; rp.ldw = ldw; rp.col_off = col_off; rp.enc_map = enc_map;
    rp.colsum_part = cs_part; rp.colsum_out = colsum_out; rp.colsum_ld = colsum_ld; rp.batch = batch; rp.spi = (int)spi;
    rp.vec_part = vec_part; rp.vec_out = vec_out ? vec_out : nullptr;
    const long total = (long)n_crop * k_crop;
    const bool vec4 = total >= 16384 && (k_crop & 3) == 0 && (TK & 3) == 0;
    const long per_block = vec4 ? 256 : (total >= 16384 ? 64 : 16);   // elements per block, see reduce_dw / reduce_dw_vec4
    long rblocks = (total + per_block - 1) / per_block;
    if (rblocks < 8) rblocks = 8;                               // the rider sums below run grid-stride too
    if (defer) wgrad_defer_push(defer, rp, (unsigned)rblocks);
    else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rblocks), dim3(256), 0, stream, rp);
}

void launch_wgrad(const float* A, int lda, int n_valid, const float* B, int ldb, int k_valid, int batch,
                  long chunks_per_image, float* dW, int ldw, int col_off, int enc_map, float* colsum_out,
                  int colsum_ld, const float* vec, float* vec_out, float* scratch, hipStream_t stream, bool bf16x3,
                  int n_crop = -1, int k_crop = -1, bool small_tiles = false, WgradDefer* defer = nullptr) {
    launch_wgrad_impl(A, lda, n_valid, B, ldb, k_valid, batch, chunks_per_image, dW, ldw, col_off, enc_map, colsum_out,
                      colsum_ld, vec, vec_out, scratch, stream, bf16x3, 0, n_crop < 0 ? n_valid : n_crop,
                      k_crop < 0 ? k_valid : k_crop, small_tiles, defer);
}

// dW[n_valid x k_valid] = sum over images and pixels of A[b][n][p] * B[b][k][p] for channels-first fp32 images
// ([B][lda][P] and [B][ldb][P], P % 32 == 0); colsum_out[b * colsum_ld + n] = sum_p A[b][n][p], or with colsum_ld == 0
// colsum_out[n] = the sum over the images too.  Exact fp32 MFMA.
bool launch_wgrad16_img(const float* A, int M, const float* B, int K, int batch, long P, float* dW, int ldw, float* bias_out,
                        float* scratch, size_t scratch_floats, hipStream_t st, WgradDefer* defer);

void launch_wgrad_img(const float* A, int lda, int n_valid, const float* B, int ldb, int k_valid, int batch,
                      long pixels_per_image, float* dW, int ldw, float* colsum_out, int colsum_ld, float* scratch,
                      hipStream_t stream, WgradDefer* defer) {
    if (!dW && !colsum_out) return;            // include/gnr.h: a NULL gradient pointer == not wanted -- nothing to compute
    // round 4: the wide products (both sides >= 100 channels) go to the register-fed kernel with 16-granular tiles
    // (gnr_wgrad16.hip); the narrow high-resolution ones stay with the LDS-staged kernels below
    if (lda == n_valid && ldb == k_valid && (colsum_ld == 0 || !colsum_out) &&
        launch_wgrad16_img(A, n_valid, B, k_valid, batch, pixels_per_image, dW, ldw, colsum_out, scratch, wgrad_scratch_floats(), stream, defer))
        return;
    launch_wgrad_impl(A, lda, n_valid, B, ldb, k_valid, batch, pixels_per_image / CHUNK, dW, ldw, 0, 0, colsum_out,
                      colsum_ld, nullptr, nullptr, scratch, stream, false, pixels_per_image, n_valid, k_valid, false, defer);
}

// per-image sum of a per-sample vector: out[b * out_stride] = sum_{s in image b} v[s]   (density bias gradient); one workgroup per
// image, fixed order (round 5: one launch for all images; until then one launch per image)
__global__ __launch_bounds__(256) void vecsum_kernel(const float* __restrict__ v, long per_image, float* __restrict__ out, int out_stride) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float acc = 0.0f;
    for (long i = tid; i < per_image; i += 256) acc += v[(long)b * per_image + i];
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) out[(long)b * out_stride] = red[0];
}

void launch_vecsum(const float* v, int batch, long per_image, float* out, int out_stride, hipStream_t stream) {
    hipLaunchKernelGGL(vecsum_kernel, dim3((unsigned)batch), dim3(256), 0, stream, v, per_image, out, out_stride);
}

}  // namespace gnr
