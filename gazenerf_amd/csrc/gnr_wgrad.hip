// gnr_wgrad.hip -- weight gradients of the per-sample FC layers (gfx950).
//
//   dW[n][k] = sum_s dY[s][n] * X[s][k]      (s over all M samples of the call)
//
// dY and X are the chunk-channel-major ("CCM") dumps of the dgrad chain / training forward:
// element (chunk c, channel n, sample j) at c*32*C + n*32 + j, so the [128 channels][32 samples]
// operand tile of one chunk is ONE contiguous 16 KiB block: 256 threads stream it with four fully
// coalesced float4 loads each.  The contraction index is the sample; its order is free, so lane-half
// h of v_mfma_f32_32x32x2_f32 takes samples 16h..16h+15 of the chunk and a lane reads its operand
// values for 4 consecutive MFMA steps with one ds_read_b128.  The LDS image keeps the 128-byte rows
// but XORs the 16-byte piece index with (row/2)%8, which makes both the ds_write_b128 of the staging
// pass and the ds_read_b128 of the MFMA pass bank-conflict free (MI355X LDS: 64 banks for b128,
// non-contiguous 16-lane groups).
//
// Tiling: 128(n) x 128(k) per 256-thread workgroup = 2x2 waves of 64x64 (2x2 MFMA tiles, 64
// accumulator registers), chunks double-buffered through 64 KiB of LDS (next chunk's global loads
// are in flight during the current chunk's 64 MFMAs per wave), two workgroups per CU.  The sample
// range is split (per image) over workgroups; the 3x3 tiles of one split are placed on one XCD
// back-to-back so the operand rows they share hit that XCD's L2.  Per-split partial tiles are summed
// in a fixed order by wgrad_reduce_kernel (deterministic; the reference trains with
// cudnn.deterministic, train.py:57).
//
// Riding along on otherwise idle VALU slots:
//   * column sums of dY (bias gradients, per image)            -- waves with tile-k == 0
//   * vec^T X for a per-sample vector (density-head gradient)  -- waves with tile-n == 0
#include "gnr_chain3.h"

namespace gnr {

constexpr int WG_TN = 128, WG_TK = 128;    // workgroup tile: 2x2 waves of 64x64

struct WgradParams {
    const float* A;      // dY, CCM with C = lda
    const float* B;      // X,  CCM with C = ldb
    int lda, ldb;
    int n_valid, k_valid;
    int tiles_n, tiles_k;
    int batch, spi;               // splits per image
    long chunks_per_image;
    long chunks_per_split;
    float* partial;               // [batch*spi][tiles_n][tiles_k][128][128]
    float* colsum_part;           // [batch*spi][tiles_n*128]
    const float* vec;             // [M] or NULL
    float* vec_part;              // [batch*spi][tiles_k*128]
    // operand addressing (floats): element (image b, local chunk lc, channel n, sample j) at
    //   b*img + lc*chunk + n*row + j.  CCM dumps: row = 32, chunk = 32*ld, img = chunks_per_image*32*ld;
    //   channels-first images [B][C][P]: row = P, chunk = 32, img = C*P.
    long a_row, a_chunk, a_img, b_row, b_chunk, b_img;
};

// LDS position (in floats) of 16-byte piece `c` (0..7) of tile row `n` (0..127)
__device__ __forceinline__ int swz(int n, int c) { return n * CHUNK + ((c ^ ((n >> 1) & 7)) << 2); }

template <bool VEC>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams wp) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][WG_TN * CHUNK];     // [buffer][A/B][row*32 + sample]
    // XCD-aware placement: workgroup id -> (xcd, slot); an XCD runs the tiles of one split back-to-back
    const int tiles = wp.tiles_n * wp.tiles_k;
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int split = xcd + 8 * (slot / tiles);
    const int tile = slot % tiles;
    if (split >= wp.batch * wp.spi) return;
    const int tn = tile / wp.tiles_k, tk = tile - tn * wp.tiles_k;

    // Two workgroups share each SIMD's matrix pipe.  With equal priority they advance in lock-step
    // and reach their per-chunk barrier together, idling the pipe; a static priority for the wave
    // in the odd hardware slot (HW_ID.wave_id, scalar) lets one stream MFMAs while the other
    // stages/syncs, and the roles alternate by themselves.
    if (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) __builtin_amdgcn_s_setprio(1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int b = split / wp.spi, sp = split - b * wp.spi;
    const long c0 = (long)b * wp.chunks_per_image + (long)sp * wp.chunks_per_split;
    long c1 = c0 + wp.chunks_per_split;
    const long cmax = (long)(b + 1) * wp.chunks_per_image;
    if (c1 > cmax) c1 = cmax;

    // staging: float4 piece f = tid + 256 q: tile row f/8, piece f%8 (rows clamped into the buffer:
    // rows/cols beyond it only feed outputs that are dropped)
    const float* ga[4];
    const float* gb[4];
    int lpos[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f = tid + 256 * q, r = f >> 3, c = f & 7;
        int n = tn * WG_TN + r; if (n >= wp.lda) n = wp.lda - 1;
        int k = tk * WG_TK + r; if (k >= wp.ldb) k = wp.ldb - 1;
        ga[q] = wp.A + (long)b * wp.a_img + (long)n * wp.a_row + 4 * c - (long)b * wp.chunks_per_image * wp.a_chunk;
        gb[q] = wp.B + (long)b * wp.b_img + (long)k * wp.b_row + 4 * c - (long)b * wp.chunks_per_image * wp.b_chunk;
        lpos[q] = swz(r, c);
    }
    const long strideA = wp.a_chunk, strideB = wp.b_chunk;
    f32x4 ra[4], rb[4];
    auto gload = [&](long c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = *(const f32x4*)(ga[q] + c * strideA);
            rb[q] = *(const f32x4*)(gb[q] + c * strideB);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *(f32x4*)&lds[buf][0][lpos[q]] = ra[q];
            *(f32x4*)&lds[buf][1][lpos[q]] = rb[q];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
    float cs[2] = {0.0f, 0.0f}, vs[2] = {0.0f, 0.0f};
    const float* pv = VEC ? wp.vec + 16 * lh : nullptr;

    // operand read positions of this lane: rows wn*64 + 32x + li (A) / wk*64 + 32y + li (B), pieces 4lh + g
    int apos[2][4], bpos[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            apos[x][g] = swz(wn * 64 + 32 * x + li, 4 * lh + g);
            bpos[x][g] = swz(wk * 64 + 32 * x + li, 4 * lh + g);
        }

    if (c0 < c1) {
        gload(c0);
        lstore(0);
    }
    __syncthreads();
    for (long c = c0; c < c1; ++c) {
        const int buf = (int)((c - c0) & 1);
        if (c + 1 < c1) gload(c + 1);
        f32x4 v4[4];
        if (VEC) {
#pragma unroll
            for (int g = 0; g < 4; ++g) v4[g] = *(const f32x4*)(pv + c * CHUNK + 4 * g);
        }
        // all 16 operand reads of the chunk first (one exposed LDS latency per 64 MFMAs, not four)
        f32x4 av[2][4], bv[2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            av[0][g] = *(const f32x4*)&lds[buf][0][apos[0][g]];
            av[1][g] = *(const f32x4*)&lds[buf][0][apos[1][g]];
            bv[0][g] = *(const f32x4*)&lds[buf][1][bpos[0][g]];
            bv[1][g] = *(const f32x4*)&lds[buf][1][bpos[1][g]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = av[0][g][e], a1 = av[1][g][e], b0 = bv[0][g][e], b1 = bv[1][g][e];
                acc[0][0] = mfma32(a0, b0, acc[0][0]);
                acc[0][1] = mfma32(a0, b1, acc[0][1]);
                acc[1][0] = mfma32(a1, b0, acc[1][0]);
                acc[1][1] = mfma32(a1, b1, acc[1][1]);
                cs[0] += a0;
                cs[1] += a1;
                if (VEC) {
                    vs[0] = fmaf(v4[g][e], b0, vs[0]);
                    vs[1] = fmaf(v4[g][e], b1, vs[1]);
                }
            }
        }
        if (c + 1 < c1) lstore(buf ^ 1);
        __syncthreads();
    }

    float* pt = wp.partial + (((long)split * wp.tiles_n + tn) * wp.tiles_k + tk) * (long)(WG_TN * WG_TK);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = wn * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int jx = wk * 64 + y * 32 + li;
                pt[i * WG_TK + jx] = acc[x][y][r];
            }
    if (tk == 0 && wk == 0) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const float t = cs[x] + __shfl_xor(cs[x], 32);
            if (lh == 0) wp.colsum_part[(long)split * (wp.tiles_n * WG_TN) + tn * WG_TN + wn * 64 + 32 * x + li] = t;
        }
    }
    if (VEC && tn == 0 && wn == 0) {
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const float t = vs[y] + __shfl_xor(vs[y], 32);
            if (lh == 0) wp.vec_part[(long)split * (wp.tiles_k * WG_TK) + tk * WG_TK + wk * 64 + 32 * y + li] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16x3 variant: same tiling, split and reduce; the operands are split hi/lo (3-term product, fp32
// accumulate) while they are staged, so the LDS image is four bf16 planes per buffer
// (A_hi, A_lo, B_hi, B_lo: [128 rows][32 samples] = 8 KiB each) and a chunk costs 24
// v_mfma_f32_32x32x16_bf16 per wave (768 matrix-pipe cycles) instead of 64 fp32 MFMAs (4096).
//
// Operands arrive in the channel-quad layout of the bf16x3 chain kernels (gnr_chain3.h): element
// (chunk, channel n, sample j) at (n>>2)*128 + 4j + (n&3).  The [128 channels][32 samples] tile of a
// chunk is still one contiguous 16 KiB block; thread (g = tid/8, p = tid%8) loads the 64 contiguous
// bytes of channel quad g, samples 4p..4p+3, re-reads its 4x4 register block channel by channel (the
// transpose is free), splits pairs of consecutive samples and writes 8 bytes of hi and of lo per channel.
//
// LDS row = 64 bytes = 4 pieces of 8 samples; piece q of row n sits at piece q ^ ((n >> 2) & 3), which
// keeps the 16 rows of every ds_read_b128 lane group on 16 different 16-byte slots.  A lane's operand
// for K-step ks (16 samples) is piece 2*lh + ks of its row: lane-half lh contracts samples
// 16 lh .. 16 lh + 15 of the chunk, the same free choice of order as in the fp32 kernel.
// The bias / density-head riders are taken from the fp32 values in the staging registers.
// ---------------------------------------------------------------------------------------------
// Rows are additionally swapped in pairs for odd channel quads (n ^ ((n>>2)&1)): the 16 lanes of a
// ds_write_b64 group cover two quads x one channel, and the swap puts those two 64-byte rows in different
// halves of the 128-byte bank window (first version: 33 % of all LDS cycles were bank conflicts, PMC).
__device__ __forceinline__ int swz3(int n, int q) {   // in bf16 elements
    return (n ^ ((n >> 2) & 1)) * 32 + ((q ^ ((n >> 2) & 3)) << 3);
}

template <bool VEC>
__global__ __launch_bounds__(256, 2) void wgrad3_kernel(const WgradParams wp) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][4][WG_TN * CHUNK];   // [buffer][A_hi,A_lo,B_hi,B_lo]
    const int tiles = wp.tiles_n * wp.tiles_k;
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int split = xcd + 8 * (slot / tiles);
    const int tile = slot % tiles;
    if (split >= wp.batch * wp.spi) return;
    const int tn = tile / wp.tiles_k, tk = tile - tn * wp.tiles_k;
    if (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) __builtin_amdgcn_s_setprio(1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int b = split / wp.spi, sp = split - b * wp.spi;
    const long c0 = (long)b * wp.chunks_per_image + (long)sp * wp.chunks_per_split;
    long c1 = c0 + wp.chunks_per_split;
    const long cmax = (long)(b + 1) * wp.chunks_per_image;
    if (c1 > cmax) c1 = cmax;

    // staging: thread owns channel quad g (tile rows 4g..4g+3), samples 4p..4p+3 of both operands
    // (quads clamped into the tensor: rows beyond it only feed outputs that are dropped)
    const int g = tid >> 3, p = tid & 7;
    int qa = tn * (WG_TN / 4) + g; if (qa >= wp.lda / 4) qa = wp.lda / 4 - 1;
    int qb = tk * (WG_TK / 4) + g; if (qb >= wp.ldb / 4) qb = wp.ldb / 4 - 1;
    const float* ga = wp.A + (long)qa * 128 + 16 * p;
    const float* gb = wp.B + (long)qb * 128 + 16 * p;
    int lpos[4];                                  // row 4g + e, samples 4p..4p+3: piece p>>1, half p&1
#pragma unroll
    for (int e = 0; e < 4; ++e) lpos[e] = swz3(4 * g + e, p >> 1) + 4 * (p & 1);
    const long strideA = (long)CHUNK * wp.lda, strideB = (long)CHUNK * wp.ldb;
    f32x4 ra[4], rb[4];                           // [sample 4p + s] = 4 channels
    auto gload = [&](long c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ra[s] = *(const f32x4*)(ga + c * strideA + 4 * s);
            rb[s] = *(const f32x4*)(gb + c * strideB + 4 * s);
        }
    };
    float cs[4] = {0.0f, 0.0f, 0.0f, 0.0f}, vs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 vv;                                     // vec[samples 4p..4p+3]
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto lstore = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u32x2 h, l;
            unsigned hh, ll;
            split_pair(ra[0][e], ra[1][e], hh, ll); h.x = hh; l.x = ll;
            split_pair(ra[2][e], ra[3][e], hh, ll); h.y = hh; l.y = ll;
            *(u32x2*)&lds[buf][0][lpos[e]] = h;
            *(u32x2*)&lds[buf][1][lpos[e]] = l;
            cs[e] += (ra[0][e] + ra[1][e]) + (ra[2][e] + ra[3][e]);
            split_pair(rb[0][e], rb[1][e], hh, ll); h.x = hh; l.x = ll;
            split_pair(rb[2][e], rb[3][e], hh, ll); h.y = hh; l.y = ll;
            *(u32x2*)&lds[buf][2][lpos[e]] = h;
            *(u32x2*)&lds[buf][3][lpos[e]] = l;
            if (VEC) {
                float d = vv.x * rb[0][e];
                d = fmaf(vv.y, rb[1][e], d);
                d = fmaf(vv.z, rb[2][e], d);
                d = fmaf(vv.w, rb[3][e], d);
                vs[e] += d;
            }
        }
    };
    auto vload = [&](long c) {
        if (VEC) vv = *(const f32x4*)(wp.vec + c * CHUNK + 4 * p);
    };

    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[2][2] = {{zero, zero}, {zero, zero}};
    // operand read positions: rows wn*64 + 32x + li (A) / wk*64 + 32y + li (B), K-step ks -> piece 2 lh + ks
    int apos[2][2], bpos[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            apos[x][ks] = swz3(wn * 64 + 32 * x + li, 2 * lh + ks);
            bpos[x][ks] = swz3(wk * 64 + 32 * x + li, 2 * lh + ks);
        }

    if (c0 < c1) {
        gload(c0);
        vload(c0);
        lstore(0);
    }
    __syncthreads();
    for (long c = c0; c < c1; ++c) {
        const int buf = (int)((c - c0) & 1);
        if (c + 1 < c1) { gload(c + 1); vload(c + 1); }
        u32x4 ah[2][2], al[2][2], bh[2][2], bl[2][2];      // [tile][K-step]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                ah[x][ks] = *(const u32x4*)&lds[buf][0][apos[x][ks]];
                al[x][ks] = *(const u32x4*)&lds[buf][1][apos[x][ks]];
                bh[x][ks] = *(const u32x4*)&lds[buf][2][bpos[x][ks]];
                bl[x][ks] = *(const u32x4*)&lds[buf][3][bpos[x][ks]];
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
                        acc[x][y] = mfma_bf(term == 1 ? al[x][ks] : ah[x][ks], term == 2 ? bl[y][ks] : bh[y][ks], acc[x][y]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < c1) lstore(buf ^ 1);
        __syncthreads();
    }

    float* pt = wp.partial + (((long)split * wp.tiles_n + tn) * wp.tiles_k + tk) * (long)(WG_TN * WG_TK);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = wn * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int jx = wk * 64 + y * 32 + li;
                pt[i * WG_TK + jx] = acc[x][y][r];
            }
    // riders: the eight threads p = 0..7 of a quad hold its eight 4-sample pieces
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = cs[e];
        t += __shfl_xor(t, 1);
        t += __shfl_xor(t, 2);
        t += __shfl_xor(t, 4);
        if (tk == 0 && p == 0) wp.colsum_part[(long)split * (wp.tiles_n * WG_TN) + tn * WG_TN + 4 * g + e] = t;
        if (VEC) {
            float u = vs[e];
            u += __shfl_xor(u, 1);
            u += __shfl_xor(u, 2);
            u += __shfl_xor(u, 4);
            if (tn == 0 && p == 0) wp.vec_part[(long)split * (wp.tiles_k * WG_TK) + tk * WG_TK + 4 * g + e] = u;
        }
    }
}

struct WgradReduceParams {
    const float* partial;
    int splits, tiles_n, tiles_k;
    int n_valid, k_valid;
    float* dW;          // destination matrix (NULL: skip)
    int ldw, col_off;
    int enc_map;        // 1: column k is an encoding slot (2*step+h) -> reference channel
    // optional extras
    const float* colsum_part; float* colsum_out; int colsum_ld; int batch, spi;   // out[b][n]
    const float* vec_part; float* vec_out;                                         // out[k]
};

__global__ void wgrad_reduce_kernel(const WgradReduceParams rp) {
    const long total = (long)rp.n_valid * rp.k_valid;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long)gridDim.x * blockDim.x;
    if (rp.dW)
        for (long e = gid; e < total; e += gsz) {
            const int n = (int)(e / rp.k_valid), k = (int)(e % rp.k_valid);
            const int tn = n / WG_TN, i = n % WG_TN, tk = k / WG_TK, j = k % WG_TK;
            // fixed summation order (deterministic); 8 loads in flight per thread instead of a
            // load -> add dependency chain over ~112 splits
            float acc = 0.0f;
            const float* src = rp.partial + ((long)tn * rp.tiles_k + tk) * (long)(WG_TN * WG_TK) + i * WG_TK + j;
            const long sstride = (long)rp.tiles_n * rp.tiles_k * (WG_TN * WG_TK);
            int sp = 0;
            for (; sp + 8 <= rp.splits; sp += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + (sp + u) * sstride);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; sp < rp.splits; ++sp) acc += src[sp * sstride];
            int col = k;
            if (rp.enc_map) {
                col = enc_channel(k >> 1, k & 1);
                if (col < 0) continue;
            }
            rp.dW[(long)n * rp.ldw + rp.col_off + col] = acc;
        }
    if (rp.colsum_out)
        for (long e = gid; e < (long)rp.batch * rp.n_valid; e += gsz) {
            const int b = (int)(e / rp.n_valid), n = (int)(e % rp.n_valid);
            float acc = 0.0f;
            for (int sp = 0; sp < rp.spi; ++sp)
                acc += rp.colsum_part[((long)b * rp.spi + sp) * (rp.tiles_n * WG_TN) + n];
            rp.colsum_out[(long)b * rp.colsum_ld + n] = acc;
        }
    if (rp.vec_out)
        for (long e = gid; e < rp.k_valid; e += gsz) {
            float acc = 0.0f;
            for (int sp = 0; sp < rp.splits; ++sp) acc += rp.vec_part[(long)sp * (rp.tiles_k * WG_TK) + e];
            rp.vec_out[e] = acc;
        }
}

constexpr int WG_MAX_BLOCKS = 1024;       // splits * tiles bound: ~2 rounds of 2 workgroups per CU

size_t wgrad_scratch_floats() {
    return (size_t)WG_MAX_BLOCKS * WG_TN * WG_TK + (size_t)WG_MAX_BLOCKS * WG_TN + (size_t)WG_MAX_BLOCKS * WG_TK;
}

// dW[n_valid x k_valid] (+ col_off, optional encoding-slot map) = A^T B over all chunks.
// Optional: colsum_out[b][n] = per-image column sums of A; vec_out[k] = vec^T B.
static void launch_wgrad_impl(const float* A, int lda, int n_valid, const float* B, int ldb, int k_valid, int batch,
                              long chunks_per_image, float* dW, int ldw, int col_off, int enc_map, float* colsum_out,
                              int colsum_ld, const float* vec, float* vec_out, float* scratch, hipStream_t stream,
                              bool bf16x3, long pixels_per_image) {
    WgradParams wp{};
    wp.A = A; wp.B = B; wp.lda = lda; wp.ldb = ldb; wp.n_valid = n_valid; wp.k_valid = k_valid;
    if (pixels_per_image > 0) {       // channels-first images [B][C][P]
        wp.a_row = pixels_per_image; wp.a_chunk = CHUNK; wp.a_img = (long)lda * pixels_per_image;
        wp.b_row = pixels_per_image; wp.b_chunk = CHUNK; wp.b_img = (long)ldb * pixels_per_image;
    } else {                          // chunk-channel-major dumps
        wp.a_row = CHUNK; wp.a_chunk = (long)CHUNK * lda; wp.a_img = chunks_per_image * wp.a_chunk;
        wp.b_row = CHUNK; wp.b_chunk = (long)CHUNK * ldb; wp.b_img = chunks_per_image * wp.b_chunk;
    }
    wp.tiles_n = (n_valid + WG_TN - 1) / WG_TN;
    wp.tiles_k = (k_valid + WG_TK - 1) / WG_TK;
    const int tiles = wp.tiles_n * wp.tiles_k;
    // Split count: the kernel places split s on XCD s % 8 (32 CUs x 2 resident workgroups = 64 slots
    // per XCD).  Every XCD must get the SAME number of workgroups and fill whole rounds, otherwise
    // the launch waits for one XCD's straggler round (113 splits instead of 112 cost 40 %):
    // splits = 8 * floor(64 slots / tiles), made divisible by the batch: ONE round of workgroups.  (Two rounds
    // ran the GEMM no faster and doubled the partial tiles the reduce kernel has to sum: 60 -> 27 us per layer.)
    long splits_total = 8L * (64 / tiles);
    if (splits_total < 8) splits_total = 8;
    long spi = splits_total / batch;
    if (spi < 1) spi = 1;
    if (spi > chunks_per_image) spi = chunks_per_image;
    while ((long)batch * spi * tiles > WG_MAX_BLOCKS && spi > 1) --spi;
    wp.batch = batch; wp.spi = (int)spi;
    wp.chunks_per_image = chunks_per_image;
    wp.chunks_per_split = (chunks_per_image + spi - 1) / spi;
    wp.partial = scratch;
    float* cs_part = scratch + (size_t)WG_MAX_BLOCKS * WG_TN * WG_TK;
    float* vec_part = cs_part + (size_t)WG_MAX_BLOCKS * WG_TN;
    wp.colsum_part = cs_part;
    wp.vec = vec_out ? vec : nullptr;
    wp.vec_part = vec_part;
    const int splits = batch * (int)spi;
    const unsigned blocks = (unsigned)(8 * ((splits + 7) / 8) * tiles);
    if (bf16x3) {
        if (wp.vec) hipLaunchKernelGGL((wgrad3_kernel<true>), dim3(blocks), dim3(256), 0, stream, wp);
        else hipLaunchKernelGGL((wgrad3_kernel<false>), dim3(blocks), dim3(256), 0, stream, wp);
    } else {
        if (wp.vec) hipLaunchKernelGGL((wgrad_kernel<true>), dim3(blocks), dim3(256), 0, stream, wp);
        else hipLaunchKernelGGL((wgrad_kernel<false>), dim3(blocks), dim3(256), 0, stream, wp);
    }
    WgradReduceParams rp{};
    rp.partial = scratch; rp.splits = splits; rp.tiles_n = wp.tiles_n; rp.tiles_k = wp.tiles_k;
    rp.n_valid = n_valid; rp.k_valid = k_valid; rp.dW = dW; rp.ldw = ldw; rp.col_off = col_off; rp.enc_map = enc_map;
    rp.colsum_part = cs_part; rp.colsum_out = colsum_out; rp.colsum_ld = colsum_ld; rp.batch = batch; rp.spi = (int)spi;
    rp.vec_part = vec_part; rp.vec_out = vec_out ? vec_out : nullptr;
    const long total = (long)n_valid * k_valid;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, rp);
}

void launch_wgrad(const float* A, int lda, int n_valid, const float* B, int ldb, int k_valid, int batch,
                  long chunks_per_image, float* dW, int ldw, int col_off, int enc_map, float* colsum_out,
                  int colsum_ld, const float* vec, float* vec_out, float* scratch, hipStream_t stream, bool bf16x3) {
    launch_wgrad_impl(A, lda, n_valid, B, ldb, k_valid, batch, chunks_per_image, dW, ldw, col_off, enc_map, colsum_out,
                      colsum_ld, vec, vec_out, scratch, stream, bf16x3, 0);
}

// dW[n_valid x k_valid] = sum over images and pixels of A[b][n][p] * B[b][k][p] for channels-first fp32 images
// ([B][lda][P] and [B][ldb][P], P % 32 == 0); colsum_out[b][n] = sum_p A[b][n][p].  Exact fp32 MFMA.
void launch_wgrad_img(const float* A, int lda, int n_valid, const float* B, int ldb, int k_valid, int batch,
                      long pixels_per_image, float* dW, int ldw, float* colsum_out, int colsum_ld, float* scratch,
                      hipStream_t stream) {
    launch_wgrad_impl(A, lda, n_valid, B, ldb, k_valid, batch, pixels_per_image / CHUNK, dW, ldw, 0, 0, colsum_out,
                      colsum_ld, nullptr, nullptr, scratch, stream, false, pixels_per_image);
}

// per-image sum of a per-sample vector: out[b] = sum_{s in image b} v[s]   (density bias gradient)
__global__ __launch_bounds__(256) void vecsum_kernel(const float* __restrict__ v, long per_image, float* __restrict__ out) {
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
    if (tid == 0) out[b] = red[0];
}

void launch_vecsum(const float* v, int batch, long per_image, float* out, int out_stride, hipStream_t stream) {
    // out[b * out_stride]: written through a strided view by launching per image
    for (int b = 0; b < batch; ++b)
        hipLaunchKernelGGL(vecsum_kernel, dim3(1), dim3(256), 0, stream, v + (long)b * per_image, per_image,
                           out + (long)b * out_stride);
}

}  // namespace gnr
