// gnr_wgrad16.hip -- weight gradients of the upsampler's wide 1x1 convolutions, round 4, gfx950.
//
//   dW[n][k] = sum over images and pixels of dY[b][n][p] X[b][k][p],   db[n] = sum dY[b][n][p]
// for channels-first fp32 images (models/pixel_shuffle_upsample.py:19-31, models/neural_renderer.py:60-82 are the
// convolutions whose autograd this is).  Until round 4 these products went through the MLP's wgrad2w_kernel: 192 x 192
// workgroup tiles, built for 384-wide layers -- the upsampler's channel counts are one past a power of two (129, 258, 516,
// 1032), so 516 x 258 filled 60 % of its tiles and 258 x 129 / 129 x 258 45 % (profiles/r3_n1_launches.txt: 0.37-0.53 of the
// fp32-MFMA peak on these five products).  This kernel is conv16_kernel's recipe applied to the contraction over pixels:
// no LDS, no barrier, independent waves, several per SIMD, operands straight from memory into v_mfma_f32_16x16x4_f32 --
//   * a wave owns MT x NT tiles of 16 x 16 outputs over its split's pixels; padding is to multiples of 16 MT / 16 NT
//     ((3,6) and (6,3): 516 = 11 x 48, 1032 = 11 x 96 - 24, 258 + 1 -> 3 x 96, 129 + 1 -> 3 x 48: <= 6 % in every product);
//   * both operands are "rows of pixels": lane (li, g) loads pixels 16 h + 4 g .. + 3 of row li of each of its tiles (one
//     b128 per tile per 16-pixel half step = 4 MFMA steps; the contraction order over the pixels is free), two operand
//     sets, one half step ahead; three waves per SIMD (148 VGPRs).  Measured and dropped: 32-pixel steps with both halves of
//     a 128-byte line requested back to back (three operand sets, 204 VGPRs, two waves per SIMD): 306 us against 272 on
//     516 x 258 -- the second wave slot matters, the line halves do not;
//   * the bias gradient is the last padded column of the product, against a row of ones that is never loaded (the lane that
//     would hold it adds 1.0 to the zeros the descriptor's bound returns);
//   * split-K over pixel ranges inside an image, one round of waves; partial tiles are summed in a fixed order by
//     wgrad16_reduce_kernel (deterministic; the reference trains with cudnn.deterministic, train.py:57).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gnr_device.h"
#include "gnr_wgrad.h"

namespace gnr {

namespace {

struct Wgrad16Params {
    const float* A; const float* B;      // dY [batch][M][P], X [batch][K][P]
    int M, K, P, batch;
    int mg, kg;                          // groups of MT / NT tiles
    int spi, kbs;                        // splits per image, 32-pixel steps per split
    int ones_col;                        // >= 0: the last padded column receives sum_p dY (bias gradient)
    float* partial;                      // [batch * spi][mg * 16 MT][kg * 16 NT]
};

constexpr int W16_WPB = 4;
constexpr int W16_OCC = 3;               // waves per SIMD (148 VGPRs)

__device__ __forceinline__ f32x4 mfma16g(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int MT, int NT>
__global__ __launch_bounds__(64 * W16_WPB, W16_OCC) void wgrad16_kernel(const Wgrad16Params wp) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 15, g = lane >> 4;
    const unsigned tiles = (unsigned)(wp.mg * wp.kg);
    const unsigned witems = (unsigned)(wp.batch * wp.spi) * tiles;
    const unsigned wg_items = (witems + W16_WPB - 1) / W16_WPB;
    // an XCD takes a contiguous range of (split, tile) items: the tiles of a split share its operand rows in that XCD's L2
    const unsigned per_xcd = (wg_items + 7u) >> 3;
    const unsigned item = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || item >= wg_items) return;
    const unsigned wi = item * W16_WPB + (unsigned)wave;
    if (wi >= witems) return;
    const unsigned split = wi / tiles, tile = wi - split * tiles;
    const int mgi = (int)(tile / (unsigned)wp.kg), kgi = (int)(tile - (unsigned)mgi * (unsigned)wp.kg);
    const int b = (int)(split / (unsigned)wp.spi), sp = (int)(split - (unsigned)b * (unsigned)wp.spi);
    const int kb0 = sp * wp.kbs;
    int nkb = wp.P / 32 - kb0;
    if (nkb > wp.kbs) nkb = wp.kbs;                               // <= 0: a split past the image's end writes zeros
    const int m0 = mgi * (16 * MT), k0 = kgi * (16 * NT);

    const long a_rows = wp.M - m0, b_rows = wp.K - k0 > 0 ? wp.K - k0 : 0;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(wp.A + ((long)b * wp.M + m0) * wp.P), 0, (int)(a_rows * wp.P * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(wp.B + ((long)b * wp.K + k0) * wp.P), 0, (int)(b_rows * wp.P * 4), 0x00020000);
    const unsigned row = ((unsigned)li * (unsigned)wp.P + 4u * (unsigned)g) * 4u;       // this lane's row / pixel quad inside a tile
    const unsigned tstride = (unsigned)__builtin_amdgcn_readfirstlane((int)(16u * (unsigned)wp.P * 4u));     // bytes between row tiles

    // the bias column is the LAST column of the padded product (the host pads to at least one column past K): tile NT - 1,
    // lane 15 of the waves of the last column group -- 1.0 is added to the zeros the descriptor's bound returns there
    const float one_last = (wp.ones_col >= 0 && kgi == wp.kg - 1 && li == 15) ? 1.0f : 0.0f;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    f32x4 A1[2][MT], B1[2][NT];
    auto load = [&](int kb, int half, f32x4 (&A)[MT], f32x4 (&Bv)[NT]) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((2 * (kb0 + kb) + half) * 64);
        // the row-tile stride rides in the scalar offset: one address register per lane
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            A[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, row, (int)(so + (unsigned)mt * tstride), 0));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            Bv[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, row, (int)(so + (unsigned)nt * tstride), 0));
    };
    auto compute = [&](const f32x4 (&A)[MT], const f32x4 (&Bv)[NT]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float blast = Bv[NT - 1][s] + one_last;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16g(A[mt][s], nt == NT - 1 ? blast : Bv[nt][s], acc[mt][nt]);
        }
    };
    if (nkb > 0) {                                                // 16-pixel half steps, two operand sets, one half ahead
        load(0, 0, A1[0], B1[0]);
        for (int kb = 0; kb < nkb; ++kb) {
            load(kb, 1, A1[1], B1[1]);
            __builtin_amdgcn_sched_barrier(0);
            compute(A1[0], B1[0]);
            __builtin_amdgcn_sched_barrier(0);
            load(kb + 1 < nkb ? kb + 1 : kb, 0, A1[0], B1[0]);
            __builtin_amdgcn_sched_barrier(0);
            compute(A1[1], B1[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // register e of acc[mt][nt]: row m0 + 16 mt + 4 g + e, column k0 + 16 nt + li
    const long n_pad = (long)wp.mg * (16 * MT), k_pad = (long)wp.kg * (16 * NT);
    float* dst = wp.partial + ((long)split * n_pad + m0 + 4 * g) * k_pad + k0 + li;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) dst[(long)(16 * mt + e) * k_pad + 16 * nt] = acc[mt][nt][e];
}


// dW[n][k] = sum_s partial[s][n][k] (s ascending: fixed order), bias[n] = the last padded column.  One thread per output, k fastest.
__global__ __launch_bounds__(256) void wgrad16_reduce_kernel(const Wgrad16ReduceParams rp) { wgrad16_reduce_body(rp, blockIdx.x); }

}  // namespace

// Returns false when the product is left to launch_wgrad_img's other kernels (narrow or HBM-bound shapes, odd pixel counts,
// not enough scratch).  bias_out: [M] summed over the images, or NULL.
bool launch_wgrad16_img(const float* A, int M, const float* B, int K, int batch, long P, float* dW, int ldw, float* bias_out,
                        float* scratch, size_t scratch_floats, hipStream_t st, WgradDefer* defer) {
    // Only where the 192 x 192 tiles of the LDS-staged kernel are badly filled: measured per 7 images (us, this kernel / the
    // old path) 516 x 258: 272 / 366, 258 x 129: 90 / 128, 129 x 258: 86 / 129, 516 x 258 at 64 x 64: 79 / 101 -- but
    // 1032 x 516 (fill 0.80): 288 / 273-285, and the tile-friendly, HBM-bound 256 x 128 at 256 x 256 (128 x 128 tiles): 315 / 256.
    const double fill192 = (double)M * K / ((double)((M + 191) / 192) * ((K + 191) / 192) * 192.0 * 192.0);
    if (M < 100 || K < 100 || !(M % 64 || K % 64) || fill192 >= 0.7 || P % 32 ||
        (long)M * P * 4 >= (1L << 32) || (long)K * P * 4 >= (1L << 32))        // one 32-bit buffer descriptor per image operand
        return false;
    const int kw = K + (bias_out ? 1 : 0);
    const int mt16 = (M + 15) / 16, kt16 = (kw + 15) / 16;
    // (3,6) or (6,3): the orientation with fewer padded tiles
    const long pad36 = (long)((mt16 + 2) / 3 * 3) * ((kt16 + 5) / 6 * 6), pad63 = (long)((mt16 + 5) / 6 * 6) * ((kt16 + 2) / 3 * 3);
    const bool o36 = pad36 <= pad63;
    const int MT = o36 ? 3 : 6, NT = o36 ? 6 : 3;
    Wgrad16Params wp{};
    wp.A = A; wp.B = B; wp.M = M; wp.K = K; wp.P = (int)P; wp.batch = batch;
    wp.mg = (mt16 + MT - 1) / MT; wp.kg = (kt16 + NT - 1) / NT;
    wp.ones_col = bias_out ? 1 : -1;
    const long tiles = (long)wp.mg * wp.kg, area = (long)wp.mg * 16 * MT * wp.kg * 16 * NT;
    const long kb_img = P / 32;
    // one round of waves: 1024 SIMDs x 2 waves
    long spi = 1024 * W16_OCC / (tiles * batch);
    if (spi < 1) spi = 1;
    if (spi > kb_img / 4) spi = kb_img / 4 > 0 ? kb_img / 4 : 1;                 // >= 128 pixels per split
    if (defer && defer->arena_floats < scratch_floats) scratch_floats = defer->arena_floats;      // queued reduction (gnr_wgrad.h): partials from the caller's arena
    while (spi > 1 && (size_t)(batch * spi * area) > scratch_floats) --spi;
    if ((size_t)(batch * spi * area) > scratch_floats) return false;
    wp.kbs = (int)((kb_img + spi - 1) / spi);
    wp.spi = (int)((kb_img + wp.kbs - 1) / wp.kbs);
    if (defer) {
        scratch = wgrad_defer_take(defer, (size_t)batch * wp.spi * area, st);
        if (!scratch) return false;
    }
    wp.partial = scratch;
    const long witems = (long)batch * wp.spi * tiles, wg_items = (witems + W16_WPB - 1) / W16_WPB;
    const unsigned blocks = (unsigned)(8 * ((wg_items + 7) / 8));
    if (o36) hipLaunchKernelGGL((wgrad16_kernel<3, 6>), dim3(blocks), dim3(64 * W16_WPB), 0, st, wp);
    else hipLaunchKernelGGL((wgrad16_kernel<6, 3>), dim3(blocks), dim3(64 * W16_WPB), 0, st, wp);
    Wgrad16ReduceParams rp{};
    rp.partial = scratch; rp.splits = batch * wp.spi; rp.n_pad = (long)wp.mg * 16 * MT; rp.k_pad = (long)wp.kg * 16 * NT;
    rp.M = M; rp.K = K; rp.dW = dW; rp.ldw = ldw; rp.bias = bias_out;
    const long total = (long)M * kw;
    if (defer) wgrad_defer_push16(defer, rp, (unsigned)((total + 255) / 256));
    else hipLaunchKernelGGL(wgrad16_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, rp);
    return true;
}

}  // namespace gnr
