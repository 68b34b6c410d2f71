// gnr_conv16.hip -- the 1x1 convolutions of the upsampler as register-fed fp32 MFMA GEMMs, gfx950.
//
// C[M][N] = epilogue(A[M][K] B[K][N]) for channels-first images (N = pixels, contiguous): the 1x1 convolutions of
// PixelShuffleUpsample (models/pixel_shuffle_upsample.py:19-42) and NeuralRenderer.feat_layers
// (models/neural_renderer.py:60-113) and their data gradients (A = W^T through strides).
//
// Round 2's conv_gemm_kernel staged both operands through LDS (LDS-DMA, one barrier per 16 k) and ran at 0.45-0.70 of the
// fp32-MFMA peak on these shapes: M is one channel past a multiple of 128 (129, 258, 516, 1032: 10-33 % padded rows),
// K is 64-516 (4-33 k-tiles: prologue, epilogue and barriers are a large share), and at B = 1 most layers launch fewer
// workgroups than the chip has slots.  This kernel applies what round 3 measured for the MLP chain (DESIGN.md section 5,
// tools/ubench/mfma_2w.hip): two or more INDEPENDENT waves per SIMD -- no LDS, no barrier -- each feeding
// v_mfma_f32_16x16x4_f32 straight from registers reach 97 % of the matrix pipe, because one wave's loads and waits issue
// beside the other's MFMAs.  Measured here (profiles/r3_n1_conv16_experiments.txt): the main loops run at 0.85-0.90 of
// the peak; the epilogue's stores do NOT hide under the other waves' MFMAs -- they add their duration -- whatever the tile
// shape, the workgroup size or the phase relation of the waves, which is what is left of the distance to the roof.
//
// A wave owns MT row tiles (16 channels) x NT pixel tiles (16 pixels) over the WHOLE contraction:
//   * A (weights, <= 2 MB, re-laid out once per call by conv16_pack_kernel as [slice][k-block][row tile][lane][4]):
//     one buffer_load_b128 per row tile per 16 k = the lane's A values of 4 MFMA steps; every wave of a row slice
//     streams the same bytes -- L1 / L2 hits.
//   * B (activations): the contraction order is free, so lane group g = l >> 4 takes k = 16 kb + 4 s + g at step s;
//     a lane's load is NT CONSECUTIVE PIXELS of that row (b128 / b64) -- pixel NT*li + t feeds column li of pixel tile
//     t, so one load feeds NT MFMAs and the lane ends up owning NT consecutive pixels of each of its rows: vector
//     stores in the epilogue.  Rows k >= K read zeros through the buffer descriptor's bound.
//   * both operand sets of block kb+1 are requested before the MFMAs of block kb (register double buffer): a whole
//     block of MFMAs (MT*NT*4 x 32 cycles) covers the latency.  vmcnt is in-order, so A cannot run a shorter prefetch
//     distance than B without forcing B's loads home early -- hence 2 x MT x 4 A registers.
//   * MT x NT per layer from a small cost model (padded work x fill of the 1024 SIMDs; conv16_plan): measured, the
//     smallest tile (2 x 4: 84 VGPRs, five waves per SIMD) is at or near the best for every plain GEMM and wins the ties;
//     1032 rows = 5 x 13 tiles exactly and 516 = 3 x 11 are kept for shapes where padding decides.
//   * workgroup id -> (XCD, slot): an XCD takes a contiguous range of (pixel tile, row slice) items, row slice fastest,
//     so the slices of one pixel tile re-read its B rows from the same L2.
// BLUR instances (forward feat_layers): the B operand is blur(u), computed on the fly from three rows of u (reflect
// padding == kornia filter2d border_type='reflect', blur_kernel's taps applied rows-first) -- the blurred map is never
// written; the backward uses blur's adjoint on the (half as wide) gradient instead (gnr_upsample.hip).
#include "gnr_conv16.h"

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gnr_chain16.h"

namespace gnr {

int fail(const char* fmt, ...);

namespace {

#ifndef GNR_C16_ABL
#define GNR_C16_ABL 0       // timing experiments (wrong results; tools/ab_n1.sh): 1 no epilogue, 2 B rows of k-block 0 only, 4 A of k-block 0
                            // only, 8 (blur_lds) no stencil / MFMAs, 16 (blur_lds) no halo loads, 32 epilogue without its loads, 64 epilogue without its stores, 128 stores folded into
                            // a 1 MiB window (L2-resident)
#endif
constexpr int WPB = 4;       // waves per workgroup: they share a row slice (the A stream hits in L1) and take adjacent pixels
constexpr float LEAK16 = 0.2f;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma16c(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int NT> struct Pix;
template <> struct Pix<4> {
    typedef f32x4 T;
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)soff, 0));
    }
};
template <> struct Pix<2> {
    typedef f32x2 T;
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (int)soff, 0));
    }
};
// forward Blur taps at position u of n (reflect padding: the out-of-range neighbour folds onto the inner one)
__device__ __forceinline__ void blur_taps16(int u, int n, float& wl, float& wc, float& wr) {
    wc = 0.5f;
    wl = u >= 1 ? 0.25f : 0.0f;
    wr = u + 1 < n ? 0.25f : 0.0f;
    if (u == 0) wr += 0.25f;          // in[-1] -> in[1]
    if (u == n - 1) wl += 0.25f;      // in[n]  -> in[n-2]
}
__device__ __forceinline__ float load1(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (int)soff, 0));
}

// dst[(((slice*nkb + kb)*MT + mt)*64 + lane)*4 + s] = A(16 (slice MT + mt) + lane%16, 16 kb + 4 s + lane/16), 0 outside
__global__ __launch_bounds__(256) void conv16_pack_kernel(const Conv16PackJobs jobs) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    int j = 0;
    while (j < jobs.n && i >= jobs.j[j].floats) { i -= jobs.j[j].floats; ++j; }
    if (j >= jobs.n) return;
    const Conv16PackJobs::Job& J = jobs.j[j];
    const int s = (int)(i & 3), lane = (int)((i >> 2) & 63);
    long r = i >> 8;
    const int mt = (int)(r % J.MT); r /= J.MT;
    const int kb = (int)(r % J.nkb);
    const int slice = (int)(r / J.nkb);
    const int row = 16 * (slice * J.MT + mt) + (lane & 15);
    const int k = J.kchain ? 16 * kb + 4 * (lane >> 4) + s : 16 * kb + 4 * s + (lane >> 4);
    const int m = J.perm4 ? (row >> 2) + (row & 3) * (J.M >> 2) : row;       // perm4: a lane's four rows are channels cb + q M/4
    jobs.dst[J.dst_off + i] = (row < J.M && k < J.K) ? J.W[(long)m * J.rs + (long)k * J.cs] : 0.0f;
}

// The plain epilogue (bias, LeakyReLU, mask, accumulate, store) with the optional RGB rider: register e of acc[mt][t] is channel
// m0 + 16 mt + 4 g + e at pixel n + t of image b.  Shared by conv16_kernel and conv16_blur_lds_kernel.
// PRE (conv16_blur_lds_kernel, round 5): the epilogue's own operands -- the bias of the lane's MT x 4 channels and the running
// RGB values its rider accumulates into -- were requested before the last k-block's MFMAs (conv16_epilogue_prefetch) and arrive
// as registers: a workgroup whose waves meet at barriers cannot hide two dependent load latencies at its end behind other waves
// (measured: the epilogue without its stores was 54 of that kernel's 241 us).  Same values, same arithmetic.
template <int MT, int NT>
struct Conv16EpiPre {
    static constexpr bool BIAS = MT <= 2;          // four row tiles: the 16 bias registers spill (168-VGPR budget); only the RGB values travel
    float bias[BIAS ? MT : 1][4]; float rgb[(3 * NT + 3) / 4];
};
template <int MT, int NT>
__device__ __forceinline__ void conv16_epilogue_prefetch(const Conv16Params& cp, int b, int n, int m0, int g, Conv16EpiPre<MT, NT>& pre) {
    if constexpr (Conv16EpiPre<MT, NT>::BIAS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + 16 * mt + 4 * g + e;
                pre.bias[mt][e] = (cp.bias && m < cp.M) ? cp.bias[m] : 0.0f;
            }
    }
    constexpr int NV = 3 * NT, NG = (NV + 3) / 4;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int idx = 4 * k + g;
        pre.rgb[k] = 0.0f;
        if (cp.rgb_w && cp.rgb_accumulate && idx < NV) {
            const int o = idx / NT, t = idx - o * NT;
            pre.rgb[k] = cp.rgb[((long)b * 3 + o) * cp.P + n + t];
        }
    }
}

template <int MT, int NT, bool PRE = false>
__device__ __forceinline__ void conv16_plain_epilogue(const Conv16Params& cp, f32x4 (&acc)[MT][NT], int b, int n, int m0, int g,
                                                      const float* rgbw, const Conv16EpiPre<MT, NT>* pre = nullptr) {
    typedef typename Pix<NT>::T pv;
    // the RGB branch on the block output (slices == 1): per lane the dot over its channels, then over the four lane groups
    float ra[3][NT];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int t = 0; t < NT; ++t) ra[o][t] = 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + 16 * mt + 4 * g + e;
            if (m >= cp.M) continue;
            pv v;
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = acc[mt][t][e];
            if constexpr (PRE && Conv16EpiPre<MT, NT>::BIAS) v += pre->bias[mt][e];      // (0 where there is no bias)
            else if (cp.bias && !(GNR_C16_ABL & 32)) v += cp.bias[m];
            if (cp.leaky) {
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] = v[t] > 0.0f ? v[t] : LEAK16 * v[t];
            }
            if (cp.mask_ref && !(GNR_C16_ABL & 32)) {
                const pv mk = *(const pv*)(cp.mask_ref + (long)b * cp.mask_batch + (long)m * cp.P + n);
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] *= mk[t] > 0.0f ? 1.0f : LEAK16;
            }
            float* dst = cp.C + (((GNR_C16_ABL & 128) ? 0x3FFFCL : -1L) & ((long)b * cp.c_batch + (long)m * cp.P + n));
            if (cp.accumulate && !(GNR_C16_ABL & 32)) v += *(const pv*)dst;
            if constexpr (NT == 4) {
                if (cp.dres_from) {
                    // unshuffle_dres_kernel's expression, term for term (un-masking by x 5.0f where the mask multiplied by 0.2f)
                    pv G[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int k = m + q * cp.M;
                        const pv d = *(const pv*)(cp.dres_from + (long)b * cp.dres_from_batch + (long)k * cp.P + n);
                        const unsigned nib4 = *(const unsigned*)(cp.dres_sign + (long)b * cp.dres_sign_batch + (long)(k >> 2) * cp.P + n);
#pragma unroll
                        for (int t = 0; t < NT; ++t) G[q][t] = d[t] * (((nib4 >> (8 * t + (k & 3))) & 1u) ? 1.0f : 1.0f / LEAK16);
                    }
                    v += (G[0] + G[1]) + (G[2] + G[3]);
                }
            }
            if ((GNR_C16_ABL & 64) && v[0] != 1.2345f) continue;
            *(pv*)dst = v;
            if (cp.rgb_w) {
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float wo = rgbw[o * cp.M + m];
#pragma unroll
                    for (int t = 0; t < NT; ++t) ra[o][t] = fmaf(wo, v[t], ra[o][t]);
                }
            }
        }
    if (cp.rgb_w) {
        // Sum over the four lane groups (rows of 16 lanes) with the gfx950 row / half swaps: for four values a, b, c, d
        //   v_permlane16_swap(a, b) -> [a0 b0 a2 b2], [a1 b1 a3 b3]  (rows; sum = a01 b01 a23 b23)
        //   v_permlane32_swap(sum_ab, sum_cd) -> [a01 b01 c01 d01], [a23 b23 c23 d23]  (sum: row g = total of value g)
        // -- three swaps and three adds per four values, and row g of the wave ends up with value 4 k + g of group k:
        // every lane finishes ONE pixel of up to three outputs (24 ds_bpermute + one lane group doing 12 sigmoids before).
        auto sw16 = [](float x, float y) {
            auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
            const unsigned r0 = r[0], r1 = r[1];       // (bit_cast of a vector ELEMENT reads element 0 with this hipcc: scalars first)
            return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
        };
        auto sw32 = [](float x, float y) {
            auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
            const unsigned r0 = r[0], r1 = r[1];       // (bit_cast of a vector ELEMENT reads element 0 with this hipcc: scalars first)
            return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
        };
        constexpr int NV = 3 * NT, NG = (NV + 3) / 4;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            float v4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v4[q] = 4 * k + q < NV ? ra[(4 * k + q) / NT][(4 * k + q) % NT] : 0.0f;
            const float tot = sw32(sw16(v4[0], v4[1]), sw16(v4[2], v4[3]));
            const int idx = 4 * k + g;
            if (idx < NV) {
                const int o = idx / NT, t = idx - o * NT;
                const long off = ((long)b * 3 + o) * cp.P + n + t;
                float r = tot + cp.rgb_bias[o];
                if (PRE) { if (cp.rgb_accumulate) r += pre->rgb[k]; }
                else if (cp.rgb_accumulate) r += cp.rgb[off];
                cp.rgb[off] = r;
                if (cp.rgb_img) {
                    r = 1.0f / (1.0f + expf(-r));
                    cp.rgb_img[off] = r;
                }
                if (cp.rgb_out) cp.rgb_out[off] = r;
            }
        }
    }
}

// The PixelShuffleUpsample tail as an epilogue of conv16_kernel<.., SHUF> (a function of its own since round 6, when a persistent form of
// the kernel shared it: profiles/r6_n1_experiments.txt -- measured slower, removed): register e of
// acc[mt][t] is channel m0 + 16 mt + 4 g + e at pixel n + t of image b.
template <int MT, int NT>
__device__ __forceinline__ void conv16_shuffle_epilogue(const Conv16Params& cp, f32x4 (&acc)[MT][NT], int b, int n, int m0, int g) {
    typedef typename Pix<NT>::T pv;
    // PixelShuffleUpsample tail.  A lane's four registers of one tile are in-channels 4c..4c+3 of its NT pixels, i.e.
    // the 2x2 output blocks of out-channel c at NT consecutive x: two rows of 2 NT consecutive floats; the four
    // pre-activation signs per pixel go into one nibble, the lane's pixels into one 8 NT-bit store.
    const int Cq = cp.M / 4;
    const int py = n / cp.W, px = n - py * cp.W;
    // x.repeat(1,4,1,1): in-channel m reads x channel m % (M/4).  ONE division per lane; the channels of the lane's
    // tiles follow by adding 16 mt + e and wrapping (a runtime modulo per channel was ~35 VALU, eight times per wave)
    const int rbase = (m0 + 4 * g) % Cq;
    auto res_row = [&](int add) {
        int rr = rbase + add;
        if (Cq >= 16 * MT + 4) { if (rr >= Cq) rr -= Cq; }       // add < 16 MT + 4 <= Cq: at most one wrap
        else rr %= Cq;
        return rr;
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int mb = m0 + 16 * mt + 4 * g;                  // multiple of 4
        if (mb >= cp.M) continue;
        float v[4][NT];                                       // [e: channel][t: pixel]
        unsigned nib = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = mb + e;
            const float bias = (GNR_C16_ABL & 32) ? 0.5f : cp.bias[m];
            // x.repeat(1,4,1,1): in-channel m reads x channel m % (M/4)
            pv res;
            if (GNR_C16_ABL & 32) res = pv(0.25f);
            else res = *(const pv*)(cp.res + (long)b * cp.res_batch + (long)res_row(16 * mt + e) * cp.P + n);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float u = acc[mt][t][e] + bias;
                nib |= (u > 0.0f ? 1u : 0u) << (8 * t + e);
                u = u > 0.0f ? u : LEAK16 * u;
                v[e][t] = u + res[t];
            }
        }
        if ((GNR_C16_ABL & 64) && (nib != 0x12345u || v[0][0] != 1.2345f)) continue;
        unsigned char* sp = cp.sign_out + (long)b * cp.sign_batch + (long)(mb >> 2) * cp.P + n;
        if constexpr (NT == 4) *(unsigned*)sp = nib;
        else *(unsigned short*)sp = (unsigned short)nib;
        // pixel_shuffle(2): in-channel 4c + 2i + j -> out (c, 2y+i, 2x+j)
        float* dst = cp.C + (((GNR_C16_ABL & 128) ? 0x3FFF8L : -1L) & ((long)b * cp.c_batch + (long)(mb >> 2) * (4L * cp.P) + (long)(2 * py) * (2 * cp.W) + 2 * px));
#pragma unroll
        for (int t = 0; t < NT; t += 2) {
            *(f32x4*)(dst + 2 * t) = f32x4{v[0][t], v[1][t], v[0][t + 1], v[1][t + 1]};
            *(f32x4*)(dst + 2 * cp.W + 2 * t) = f32x4{v[2][t], v[3][t], v[2][t + 1], v[3][t + 1]};
        }
    }
}

template <int MT, int NT, bool SHUF, bool BLUR>
__global__ __launch_bounds__(64 * WPB, 2) void conv16_kernel(const Conv16Params cp) {
    typedef typename Pix<NT>::T pv;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 15, g = lane >> 4;
    const int slices = cp.plan.slices;
    const unsigned items = (unsigned)((long)cp.batch * cp.P / (16 * WPB * NT)) * (unsigned)slices;
    const unsigned per_xcd = (items + 7u) >> 3;
    const unsigned item = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (item >= items) return;
    const unsigned pt = item / (unsigned)slices;
    const int ms = (int)(item - pt * (unsigned)slices);
    const unsigned pixg = pt * (unsigned)(16 * WPB * NT) + (unsigned)wave * (unsigned)(16 * NT);     // batch * P < 2^31
    const int b = (int)(pixg / (unsigned)cp.P);
    const int p0 = (int)(pixg - (unsigned)b * (unsigned)cp.P);    // the wave's first pixel inside image b
    const int nkb = cp.plan.nkb;
    const int m0 = ms * (16 * MT);
    // the RGB branch's 3 x M weights wait in LDS for the epilogue (the only LDS of the kernel, one barrier at the start:
    // fetched from memory in the epilogue their latency was exposed -- 307 instead of 241 us at 7 x 32 x 512 x 512)
    __shared__ float rgbw[SHUF ? 1 : 3 * 16 * MT];
    if constexpr (!SHUF) {
        if (cp.rgb_w) {                                           // kernel argument: uniform
            for (int i = (int)threadIdx.x; i < 3 * cp.M; i += 64 * WPB) rgbw[i] = cp.rgb_w[i];
            __syncthreads();
        }
    }

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(cp.At + (long)ms * nkb * (MT * 256)), 0, nkb * (MT * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(cp.B + (long)b * cp.b_batch), 0, (int)((long)cp.K * cp.P * 4), 0x00020000);
    const unsigned voffA = (unsigned)lane * 16u;
    const unsigned rowB = (unsigned)cp.P * 4u;                    // bytes per k row
    const int n = p0 + NT * li;                                   // this lane's NT consecutive pixels
    const unsigned voffB = ((unsigned)g * (unsigned)cp.P + (unsigned)n) * 4u;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    f32x4 Aq[2][MT];
    auto load_a = [&](int kb, f32x4 (&A)[MT]) {
        const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane(((GNR_C16_ABL & 4) ? 0 : kb) * (MT * 1024));      // keep the stream offset scalar
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            A[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA, (int)(sa + (unsigned)mt * 1024u), 0));
    };
    auto compute = [&](const f32x4 (&A)[MT], const pv (&Bv)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mt][t] = mfma16c(A[mt][s], Bv[s][t], acc[mt][t]);
    };

    if constexpr (!BLUR) {
        pv Bq[2][4];
        auto load_b = [&](int kb, pv (&Bv)[4]) {
            const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((GNR_C16_ABL & 2) ? 0 : kb) * 16u * rowB));
#pragma unroll
            for (int s = 0; s < 4; ++s) Bv[s] = Pix<NT>::load(rsB, voffB, sb + (unsigned)s * 4u * rowB);
        };
        load_b(0, Bq[0]);
        load_a(0, Aq[0]);
        int kb = 0;
        for (; kb + 1 < nkb; kb += 2) {
            load_b(kb + 1, Bq[1]);
            load_a(kb + 1, Aq[1]);
            __builtin_amdgcn_sched_barrier(0);
            compute(Aq[0], Bq[0]);
            __builtin_amdgcn_sched_barrier(0);
            const int k2 = kb + 2 < nkb ? kb + 2 : nkb - 1;       // even nkb: one redundant request at the end
            load_b(k2, Bq[0]);
            load_a(k2, Aq[0]);
            __builtin_amdgcn_sched_barrier(0);
            compute(Aq[1], Bq[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nkb & 1) compute(Aq[0], Bq[0]);
    } else {
        // B = blur(u): rows y-1, y, y+1 of the lane's NT pixels plus the two edge neighbours, reflect padding.
        const int W = cp.W, H = cp.H;
        const int y = n / W, x = n - y * W;
        float wl[NT], wr[NT], yl, yc, yr;
        {
            float wc;
            blur_taps16(y, H, yl, yc, yr);
#pragma unroll
            for (int e = 0; e < NT; ++e) blur_taps16(x + e, W, wl[e], wc, wr[e]);
        }
        const int y0 = y >= 1 ? y - 1 : y, y2 = y + 1 < H ? y + 1 : y;
        // The two neighbours outside a lane's NT pixels belong to the lanes next to it: a DPP row shift (a row = the 16
        // lanes of one k-group) supplies them, except at the two ends of the wave's pixel run.  Those 2 x 16 rows x 3 image
        // rows = 96 values per k-block are fetched by TWO dword loads (lane l: k-step l / 16, k-group (l / 4) % 4, image
        // row l % 4), reduced over the three image rows inside their lane quads and handed to the end lanes by ds_bpermute.
        // The stencil is applied rows-first (its taps along the row do not depend on the row): NT + 2 column sums, then
        // three taps -- 30 VALU per operand quad, not 48; blur_kernel's order differs by rounding only.  (Round-3 measurement: one dword load per neighbour per
        // lane -- 24 per k-block, each touching as many cache lines as a 16-byte load -- cost more than the MFMAs:
        // feat_layers 161 / 196 / 343 us with them, 121 / 125 / 197 without.)
        const int x0w = p0 - y * W;                                // first column of the wave's run (wave-uniform)
        unsigned vc[3], ve_l, ve_r;
        {
            const int ys[3] = {y0, y, y2};
#pragma unroll
            for (int q = 0; q < 3; ++q) vc[q] = ((unsigned)g * (unsigned)cp.P + (unsigned)(ys[q] * W) + (unsigned)x) * 4u;
            const int es = lane >> 4, eg = (lane >> 2) & 3, eq = (lane & 3) < 3 ? (lane & 3) : 2;
            const unsigned ebase = (unsigned)(4 * es + eg) * (unsigned)cp.P + (unsigned)(ys[eq] * W);
            ve_l = (ebase + (unsigned)(x0w >= 1 ? x0w - 1 : 0)) * 4u;                       // weight 0 when outside (reflect folds inward)
            ve_r = (ebase + (unsigned)(x0w + 16 * NT < W ? x0w + 16 * NT : W - 1)) * 4u;
        }
        const int perm_base = 16 * g;                              // ds_bpermute byte address of source lane 4 g (+ 16 s + q)
        pv rc[4][3];
        float edge_l, edge_r;
        auto load_raw = [&](int kb) {
            const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((GNR_C16_ABL & 2) ? 0 : kb) * 16u * rowB));
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < 3; ++q) rc[s][q] = Pix<NT>::load(rsB, vc[q], sb + (unsigned)s * 4u * rowB);
            edge_l = load1(rsB, ve_l, sb);
            edge_r = load1(rsB, ve_r, sb);
        };
        // The raw rows of block kb+1 are requested before the MFMAs of block kb and folded into its B operand right after
        // them: only the 4 NT operand registers are carried from one iteration to the next (with the raw rows carried
        // instead, hipcc copies all 18 NT of them at the loop head and spills).
        pv Bv[2][4];
        auto combine = [&](pv (&Bo)[4]) {
            // the three image rows of an end column sit in three adjacent lanes of the edge load (l % 4 = row): their
            // weighted sum by quad broadcasts, once per k-block
            auto quad_rows = [&](float v) {
                const int iv = __builtin_bit_cast(int, v);
                const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(iv, 0x00, 0xf, 0xf, true));     // quad_perm [0,0,0,0]
                const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(iv, 0x55, 0xf, 0xf, true));     // [1,1,1,1]
                const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(iv, 0xAA, 0xf, 0xf, true));     // [2,2,2,2]
                return yl * r0 + yc * r1 + yr * r2;
            };
            const float comb_l = quad_rows(edge_l), comb_r = quad_rows(edge_r);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // rows first (the taps along the row do not depend on the row): NT + 2 columns, then the three taps
                float col[NT + 2];
                {
#pragma unroll
                    for (int e = 0; e < NT; ++e) col[e + 1] = yl * rc[s][0][e] + yc * rc[s][1][e] + yr * rc[s][2][e];
                    const float el = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_base + 64 * s, __builtin_bit_cast(int, comb_l)));
                    const float er = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_base + 64 * s, __builtin_bit_cast(int, comb_r)));
                    col[0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, el), __builtin_bit_cast(int, col[NT]), 0x111, 0xf, 0xf, false));
                    col[NT + 1] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, er), __builtin_bit_cast(int, col[1]), 0x101, 0xf, 0xf, false));
                }
#pragma unroll
                for (int e = 0; e < NT; ++e) Bo[s][e] = wl[e] * col[e] + 0.5f * col[e + 1] + wr[e] * col[e + 2];
            }
        };
        load_raw(0);
        load_a(0, Aq[0]);
        combine(Bv[0]);
        int kb = 0;
        for (; kb + 1 < nkb; kb += 2) {
            load_raw(kb + 1);
            load_a(kb + 1, Aq[1]);
            __builtin_amdgcn_sched_barrier(0);
            compute(Aq[0], Bv[0]);
            __builtin_amdgcn_sched_barrier(0);
            combine(Bv[1]);
            __builtin_amdgcn_sched_barrier(0);
            const int k2 = kb + 2 < nkb ? kb + 2 : nkb - 1;
            load_raw(k2);
            load_a(k2, Aq[0]);
            __builtin_amdgcn_sched_barrier(0);
            compute(Aq[1], Bv[1]);
            __builtin_amdgcn_sched_barrier(0);
            combine(Bv[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nkb & 1) compute(Aq[0], Bv[0]);
    }

    // ---- epilogue: register e of acc[mt][t] is channel m0 + 16 mt + 4 g + e at pixel n + t ----
    if ((GNR_C16_ABL & 1) && cp.K != -12345) return;
    if constexpr (SHUF) {
        conv16_shuffle_epilogue<MT, NT>(cp, acc, b, n, m0, g);
    } else {
        conv16_plain_epilogue<MT, NT>(cp, acc, b, n, m0, g, rgbw);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// conv16_blur_lds_kernel (round 4, late): the blur-fused feat_layers GEMM with its operand staged through LDS.
// The register-fed BLUR instances of conv16_kernel are bound by the vector L1, not by the matrix pipe or by latency: every wave
// requests the three image rows its stencil needs (12 KB per k-block) plus two 64-lane gathers for the halo columns, twelve waves
// per CU -- more L1 cycles than MFMA cycles (profiles/r4_n1_pmc_fwdbwd_b7.txt: mfma_busy 0.16-0.42).  Here a
// workgroup owns 4 image rows x 64 columns (wave w: row y0 + w, a lane 4 consecutive pixels as before): per k-block its 256
// threads bring 16 channels x 6 rows (the tile + one halo row above and below) x 64 columns and the two halo columns into LDS
// ONCE -- 24 KB and 192 gather lines where the four waves requested 48 KB and 384 -- and every wave applies the stencil
// (rows first, then the three taps: the SAME expressions as conv16_kernel, so the operand values are bit-identical) to its
// three rows from LDS.  Global loads of block kb + 1 are in flight in 25 staging registers during the MFMAs of block kb; two
// workgroup barriers per k-block (one LDS buffer: 27.6 KB, four workgroups per CU).  One row slice only (M <= 64: the RGB rider
// stays on the epilogue), W % 64 == 0, H % 4 == 0; everything else keeps the register-fed kernel.
// ---------------------------------------------------------------------------------------------------------------
constexpr int BL_RS = 72;           // floats per LDS row (halo 3 | 64 pixels 4..67 | halo 68)
constexpr int BL_ROWS = 4;          // image rows per workgroup tile = waves per workgroup.  8 (halo 1.25 x instead of 1.5 x, 46 KB of LDS, two
                                    // workgroups per CU) was measured in round 5: 246 / 183 us against 240 / 137 (profiles/r5_n1_experiments.txt)
template <int MT, int ROWS, bool ODD>          // ODD: the number of k-blocks is odd (which Aq set the last block uses is then static)
__global__ __launch_bounds__(64 * ROWS, (ROWS == 4 ? (MT <= 2 ? 4 : 3) : (MT <= 2 ? 4 : 2))) void conv16_blur_lds_kernel(const Conv16Params cp) {
    constexpr int NT = 4, TR = ROWS + 2, NTH = 64 * ROWS;
    constexpr int SEGS = 16 * TR, NJ = (SEGS * 16 + NTH - 1) / NTH, NHALO = 2 * 16 * TR;      // 256-byte row segments, b128 pieces per thread, halo elements
    static_assert(SEGS * 16 == NJ * NTH, "every thread stages the same number of pieces");
    static_assert(NHALO <= NTH, "one halo element per thread");
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int W = cp.W, H = cp.H;
    const unsigned tiles_x = (unsigned)W >> 6, tiles_y = (unsigned)H / ROWS;
    const unsigned items = (unsigned)cp.batch * tiles_y * tiles_x;
    const unsigned per_xcd = (items + 7u) >> 3;
    const unsigned item = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (item >= items) return;
    const int b = (int)(item / (tiles_y * tiles_x));
    const unsigned rem = item - (unsigned)b * (tiles_y * tiles_x);
    const int y0 = ROWS * (int)(rem / tiles_x), x0 = 64 * (int)(rem - (rem / tiles_x) * tiles_x);
    const int y = y0 + wave;
    const int n = y * W + x0 + NT * li;                            // this lane's 4 consecutive pixels inside image b
    const int nkb = cp.plan.nkb;

    __shared__ float rgbw[3 * 16 * MT];
    __shared__ float tile[16 * TR * BL_RS];
    if (cp.rgb_w)
        for (int i = tid; i < 3 * cp.M; i += NTH) rgbw[i] = cp.rgb_w[i];       // visible after the first barrier below

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)cp.At, 0, nkb * (MT * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(cp.B + (long)b * cp.b_batch), 0, (int)((long)cp.K * cp.P * 4), 0x00020000);
    const unsigned voffA = (unsigned)lane * 16u;
    const unsigned rowB = (unsigned)cp.P * 4u;

    float wl[NT], wr[NT], yl, yc, yr;
    {
        float wc;
        blur_taps16(y, H, yl, yc, yr);
#pragma unroll
        for (int e = 0; e < NT; ++e) blur_taps16(x0 + NT * li + e, W, wl[e], wc, wr[e]);
    }
    // staging: thread t brings 16-byte piece t % 16 of the row segments t / 16 + (NTH / 16) j (segment = channel * TR + tile row)
    // and, for t < NHALO, one halo column element.  Rows / columns outside the image are clamped onto a valid one: their taps are zero.
    unsigned vst[NJ], lst[NJ], vh, lh;
    {
        const int piece = tid & 15;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int seg = (tid >> 4) + (NTH / 16) * j, ch = seg / TR, r = seg - ch * TR;
            int iy = y0 - 1 + r;
            iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
            vst[j] = ((unsigned)ch * (unsigned)cp.P + (unsigned)(iy * W + x0 + 4 * piece)) * 4u;
            lst[j] = (unsigned)((ch * TR + r) * BL_RS + 4 + 4 * piece);
        }
        const int side = tid >= NHALO / 2 ? 1 : 0, idx = tid - (NHALO / 2) * side, ch = idx / TR, r = idx - ch * TR;
        int iy = y0 - 1 + r;
        iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
        int ix = side ? x0 + 64 : x0 - 1;
        ix = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        vh = tid < NHALO ? ((unsigned)ch * (unsigned)cp.P + (unsigned)(iy * W + ix)) * 4u : 0xFFFFFF00u;
        lh = (unsigned)((ch * TR + r) * BL_RS + (side ? 68 : 3));
    }
    f32x4 stg[NJ];                                     // ONE staging set: a second k-block in flight was measured (148 / 202 VGPRs, three /
    float sth;                                         // two waves per SIMD: 255 / 156 us against 238 / 139) -- occupancy beats prefetch depth here
    auto load_stage = [&](int kb) {                    // rows >= K: beyond the descriptor's bound -- zeros, no request
        const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((GNR_C16_ABL & 2) ? 0 : kb) * 16u * rowB));
#pragma unroll
        for (int j = 0; j < NJ; ++j) stg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, vst[j], (int)sb, 0));
        if (!(GNR_C16_ABL & 16)) sth = load1(rsB, vh, sb);          // 16: timing experiment, no halo-column loads
        else sth = 0.0f;
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) *(f32x4*)&tile[lst[j]] = stg[j];
        if (tid < NHALO) tile[lh] = sth;
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 Aq[2][MT];
    auto load_a = [&](int kb, f32x4 (&A)[MT]) {
        const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane(kb * (MT * 1024));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            A[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA, (int)(sa + (unsigned)mt * 1024u), 0));
    };
    // lane group g takes k = 16 kb + 4 s + g at step s (the packed A operand's order): channel 4 s + g of the staged block,
    // tile rows wave .. wave + 2 = image rows y - 1, y, y + 1
    const float* my = &tile[(g * TR + wave) * BL_RS + 4 + 4 * li];
    auto compute = [&](const f32x4 (&A)[MT]) {
        if ((GNR_C16_ABL & 8) && cp.K != -12345) return;          // timing experiment: no stencil, no MFMAs
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* base = my + s * (4 * TR * BL_RS);
            const f32x4 c0 = *(const f32x4*)base, c1 = *(const f32x4*)(base + BL_RS), c2 = *(const f32x4*)(base + 2 * BL_RS);
            float col[NT + 2];
            col[0] = yl * base[-1] + yc * base[BL_RS - 1] + yr * base[2 * BL_RS - 1];
#pragma unroll
            for (int e = 0; e < NT; ++e) col[e + 1] = yl * c0[e] + yc * c1[e] + yr * c2[e];
            col[NT + 1] = yl * base[NT] + yc * base[BL_RS + NT] + yr * base[2 * BL_RS + NT];
            f32x4 Bo;
#pragma unroll
            for (int e = 0; e < NT; ++e) Bo[e] = wl[e] * col[e] + 0.5f * col[e + 1] + wr[e] * col[e + 2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mt][t] = mfma16c(A[mt][s], Bo[t], acc[mt][t]);
        }
    };
    // one k-block: the next block's A rows, this block's MFMAs, then LDS <- the staged block kb + 1 and (REQ) staging <- block kb + 2
    auto step = [&](auto cur, int kb, auto req) {
        constexpr int CUR = decltype(cur)::value;
        load_a(kb + 1, Aq[1 - CUR]);
        compute(Aq[CUR]);
        __syncthreads();
        store_stage();
        if constexpr (decltype(req)::value) load_stage(kb + 2);
        __syncthreads();
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using Req = std::true_type;
    using NoReq = std::false_type;

    load_stage(0);
    load_a(0, Aq[0]);
    store_stage();
    if (1 < nkb) load_stage(1);
    __syncthreads();
    // The last block is peeled, so that its MFMAs run with the staging registers free: that is where the epilogue's own operands
    // are requested (conv16_epilogue_prefetch).  Requests past the last block read zeros through the descriptor's bound.
    Conv16EpiPre<MT, NT> pre;
    int kb = 0;
    if constexpr (ODD) {
        for (; kb + 1 < nkb; kb += 2) {
            step(C0{}, kb, Req{});
            step(C1{}, kb + 1, Req{});
        }
        conv16_epilogue_prefetch<MT, NT>(cp, b, n, 0, g, pre);
        compute(Aq[0]);
    } else {
        for (; kb + 2 < nkb; kb += 2) {
            step(C0{}, kb, Req{});
            step(C1{}, kb + 1, Req{});
        }
        step(C0{}, kb, NoReq{});
        conv16_epilogue_prefetch<MT, NT>(cp, b, n, 0, g, pre);
        compute(Aq[1]);
    }

    if ((GNR_C16_ABL & 1) && cp.K != -12345) return;              // timing experiment: no epilogue
    conv16_plain_epilogue<MT, NT, true>(cp, acc, b, n, 0, g, rgbw, &pre);
}

// du = Wf^T g with the adjoint of the PixelShuffleUpsample tail in the epilogue (round 4; until then the GEMM wrote du
// [C][2S x 2S] and unshuffle_bwd4_kernel read it back: 2 x 470 MB per 7 images at the 256 x 256 level).  NT = 8: a wave owns
// one low-resolution row y and 32 columns, a lane the 2 x 2 blocks of TWO adjacent low-resolution pixels -- its B load per
// k row is 4 consecutive pixels of image row 2y and of row 2y + 1 (two b128; 16 lanes = 256 contiguous bytes), pixel tile
// t = 4 i + 2 xs + j is sub-pixel (i, j) of its pixel xs.  The rows are packed with perm4 (conv16_pack_kernel): register q of
// a lane's accumulator quad is channel cb + q C/4, so the four terms of every x.repeat-adjoint sum
//   dres(4 cb + e) = sum_q G(cb + q C/4, sub-pixel e)
// sit in ONE lane (the same fixed order as unshuffle_bwd4_kernel: results are bit-identical to the two-kernel path), and
// dpre2 = G * lrelu'(pre2) leaves as 8-byte stores (16 lanes = 128 contiguous bytes of one channel plane).
// (nontemporal stores of dpre2 / dres were measured in round 5: 213 us against 193 at the 64-channel level -- profiles/r5_n1_experiments.txt)
__device__ __forceinline__ void ustore2(float* p, f32x2 v) {
    if ((GNR_C16_ABL & 64) && v.x != 1.2345f) return;             // timing experiment: no stores
    *(f32x2*)p = v;
}
template <int MT, bool PERM>
__global__ __launch_bounds__(64 * WPB, 2) void conv16_unshuffle_kernel(const Conv16Params cp) {
    constexpr int NT = 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 15, g = lane >> 4;
    const int slices = cp.plan.slices;
    const int S = cp.W, xchunks = S >> 5;                          // low-resolution side; 32-column chunks per row
    const unsigned wave_items = (unsigned)cp.batch * (unsigned)(S * xchunks);
    const unsigned items = (wave_items / WPB) * (unsigned)slices;
    const unsigned per_xcd = (items + 7u) >> 3;
    const unsigned item = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (item >= items) return;
    const unsigned pt = item / (unsigned)slices;
    const int ms = (int)(item - pt * (unsigned)slices);
    const unsigned wi = pt * WPB + (unsigned)wave;                 // (image, row, chunk), chunk fastest
    const int b = (int)(wi / (unsigned)(S * xchunks));
    const int rem = (int)(wi - (unsigned)b * (unsigned)(S * xchunks));
    const int y = rem / xchunks, xw = (rem - y * xchunks) << 5;
    const int nkb = cp.plan.nkb;
    const int m0 = ms * (16 * MT);

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(cp.At + (long)ms * nkb * (MT * 256)), 0, nkb * (MT * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(cp.B + (long)b * cp.b_batch), 0, (int)((long)cp.K * cp.P * 4), 0x00020000);
    const unsigned voffA = (unsigned)lane * 16u;
    const unsigned rowB = (unsigned)cp.P * 4u;                    // bytes per k row (P = 4 S S high-resolution pixels)
    const int x0 = xw + 2 * li;                                   // the lane's two low-resolution pixels: x0, x0 + 1
    unsigned voffB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) voffB[i] = ((unsigned)g * (unsigned)cp.P + (unsigned)((2 * y + i) * (2 * S) + 2 * x0)) * 4u;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    f32x4 Aq[2][MT];
    f32x4 Bq[2][4][2];
    auto load_a = [&](int kb, f32x4 (&A)[MT]) {
        const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane(kb * (MT * 1024));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            A[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA, (int)(sa + (unsigned)mt * 1024u), 0));
    };
    auto load_b = [&](int kb, f32x4 (&Bv)[4][2]) {
        const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)kb * 16u * rowB));
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                Bv[s][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB[i], (int)(sb + (unsigned)s * 4u * rowB), 0));
    };
    auto compute = [&](const f32x4 (&A)[MT], const f32x4 (&Bv)[4][2]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[mt][t] = mfma16c(A[mt][s], Bv[s][t >> 2][t & 3], acc[mt][t]);
    };
    load_b(0, Bq[0]);
    load_a(0, Aq[0]);
    int kb = 0;
    for (; kb + 1 < nkb; kb += 2) {
        load_b(kb + 1, Bq[1]);
        load_a(kb + 1, Aq[1]);
        __builtin_amdgcn_sched_barrier(0);
        compute(Aq[0], Bq[0]);
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = kb + 2 < nkb ? kb + 2 : nkb - 1;
        load_b(k2, Bq[0]);
        load_a(k2, Aq[0]);
        __builtin_amdgcn_sched_barrier(0);
        compute(Aq[1], Bq[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (nkb & 1) compute(Aq[0], Bq[0]);

    // ---- epilogue: register q of acc[mt][4 i + 2 xs + j] is du(channel of row q, 2y + i, 2 (x0 + xs) + j) ----
    const long Plo = (long)S * S;
    const long pix = (long)y * S + x0;
    if constexpr (PERM) {
        // rows packed with perm4: row q of a lane's quad is channel cb + q Cq -- the four x.repeat-adjoint terms in one lane
        const int Cq = cp.M >> 2;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int cb = (m0 >> 2) + 4 * mt + g;
            if (cb >= Cq) continue;
            f32x2 G[4][4];                                        // [q][sub-pixel e = 2 i + j] over the lane's two pixels
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = cb + q * Cq;
                const unsigned nib2 = *(const unsigned short*)(cp.sign_in + (long)b * cp.sign_batch + (long)c * Plo + pix);
                float* dst = cp.C + (long)b * cp.c_batch + (long)(4 * c) * Plo + pix;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    G[q][e] = f32x2{acc[mt][4 * (e >> 1) + (e & 1)][q], acc[mt][4 * (e >> 1) + 2 + (e & 1)][q]};
                    ustore2(dst + (long)e * Plo, f32x2{G[q][e].x * (((nib2 >> e) & 1u) ? 1.0f : LEAK16),
                                                       G[q][e].y * (((nib2 >> (8 + e)) & 1u) ? 1.0f : LEAK16)});
                }
            }
            float* dr = cp.dres + (long)b * cp.dres_batch + (long)(4 * cb) * Plo + pix;
#pragma unroll
            for (int e = 0; e < 4; ++e) ustore2(dr + (long)e * Plo, (G[0][e] + G[1][e]) + (G[2][e] + G[3][e]));
        }
    } else {
        // any channel count: rows in their natural order, dpre2 only -- the x.repeat adjoint's four terms of an output sit in
        // four different row slices and are collected by unshuffle_dres_kernel from dpre2
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = m0 + 16 * mt + 4 * g + q;
                if (c >= cp.M) continue;
                const unsigned nib2 = *(const unsigned short*)(cp.sign_in + (long)b * cp.sign_batch + (long)c * Plo + pix);
                float* dst = cp.C + (long)b * cp.c_batch + (long)(4 * c) * Plo + pix;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    ustore2(dst + (long)e * Plo,
                            f32x2{acc[mt][4 * (e >> 1) + (e & 1)][q] * (((nib2 >> e) & 1u) ? 1.0f : LEAK16),
                                  acc[mt][4 * (e >> 1) + 2 + (e & 1)][q] * (((nib2 >> (8 + e)) & 1u) ? 1.0f : LEAK16)});
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// upchain_kernel: PixelShuffleUpsample's layer_1 -> layer_2 chained in registers (gnr_conv16.h).  T1 = row tiles of the middle
// activation (all of them in registers: 4 T1 NP), NP = pixel tiles per wave (a lane owns NP consecutive pixels).
// Weight rows stream per wave in batches of 4 row tiles (b128 per lane = 4 MFMA steps), one batch ahead, as in the MLP
// chain (gnr_chain16.h); two waves share a SIMD.  Biases wait in LDS (one barrier at the start).
// Measured (profiles/r4_n1_chain_experiment.txt): a gain only at the 64-channel level (453 -> ~395 us per 7 images, 72 -> 60 us
// per image); the 129-channel instance and the backward variant (the two data gradients chained) were slower than their two
// GEMMs -- short contractions under a 7-VALU-per-element epilogue need conv16_kernel's five waves per SIMD -- and were removed.
// ---------------------------------------------------------------------------------------------------------------
template <int T1, int NP>
__global__ __launch_bounds__(64 * WPB, 2) void upchain_kernel(const UpChainParams cp) {
    constexpr int TB = 4, MS = 8 / NP, NB1 = (T1 + TB - 1) / TB, T1P = NB1 * TB;      // MS: row tiles of a phase-2 slab (64 accumulator registers)
    typedef typename Pix<NP>::T pv;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 15, g = lane >> 4;
    const unsigned groups = (unsigned)((long)cp.batch * cp.P / (16 * WPB * NP));
    const unsigned per_xcd = (groups + 7u) >> 3;
    const unsigned item = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (item >= groups) return;
    __shared__ float bias_lds[16 * T1 + 64 * T1];                 // [M1 padded][M2 <= 2 M1 padded]
    for (int i = (int)threadIdx.x; i < cp.M1 + cp.M2; i += 64 * WPB) bias_lds[i < cp.M1 ? i : 16 * T1 + (i - cp.M1)] = i < cp.M1 ? cp.bias1[i] : cp.bias2[i - cp.M1];
    __syncthreads();
    const unsigned pixg = item * (unsigned)(16 * WPB * NP) + (unsigned)wave * (unsigned)(16 * NP);
    const int b = (int)(pixg / (unsigned)cp.P);
    const int p0 = (int)(pixg - (unsigned)b * (unsigned)cp.P);
    const int n = p0 + NP * li;
    const int nkb1 = cp.plan.nkb1, slabs2 = cp.plan.slabs2;

    const __amdgpu_buffer_rsrc_t rsA1 = __builtin_amdgcn_make_buffer_rsrc((void*)cp.A1, 0, nkb1 * (T1P * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)cp.A2, 0, slabs2 * (T1 * MS * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(cp.B + (long)b * cp.b_batch), 0, (int)((long)cp.K1 * cp.P * 4), 0x00020000);
    const unsigned voffA = (unsigned)lane * 16u;
    const unsigned rowB = (unsigned)cp.P * 4u;
    const unsigned voffB = ((unsigned)g * (unsigned)cp.P + (unsigned)n) * 4u;

    f32x4 acc1[T1][NP];
#pragma unroll
    for (int t1 = 0; t1 < T1; ++t1)
#pragma unroll
        for (int t = 0; t < NP; ++t) acc1[t1][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // ---- phase 1: acc1 = A1 B over K1 (B from memory; lane group g takes k = 16 kb + 4 s + g) ----
    {
        pv Bq[2][4];
        f32x4 Aq[2][TB];
        unsigned sa = 0;                                          // byte offset of the next A batch (wave-uniform)
        auto load_b = [&](int kb, pv (&Bv)[4]) {
            const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)kb * 16u * rowB));
#pragma unroll
            for (int s = 0; s < 4; ++s) Bv[s] = Pix<NP>::load(rsB, voffB, sb + (unsigned)s * 4u * rowB);
        };
        auto load_a = [&](f32x4 (&A)[TB]) {
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)sa);
#pragma unroll
            for (int q = 0; q < TB; ++q)
                A[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA1, voffA, (int)(so + (unsigned)q * 1024u), 0));
            sa += TB * 1024u;
        };
        // one k-block: NB1 batches; the A buffers alternate along the linear batch stream, PAR = parity of its first batch
        auto kblock = [&](auto par, const pv (&Bv)[4]) {
            constexpr int PAR = decltype(par)::value;
#pragma unroll
            for (int i = 0; i < NB1; ++i) {
                load_a(Aq[(PAR + i + 1) & 1]);                    // the next batch of the stream (runs off the end harmlessly)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int q = 0; q < TB; ++q)
                        if (i * TB + q < T1) {
#pragma unroll
                            for (int t = 0; t < NP; ++t)
                                acc1[i * TB + q][t] = mfma16c(Aq[(PAR + i) & 1][q][s], Bv[s][t], acc1[i * TB + q][t]);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        load_b(0, Bq[0]);
        load_a(Aq[0]);
        int kb = 0;
        for (; kb + 1 < nkb1; kb += 2) {
            load_b(kb + 1, Bq[1]);
            kblock(std::integral_constant<int, 0>{}, Bq[0]);
            load_b(kb + 2 < nkb1 ? kb + 2 : nkb1 - 1, Bq[0]);
            kblock(std::integral_constant<int, NB1 & 1>{}, Bq[1]);
        }
        if (nkb1 & 1) kblock(std::integral_constant<int, 0>{}, Bq[0]);
    }

    // ---- epilogue 1: the middle activation, stored and kept (row 16 t1 + 4 g + e, pixels n .. n + NP - 1) ----
#pragma unroll
    for (int t1 = 0; t1 < T1; ++t1)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = 16 * t1 + 4 * g + e;
            const bool live = m < cp.M1;
            pv v;
#pragma unroll
            for (int t = 0; t < NP; ++t) v[t] = acc1[t1][t][e];
            v += bias_lds[m];                                     // (rows >= M1: unset LDS, discarded below)
#pragma unroll
            for (int t = 0; t < NP; ++t) v[t] = v[t] > 0.0f ? v[t] : LEAK16 * v[t];
            if (live) *(pv*)(cp.out1 + (long)b * cp.out1_batch + (long)m * cp.P + n) = v;
#pragma unroll
            for (int t = 0; t < NP; ++t) acc1[t1][t][e] = live ? v[t] : 0.0f;
        }

    // ---- phase 2: slabs of MS row tiles over K2 = 16 T1 (step e of input tile j = register e of acc1[j]) ----
    f32x4 A2q[2][MS];
    unsigned sa2 = 0;
    auto load_a2 = [&](f32x4 (&A)[MS]) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)sa2);
#pragma unroll
        for (int q = 0; q < MS; ++q)
            A[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA2, voffA, (int)(so + (unsigned)q * 1024u), 0));
        sa2 += MS * 1024u;
    };
    auto slab = [&](auto par, int sl) {
        constexpr int PAR = decltype(par)::value;
        f32x4 acc2[MS][NP];
#pragma unroll
        for (int mt = 0; mt < MS; ++mt)
#pragma unroll
            for (int t = 0; t < NP; ++t) acc2[mt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // the residual rows are requested before the slab's MFMAs
        const int m0 = sl * (16 * MS);
        const int Cq = cp.M2 / 4;
        const int rbase = (m0 + 4 * g) % Cq;                      // x.repeat: in-channel m reads x channel m % Cq; one division per slab,
                                                                  // the slab's rows follow by adding < 16 MS + 4 <= Cq and wrapping once
        pv pre[MS][4];
#pragma unroll
        for (int mt = 0; mt < MS; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int rr = rbase + 16 * mt + e;
                if (rr >= Cq) rr -= Cq;
                pre[mt][e] = *(const pv*)(cp.res + (long)b * cp.res_batch + (long)rr * cp.P + n);
            }
#pragma unroll
        for (int j = 0; j < T1; ++j) {
            load_a2(A2q[(PAR + j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MS; ++mt)
#pragma unroll
                    for (int t = 0; t < NP; ++t) acc2[mt][t] = mfma16c(A2q[(PAR + j) & 1][mt][e], acc1[j][t][e], acc2[mt][t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // PixelShuffleUpsample tail, as conv16_kernel<.., SHUF>: a lane's four registers of a tile are in-channels
        // 4c .. 4c+3 of its NP pixels = the 2 x 2 output blocks of out-channel c at NP consecutive x
        const int py = n / cp.W, px = n - py * cp.W;
#pragma unroll
        for (int mt = 0; mt < MS; ++mt) {
            const int mb = m0 + 16 * mt + 4 * g;
            if (mb >= cp.M2) continue;
            float v[4][NP];
            unsigned nib = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = mb + e;
                const float bias = bias_lds[16 * T1 + m];
                const pv res = pre[mt][e];
#pragma unroll
                for (int t = 0; t < NP; ++t) {
                    float u = acc2[mt][t][e] + bias;
                    nib |= (u > 0.0f ? 1u : 0u) << (8 * t + e);
                    u = u > 0.0f ? u : LEAK16 * u;
                    v[e][t] = u + res[t];
                }
            }
            unsigned char* sp = cp.sign_out + (long)b * cp.sign_batch + (long)(mb >> 2) * cp.P + n;
            if constexpr (NP == 4) *(unsigned*)sp = nib;
            else *(unsigned short*)sp = (unsigned short)nib;
            float* dst = cp.out2 + (long)b * cp.out2_batch + (long)(mb >> 2) * (4L * cp.P) + (long)(2 * py) * (2 * cp.W) + 2 * px;
#pragma unroll
            for (int t = 0; t < NP; t += 2) {
                *(f32x4*)(dst + 2 * t) = f32x4{v[0][t], v[1][t], v[0][t + 1], v[1][t + 1]};
                *(f32x4*)(dst + 2 * cp.W + 2 * t) = f32x4{v[2][t], v[3][t], v[2][t + 1], v[3][t + 1]};
            }
        }
    };
    load_a2(A2q[0]);
    int sl = 0;
    if constexpr (T1 & 1) {
        for (; sl + 1 < slabs2; sl += 2) {
            slab(std::integral_constant<int, 0>{}, sl);
            slab(std::integral_constant<int, 1>{}, sl + 1);
        }
        if (slabs2 & 1) slab(std::integral_constant<int, 0>{}, sl);
    } else {
        for (; sl < slabs2; ++sl) slab(std::integral_constant<int, 0>{}, sl);
    }
}

std::atomic<int> g_forced_tile{0};          // 100 MT + NT, 0 = cost model (gnr_set_conv16_tile)
std::atomic<int> g_unshuffle_mt{0};         // row tiles of the fused un-shuffle GEMM pinned by gnr_set_conv16_tile(MT, 8); 0 = heuristic
struct Variant { int MT, NT; bool blur; };
// (row tiles, pixel tiles) instances; blur: the instance that reads B through the stencil exists (register budget)
// (ties in the cost model go to the earlier entry: the smaller tiles, which measured equal or better -- DESIGN.md 3.5)
// (round 4: half-width 2x2 / 4x2 instances were measured for the single-image forward and removed again -- a B = 1 GEMM is not
// short of waves, profiles/r4_n1_small_tiles.txt)
const Variant kVariants[] = {{2, 4, true}, {4, 4, true}, {8, 4, false}, {9, 2, true}, {11, 2, false}, {13, 2, false}};
constexpr int kUnshuffleMT[] = {2, 3, 4};

template <int MT, int NT>
void launch_variant(const Conv16Params& cp, unsigned blocks, hipStream_t st) {
    if (cp.shuffle) hipLaunchKernelGGL((conv16_kernel<MT, NT, true, false>), dim3(blocks), dim3(64 * WPB), 0, st, cp);
    else hipLaunchKernelGGL((conv16_kernel<MT, NT, false, false>), dim3(blocks), dim3(64 * WPB), 0, st, cp);
}
template <int MT, int NT>
void launch_variant_blur(const Conv16Params& cp, unsigned blocks, hipStream_t st) {
    if (cp.blur) hipLaunchKernelGGL((conv16_kernel<MT, NT, false, true>), dim3(blocks), dim3(64 * WPB), 0, st, cp);
    else launch_variant<MT, NT>(cp, blocks, st);
}

}  // namespace

int conv16_set_tile(int mt, int nt) {
    if (mt == 0 && nt == 0) { g_forced_tile = 0; g_unshuffle_mt = 0; return 0; }
    if (nt == 8) {                          // the fused un-shuffle GEMM's row tiles; the plain GEMMs keep the cost model
        for (int v : kUnshuffleMT)
            if (v == mt) { g_forced_tile = 0; g_unshuffle_mt = mt; return 0; }
    }
    for (const Variant& v : kVariants)
        if (v.MT == mt && v.NT == nt) { g_forced_tile = 100 * mt + nt; g_unshuffle_mt = 0; return 0; }
    return fail("gnr_set_conv16_tile: no GEMM instance with %d row tiles x %d pixel tiles (have 2x4, 4x4, 8x4, 9x2, 11x2, 13x2, and "
                "2x8, 3x8, 4x8 for the GEMM with the fused un-shuffle; 0, 0 restores the cost model)", mt, nt);
}

// du = Wf^T g + un-shuffle in one kernel: a wave's 32 low-resolution columns must lie inside one row.  With M % 4 == 0 the rows
// are packed with perm4 and the x.repeat adjoint (dres) comes out of the same epilogue; otherwise the epilogue writes dpre2 and
// unshuffle_dres_kernel collects dres from it.  A pinned plain tile (gnr_set_conv16_tile) means "the two-kernel path".
Conv16Plan conv16_plan_unshuffle(int M, int K, int side) {
    Conv16Plan p{};
    if (side % 32 || g_forced_tile.load()) return p;
    const int tiles = (M + 15) / 16;
    int mt = g_unshuffle_mt.load();
    if (!mt) {
        // fewest padded row tiles, then the smallest tile (64 channels at 256 x 256, 7 images: 2 x (2,8) 190 us, 1 x (4,8) 210 us)
        int best_pad = 1 << 30;
        for (int v : kUnshuffleMT) {
            const int sl = (tiles + v - 1) / v, pad = sl * v - tiles;
            if (pad < best_pad) { best_pad = pad; mt = v; }
        }
    }
    p.MT = mt; p.NT = 8;
    p.slices = (tiles + mt - 1) / mt;
    p.nkb = (K + 15) / 16;
    p.pack_floats = (size_t)p.slices * p.nkb * p.MT * 256;
    return p;
}

Conv16Plan conv16_plan(int M, int K, long pixels_total, int blur_w) {
    const int tiles = (M + 15) / 16;
    Conv16Plan best{};
    double best_cost = 0.0;
    // tuning / test hook (gnr_set_conv16_tile): one (MT, NT) pair for every GEMM; blur-fused GEMMs whose forced pair has
    // no stencil instance fall back to the separate stencil (MT == 0 on return), exactly as without the hook
    const int forced = g_forced_tile.load();
    const int fmt = forced / 100, fnt = forced % 100;
    for (const Variant& v : kVariants) {
        if (blur_w && (!v.blur || blur_w % (16 * v.NT))) continue;      // a wave's pixels must lie in one image row
        if (fmt && (v.MT != fmt || v.NT != fnt)) continue;
        const int slices = (tiles + v.MT - 1) / v.MT;
        const double waves = (double)slices * (double)(pixels_total / (16 * v.NT));
        // two waves share a SIMD's matrix pipe: below 1024 waves the chip is not full and a wave's length is the time
        // rounds of 1024 waves; a few rounds are whole rounds (single-image sweep: 1088 waves of a (4,4) tile take as long as
        // 2048 -- 75 us against 56 for the 2112 waves of (2,4) and 48 for 1024 waves of (9,2))
        double rounds = waves / 1024.0;
        if (rounds <= 4.0) rounds = std::ceil(rounds);
        double cost = rounds * v.MT * v.NT * (v.NT == 2 ? 1.06 : 1.0);
        // every row slice repeats the stencil's loads (three rows + edges) and its ~30 VALU per operand register: the
        // fewest slices win (measured, 7 images: M = 64 as 1 x (4,4) 185 us, as 2 x (2,4) 249 us; M = 129 as (9,2) 160 us)
        if (blur_w) cost *= 1.0 + 0.5 * (slices - 1);
        if (best.MT == 0 || cost < best_cost - 1e-9) {
            best.MT = v.MT; best.NT = v.NT; best.slices = slices;
            best_cost = cost;
        }
    }
    best.nkb = (K + 15) / 16;
    best.pack_floats = (size_t)best.slices * best.nkb * best.MT * 256;
    return best;
}

long conv16_add_job(Conv16PackJobs& jobs, const float* W, long rs, long cs, int M, int K, const Conv16Plan& plan, int perm4) {
    Conv16PackJobs::Job& J = jobs.j[jobs.n];
    long off = 0;
    for (int i = 0; i < jobs.n; ++i) off += jobs.j[i].floats;
    J.W = W; J.rs = rs; J.cs = cs; J.M = M; J.K = K; J.MT = plan.MT; J.nkb = plan.nkb; J.slices = plan.slices; J.perm4 = perm4; J.kchain = 0;
    J.dst_off = off; J.floats = (long)plan.pack_floats;
    ++jobs.n;
    return off;
}

void launch_conv16_pack(const Conv16PackJobs& jobs, hipStream_t st) {
    long total = 0;
    for (int i = 0; i < jobs.n; ++i) total += jobs.j[i].floats;
    if (total == 0) return;
    hipLaunchKernelGGL(conv16_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, jobs);
}

// The chained pair has one instance: 128 middle channels (the 64-channel level), 64 pixels per wave.  A pinned tile
// (gnr_set_conv16_tile) means the two-GEMM path.
UpChainPlan upchain_plan(int K1, int M1, int M2, long pixels_per_image) {
    UpChainPlan p{};
    if (g_forced_tile.load() || g_unshuffle_mt.load()) return p;
    const int t1 = (M1 + 15) / 16;
    const int np = t1 == 8 ? 4 : 0;                               // the one instance: 128 middle channels, 64 pixels per wave
    if (!np || pixels_per_image % (16 * WPB * np)) return p;
    // shuffle epilogue: four in-channels per out-channel, biases of both layers in LDS, one wrap of the residual row
    if (M2 > 2 * 16 * t1 || M2 % 4 || M2 / 4 < 16 * (8 / np) + 4) return p;
    p.T1 = t1; p.NP = np;
    p.nkb1 = (K1 + 15) / 16;
    const int ms = 8 / np;
    p.slabs2 = (M2 + 16 * ms - 1) / (16 * ms);
    p.pack1_floats = (size_t)p.nkb1 * ((t1 + 3) / 4 * 4) * 256;
    p.pack2_floats = (size_t)p.slabs2 * t1 * ms * 256;
    return p;
}

void upchain_add_jobs(Conv16PackJobs& jobs, const UpChainPlan& plan, const float* W1, long rs1, long cs1, int M1, int K1,
                      const float* W2, long rs2, long cs2, int M2, long* o1, long* o2) {
    Conv16Plan a{};
    a.MT = (plan.T1 + 3) / 4 * 4; a.NT = 0; a.slices = 1; a.nkb = plan.nkb1; a.pack_floats = plan.pack1_floats;
    *o1 = conv16_add_job(jobs, W1, rs1, cs1, M1, K1, a);
    Conv16Plan c{};
    c.MT = 8 / plan.NP; c.NT = 0; c.slices = plan.slabs2; c.nkb = plan.T1; c.pack_floats = plan.pack2_floats;
    *o2 = conv16_add_job(jobs, W2, rs2, cs2, M2, M1, c);
    jobs.j[jobs.n - 1].kchain = 1;
}

int launch_upchain(const UpChainParams& cp, hipStream_t st) {
    const long groups = (long)cp.batch * cp.P / (16 * WPB * cp.plan.NP);
    const unsigned blocks = (unsigned)(8 * ((groups + 7) / 8));
    if (cp.plan.T1 == 8 && cp.plan.NP == 4) hipLaunchKernelGGL((upchain_kernel<8, 4>), dim3(blocks), dim3(64 * WPB), 0, st, cp);
    else return fail("upchain: no instance for %d middle row tiles x %d pixel tiles", cp.plan.T1, cp.plan.NP);
    return 0;
}

int launch_conv16(const Conv16Params& cp, hipStream_t st) {
    // the folded x.repeat adjoint exists in the plain epilogue of the NT == 4 instances only (conv16_plain_epilogue); any other
    // route would drop it silently -- and its caller has switched `accumulate` off, so d(net) would be wrong without an error
    if (cp.dres_from && (cp.plan.NT != 4 || cp.shuffle || cp.sign_in))
        return fail("conv16: dres_from needs a plain NT == 4 instance (plan %d x %d, shuffle %d, un-shuffle %d)", cp.plan.MT, cp.plan.NT,
                    cp.shuffle, cp.sign_in != nullptr);
    if (cp.plan.NT == 8) {
        const long witems = (long)cp.batch * cp.W * (cp.W / 32) / WPB * cp.plan.slices;
        const unsigned wblocks = (unsigned)(8 * ((witems + 7) / 8));
        const int key = 2 * cp.plan.MT + (cp.dres ? 1 : 0);       // dres given: rows packed with perm4 (M % 4 == 0)
        if (cp.dres && cp.M % 4) return fail("conv16: the un-shuffle epilogue with dres needs M %% 4 == 0 (M = %d)", cp.M);
        switch (key) {
            case 5: hipLaunchKernelGGL((conv16_unshuffle_kernel<2, true>), dim3(wblocks), dim3(64 * WPB), 0, st, cp); break;
            case 7: hipLaunchKernelGGL((conv16_unshuffle_kernel<3, true>), dim3(wblocks), dim3(64 * WPB), 0, st, cp); break;
            case 9: hipLaunchKernelGGL((conv16_unshuffle_kernel<4, true>), dim3(wblocks), dim3(64 * WPB), 0, st, cp); break;
            case 4: hipLaunchKernelGGL((conv16_unshuffle_kernel<2, false>), dim3(wblocks), dim3(64 * WPB), 0, st, cp); break;
            case 6: hipLaunchKernelGGL((conv16_unshuffle_kernel<3, false>), dim3(wblocks), dim3(64 * WPB), 0, st, cp); break;
            case 8: hipLaunchKernelGGL((conv16_unshuffle_kernel<4, false>), dim3(wblocks), dim3(64 * WPB), 0, st, cp); break;
            default: return fail("conv16: no fused un-shuffle instance for MT = %d", cp.plan.MT);
        }
        return 0;
    }
    const long items = (long)cp.batch * cp.P / (16 * WPB * cp.plan.NT) * cp.plan.slices;
    const unsigned blocks = (unsigned)(8 * ((items + 7) / 8));
    // blur-fused feat_layers GEMM with one row slice of <= 64 channels on whole 4 x 64 tiles: the LDS-staged kernel (a pinned
    // tile -- gnr_set_conv16_tile -- keeps the register-fed instance: the test hook compares the two)
    if (cp.blur && !cp.shuffle && cp.plan.NT == 4 && cp.plan.slices == 1 && (cp.plan.MT == 2 || cp.plan.MT == 4) &&
        cp.W % 64 == 0 && cp.H % BL_ROWS == 0 && (long)cp.W * cp.H == cp.P && !g_forced_tile.load() &&
        items >= 512) {          // below two workgroups per CU its barriers are exposed (one 256 x 256 image: 30.7 us against 27.3)
        const long bitems = (long)cp.batch * (cp.H / BL_ROWS) * (cp.W / 64);
        const unsigned bblocks = (unsigned)(8 * ((bitems + 7) / 8));
        const bool odd = cp.plan.nkb & 1;
        if (cp.plan.MT == 2) {
            if (odd) hipLaunchKernelGGL((conv16_blur_lds_kernel<2, BL_ROWS, true>), dim3(bblocks), dim3(64 * BL_ROWS), 0, st, cp);
            else hipLaunchKernelGGL((conv16_blur_lds_kernel<2, BL_ROWS, false>), dim3(bblocks), dim3(64 * BL_ROWS), 0, st, cp);
        } else {
            if (odd) hipLaunchKernelGGL((conv16_blur_lds_kernel<4, BL_ROWS, true>), dim3(bblocks), dim3(64 * BL_ROWS), 0, st, cp);
            else hipLaunchKernelGGL((conv16_blur_lds_kernel<4, BL_ROWS, false>), dim3(bblocks), dim3(64 * BL_ROWS), 0, st, cp);
        }
        return 0;
    }
    const int key = cp.plan.MT * 10 + cp.plan.NT;
    switch (key) {
        case 132: launch_variant<13, 2>(cp, blocks, st); break;
        case 112: launch_variant<11, 2>(cp, blocks, st); break;
        case 92: launch_variant_blur<9, 2>(cp, blocks, st); break;
        case 84: launch_variant<8, 4>(cp, blocks, st); break;
        case 44: launch_variant_blur<4, 4>(cp, blocks, st); break;
        case 24: launch_variant_blur<2, 4>(cp, blocks, st); break;
        default: return fail("conv16: no instance for MT = %d, NT = %d", cp.plan.MT, cp.plan.NT);
    }
    return 0;
}

}  // namespace gnr
