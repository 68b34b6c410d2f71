// gnr_upsample.hip -- the 2-D upsampler after the volumetric hot path, gfx950.
//
// Replaces NeuralRenderer.forward (models/neural_renderer.py:100-113) with PixelShuffleUpsample
// (models/pixel_shuffle_upsample.py:33-42) and Blur (:7-16), forward and backward:
//
//   rgb = up(conv_rgb0(x));  net = x
//   block i:  a1 = lrelu(W1 net + b1);  a2 = lrelu(W2 a1 + b2)
//             u  = pixel_shuffle(a2 + repeat(net, 4), 2);  v = blur(u)
//             net = lrelu(Wf v + bf);  rgb = rgb + conv_rgb(i+1)(net);  if not last: rgb = up(rgb)
//   img = sigmoid(rgb);   up = blur o bilinear-x2 (align_corners=False);  blur = reflect-padded [1,2,1]^2/16
//
// All tensors are the reference's channels-first fp32 images [B][C][H*W].  The 1x1 convolutions are GEMMs over pixels
// (C[M][N] = A[M][K] B[K][N], N = pixels contiguous): conv16_kernel (gnr_conv16.hip), exact fp32 MFMA fed from registers
// by several independent waves per SIMD, epilogues fused: bias + LeakyReLU, the residual-repeat + pixel_shuffle store of
// the PixelShuffleUpsample tail (with the sign nibble the backward needs), LeakyReLU-derivative masks and accumulation
// for the dgrad GEMMs (A = W^T by strides).  Round 3:
//   * blur(u) is never written: the feat_layers GEMM reads its operand through the stencil; the backward applies the
//     adjoint stencil to the half-width gradient instead (blur and the 1x1 convolution commute), so u is what is saved;
//   * weight gradients: wgrad2w_kernel / wgrad_kernel (gnr_wgrad.hip) on the image layout, bias gradients as their column
//     sums over all images;
//   * the 3-channel RGB branch: forward one thread per pixel, backward ONE pass over the activations for the weight and
//     the data gradient (rgb_bwd_fused_kernel);
//   * stencils (blur adjoint, bilinear and its adjoint, un-shuffle) are HBM-bound gather kernels with 16-byte accesses.
// Work per 64x64 -> 512x512 image: 19.6 GFLOP forward (9.8 GMAC), ~0.77 GB of activation traffic.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "../../include/gnr.h"
#include "gnr_canary.h"
#include "gnr_conv16.h"
#include "gnr_device.h"
#include "gnr_wgrad.h"

namespace gnr {
int fail(const char* fmt, ...);
size_t wgrad_arena_floats(int batch, int max_m, int max_k);
void launch_wgrad_img(const float* A, int lda, int n_valid, const float* B, int ldb, int k_valid, int batch,
                      long pixels_per_image, float* dW, int ldw, float* colsum_out, int colsum_ld, float* scratch,
                      hipStream_t stream, WgradDefer* defer);

constexpr float LEAK = 0.2f;
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int UP_MAX = GNR_UPSAMPLE_MAX_BLOCKS;

// ---------------------------------------------------------------------------------------------
// stencils: out(plane, y, x) over planes = B*C images of H x W
// ---------------------------------------------------------------------------------------------
// Blur taps (reflect padding == kornia filter2d border_type='reflect'): forward row y reads y-1, y, y+1 with the
// out-of-range neighbour reflected onto the inner one; the adjoint gathers with the transposed weights.
__device__ __forceinline__ void blur_taps(int u, int n, bool adjoint, float& wl, float& wc, float& wr) {
    wc = 0.5f;
    if (!adjoint) {
        wl = u >= 1 ? 0.25f : 0.0f;
        wr = u + 1 < n ? 0.25f : 0.0f;
        if (u == 0) wr += 0.25f;          // in[-1] -> in[1]
        if (u == n - 1) wl += 0.25f;      // in[n]  -> in[n-2]
    } else {
        wl = u >= 1 ? (u == 1 ? 0.5f : 0.25f) : 0.0f;
        wr = u + 1 < n ? (u == n - 2 ? 0.5f : 0.25f) : 0.0f;
    }
}

// One thread = 4 consecutive output pixels of a row (W % 4 == 0): three float4 row loads plus the two edge
// neighbours per row, one float4 store -- the kernel is HBM-bound (reads and writes every plane once).
__global__ __launch_bounds__(256) void blur_kernel(const float* __restrict__ in, float* __restrict__ out, long planes,
                                                   int H, int W, int adjoint) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;              // index of the pixel quad
    const long total = planes * H * (W / 4);
    if (q >= total) return;
    const int xq = (int)(q % (W / 4)), y = (int)((q / (W / 4)) % H);
    const int x = 4 * xq;
    const float* p = in + (q / ((long)(W / 4) * H)) * ((long)H * W);
    float yl, yc, yr;
    blur_taps(y, H, adjoint, yl, yc, yr);
    const int y0 = y >= 1 ? y - 1 : y, y2 = y + 1 < H ? y + 1 : y;
    float wl[4], wc[4], wr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) blur_taps(x + e, W, adjoint, wl[e], wc[e], wr[e]);
    const int xm = x >= 1 ? x - 1 : x, xp = x + 4 < W ? x + 4 : x + 3;
    // The two neighbours outside the thread's four pixels are the end pixels of the adjacent lanes' quads (a strided dword
    // load per neighbour touches as many cache lines as the 16-byte load itself): lane shuffles, and a real load only in
    // the first / last lane of the wave.  At the image's left / right edge the neighbour's weight is 0 resp. the reflected
    // pixel is the thread's own.
    const int lane = threadIdx.x & 63;
    auto row = [&](int yy) {
        const float* r = p + (long)yy * W;
        const f32x4 c = *(const f32x4*)(r + x);
        float l = __shfl_up(c.w, 1), rr = __shfl_down(c.x, 1);
        if (lane == 0 || x == 0) l = r[xm];
        if (lane == 63 || x + 4 >= W) rr = r[xp];
        return f32x4{wl[0] * l + wc[0] * c.x + wr[0] * c.y, wl[1] * c.x + wc[1] * c.y + wr[1] * c.z,
                     wl[2] * c.y + wc[2] * c.z + wr[2] * c.w, wl[3] * c.z + wc[3] * c.w + wr[3] * rr};
    };
    const f32x4 a = row(y0), b = row(y), c = row(y2);
    *(f32x4*)(out + (q / ((long)(W / 4) * H)) * ((long)H * W) + (long)y * W + x) = yl * a + yc * b + yr * c;
}

// bilinear x2, align_corners=False: out[2m] = .25 in[m-1] + .75 in[m] (m = 0: in[0]); out[2m+1] = .75 in[m] + .25 in[m+1]
// blur(bilinear2x(in)) in one pass for the 3-channel RGB branch of the forward (neural_renderer.py:104-112: rgb_upsample =
// Upsample(scale 2, bilinear) + Blur): a thread = 4 consecutive pixels of an output row; the 3 x 6 bilinear values it needs
// are built from the input and combined with blur_kernel's expression (rows first), the 2S x 2S intermediate is never
// written (rounds 1-3: bilinear2x_kernel + blur_kernel).  in != out.
// Round 5: the 3 x 6 bilinear values of a thread touch only a 3 x 4 patch of the input -- rows m-1, m, m+1 of m = y / 2 and
// columns 2q-1 .. 2q+2 of its quad q: which two of them a bilinear value reads is fixed by the parity of its output row /
// column (columns: x = 4q is even, so the pattern is static; rows: one select on the parity of y).  12 loads per thread
// instead of 72 (every tap of every value fetched on its own: 38.5 us per 7 x 3 x 512 x 512 outputs, 0.6 TB/s, issue-bound
// on L1 hits); the weights still come from taps() with the edge rules, and a patch element outside the image is the clamped
// one the old index rule named or carries weight 0, so every value is the same expression of the same operands as before.
__global__ __launch_bounds__(256) void bilinear_blur_kernel(const float* __restrict__ in, float* __restrict__ out, long planes,
                                                            int H, int W) {
    const int H2 = 2 * H, W2 = 2 * W;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= planes * H2 * (W2 / 4)) return;
    const int xq = (int)(q % (W2 / 4)), y = (int)((q / (W2 / 4)) % H2);
    const int x = 4 * xq;
    const long plane = q / ((long)(W2 / 4) * H2);
    const float* p = in + plane * ((long)H * W);
    auto taps_w = [](int o, float& w0, float& w1) {             // weights of the two taps of output index o (the edge rules)
        const int m = o >> 1;
        if (o & 1) { w0 = 0.75f; w1 = 0.25f; }
        else { w0 = m >= 1 ? 0.25f : 0.0f; w1 = m >= 1 ? 0.75f : 1.0f; }
    };
    const int xm = x >= 1 ? x - 1 : x, xp = x + 4 < W2 ? x + 4 : x + 3;
    float cw0[6], cw1[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) taps_w(e == 0 ? xm : (e == 5 ? xp : x + e - 1), cw0[e], cw1[e]);
    float yl, yc, yr, wl[4], wc[4], wr[4];
    blur_taps(y, H2, false, yl, yc, yr);
#pragma unroll
    for (int e = 0; e < 4; ++e) blur_taps(x + e, W2, false, wl[e], wc[e], wr[e]);
    const int ys[3] = {y >= 1 ? y - 1 : y, y, y + 1 < H2 ? y + 1 : y};
    // the patch: rows m-1, m, m+1 and columns 2q-1 .. 2q+2, clamped into the image
    const int m = y >> 1, c1 = 2 * xq;
    const int pr[3] = {m >= 1 ? m - 1 : 0, m, m + 1 < H ? m + 1 : H - 1};
    const int pc[4] = {c1 >= 1 ? c1 - 1 : 0, c1, c1 + 1, c1 + 2 < W ? c1 + 2 : W - 1};
    float v[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[j][k] = p[(long)pr[j] * W + pc[k]];
    // output column x + e - 1 (e = 0 .. 5) reads patch columns (e / 2, e / 2 + 1): x - 1 odd -> (2q-1, 2q), x even -> the same, ...
    const bool odd = y & 1;
    f32x4 rows[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float wy0, wy1;
        taps_w(ys[r], wy0, wy1);
        // output row y - 1 + r reads patch rows (0, 1) or (1, 2): y even -> (0,1) (0,1) (1,2); y odd -> (0,1) (1,2) (1,2)
        const bool hi = r == 2 || (r == 1 && odd);
        float u[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const int k = e >> 1;
            const float a0 = hi ? v[1][k] : v[0][k], a1 = hi ? v[1][k + 1] : v[0][k + 1];
            const float b0 = hi ? v[2][k] : v[1][k], b1 = hi ? v[2][k + 1] : v[1][k + 1];
            u[e] = wy0 * (cw0[e] * a0 + cw1[e] * a1) + wy1 * (cw0[e] * b0 + cw1[e] * b1);
        }
        rows[r] = f32x4{wl[0] * u[0] + wc[0] * u[1] + wr[0] * u[2], wl[1] * u[1] + wc[1] * u[2] + wr[1] * u[3],
                        wl[2] * u[2] + wc[2] * u[3] + wr[2] * u[4], wl[3] * u[3] + wc[3] * u[4] + wr[3] * u[5]};
    }
    *(f32x4*)(out + plane * ((long)H2 * W2) + (long)y * W2 + x) = yl * rows[0] + yc * rows[1] + yr * rows[2];
}

// Adjoint of rgb_upsample = Blur o bilinear x2 in ONE pass (round 5; until then blur_kernel(adjoint) wrote the blurred
// 2H x 2W gradient and bilinear2x_adj_kernel gathered 16 of its values per output):
//   din(y, x) = sum_{a,c} wy[a] wx[c] Bt(2y - 1 + a, 2x - 1 + c),   Bt = blur^T(dout)
// A thread owns one low-resolution pixel: the 6 x 6 patch rows 2y-2 .. 2y+3, columns 2x-2 .. 2x+3 of dout (three 8-byte loads
// per row in the interior), the stencil's row pass on it (6 rows x 4 columns), then its column pass on the 4 x 4 values the
// bilinear adjoint combines -- blur_kernel's expressions (rows first), then the 4 x 4 gather of round 1-4's bilinear2x_adj_kernel; taps that fall outside
// the image carry weight 0 in both, so clamped patch elements stand in for them.
__global__ __launch_bounds__(256) void blur_bilinear_adj_kernel(const float* __restrict__ dout, float* __restrict__ din,
                                                                long planes, int H, int W) {
    const int H2 = 2 * H, W2 = 2 * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * H * W) return;
    const int x = (int)(idx % W), y = (int)((idx / W) % H);
    const float* p = dout + (idx / ((long)H * W)) * ((long)H2 * W2);
    auto clampi = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
    // bilinear adjoint taps: outputs 2m-1 .. 2m+2 read in[m] (see the forward rule above bilinear_blur_kernel)
    float wx[4], wy[4];
    wx[0] = x >= 1 ? 0.25f : 0.0f; wx[1] = x >= 1 ? 0.75f : 1.0f; wx[2] = x + 1 < W ? 0.75f : 1.0f; wx[3] = x + 1 < W ? 0.25f : 0.0f;
    wy[0] = y >= 1 ? 0.25f : 0.0f; wy[1] = y >= 1 ? 0.75f : 1.0f; wy[2] = y + 1 < H ? 0.75f : 1.0f; wy[3] = y + 1 < H ? 0.25f : 0.0f;
    // blur adjoint taps of the four columns X = 2x-1+c and the four rows Y = 2y-1+a (clamped where the bilinear weight is 0)
    float bxl[4], bxc[4], bxr[4], byl[4], byc[4], byr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        blur_taps(clampi(2 * x - 1 + c, W2), W2, true, bxl[c], bxc[c], bxr[c]);
        blur_taps(clampi(2 * y - 1 + c, H2), H2, true, byl[c], byc[c], byr[c]);
    }
    const bool interior = x >= 1 && x + 1 < W;
    float xb[6][4];                                              // row pass: patch row j, column X = 2x-1+c
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float* r = p + (long)clampi(2 * y - 2 + j, H2) * W2;
        float v[6];
        if (interior) {
            const f32x2 a = *(const f32x2*)(r + 2 * x - 2), b = *(const f32x2*)(r + 2 * x), c = *(const f32x2*)(r + 2 * x + 2);
            v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = r[clampi(2 * x - 2 + k, W2)];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[j][c] = bxl[c] * v[c] + bxc[c] * v[c + 1] + bxr[c] * v[c + 2];
    }
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float rsum = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) rsum += wx[c] * (byl[a] * xb[a][c] + byc[a] * xb[a + 1][c] + byr[a] * xb[a + 2][c]);
        acc += wy[a] * rsum;
    }
    din[idx] = acc;
}

// ---------------------------------------------------------------------------------------------
// the 3-channel RGB branch
// ---------------------------------------------------------------------------------------------
// rgb(b,o,p) = [rgb(b,o,p) +] sum_c W[o][c] net(b,c,p) + bias[o];  img = sigmoid(rgb) if wanted;  out = the caller's
// image (img if given, else rgb) -- the last block writes it directly instead of a device-to-device copy afterwards
__global__ __launch_bounds__(256) void rgb_conv_kernel(const float* __restrict__ net, int C, long P, int batch,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ rgb, int accumulate, float* __restrict__ img,
                                                       float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * P) return;
    const long b = idx / P, p = idx - b * P;
    const float* np = net + b * C * P + p;
    float a0 = bias[0], a1 = bias[1], a2 = bias[2];
    // channel order kept (three in-order fmaf chains); the activations of 8 channels are loaded ahead of their FMAs: on
    // the 64 x 64 input (16 blocks, 258 channels) a load-then-fma chain took 124 us
    int c = 0;
    for (; c + 8 <= C; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = np[(long)(c + u) * P];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = fmaf(w[c + u], v[u], a0);
            a1 = fmaf(w[C + c + u], v[u], a1);
            a2 = fmaf(w[2 * C + c + u], v[u], a2);
        }
    }
    for (; c < C; ++c) {
        const float v = np[(long)c * P];
        a0 = fmaf(w[c], v, a0);
        a1 = fmaf(w[C + c], v, a1);
        a2 = fmaf(w[2 * C + c], v, a2);
    }
    float* rp = rgb + b * 3 * P + p;
    if (accumulate) { a0 += rp[0]; a1 += rp[P]; a2 += rp[2 * P]; }
    rp[0] = a0; rp[P] = a1; rp[2 * P] = a2;
    if (img) {
        float* ip = img + b * 3 * P + p;
        a0 = 1.0f / (1.0f + expf(-a0)); a1 = 1.0f / (1.0f + expf(-a1)); a2 = 1.0f / (1.0f + expf(-a2));
        ip[0] = a0; ip[P] = a1; ip[2 * P] = a2;
    }
    if (out) {
        float* op = out + b * 3 * P + p;
        op[0] = a0; op[P] = a1; op[2 * P] = a2;
    }
}

// The RGB branch's backward in one pass over the activations (round 2: a weight-gradient kernel that re-read d(rgb) once
// per channel, then a data-gradient kernel that read the activations again):
//   dnet(b,c,p) = ([dnet(b,c,p)] + sum_o W[o][c] drgb(b,o,p)) * (net(b,c,p) > 0 ? 1 : 0.2)        [mask optional]
//   dW[o][c]    = sum_{b,p} drgb(b,o,p) net(b,c,p);  db[o] = sum drgb(b,o,p)   (channel c == C stands for the bias)
// A thread owns RGBF_IT pixel quads (their d(rgb) stays in registers) and walks the channels: per channel one b128 load of
// the activation (+ one of dnet when accumulating), one b128 store, and three partial dots that are reduced over the wave
// once per channel; a workgroup's four wave sums meet in LDS and leave as ONE partial per (channel, output):
// part[(c * nwg + wg) * 3 + o]; rgb_wsum_batch_kernel adds the workgroups' partials in a fixed order (deterministic).
// Grid (pixel groups, channel groups): at the low resolutions (4096 pixels x 258 channels) the pixels alone are a handful
// of workgroups, so the channel range is split too (blockIdx.y; the bias pseudo-channel rides with the last group).
constexpr int RGBF_IT = 4;
constexpr int RB_AHEAD = 1;             // rgb_bwd_blur_kernel: channels requested ahead (2 / 3 measured slower, profiles/r5_n1_experiments.txt)
static long rgbf_workgroups(long pixels_total) { return (pixels_total / 4 + 256 * RGBF_IT - 1) / (256 * RGBF_IT); }
static int rgbf_channel_groups(long nwg, int C) {          // ~1024 workgroups in all, at least 8 channels each
    long g = (1024 + nwg - 1) / nwg;
    if (g > C / 8) g = C / 8;
    return g < 1 ? 1 : (int)g;
}

// The same product for SMALL maps (batch * P <= 32 768 pixels: the 64 x 64 level, where 4096 threads walking 258 channels
// each were 33 dependent load rounds = 18.6 us of a 0.4 ms inference forward): a workgroup owns 64 pixels, its four waves a
// quarter of the channels each (in-order fmaf chains), partial sums combined through LDS in wave order -- deterministic;
// differs from rgb_conv_kernel by summation order only.
__global__ __launch_bounds__(256) void rgb_conv_split_kernel(const float* __restrict__ net, int C, long P, int batch,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ rgb, int accumulate, float* __restrict__ img,
                                                             float* __restrict__ out) {
    __shared__ float part[4][3][64];
    const int px = threadIdx.x & 63, cq = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 64 + px;
    const bool live = idx < (long)batch * P;
    const long b = live ? idx / P : 0, p = live ? idx - b * P : 0;
    const float* np = net + b * C * P + p;
    const int cper = (C + 3) / 4, c0 = cq * cper, c1 = c0 + cper < C ? c0 + cper : C;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = np[(long)(c + u) * P];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = fmaf(w[c + u], v[u], a0);
            a1 = fmaf(w[C + c + u], v[u], a1);
            a2 = fmaf(w[2 * C + c + u], v[u], a2);
        }
    }
    for (; c < c1; ++c) {
        const float v = np[(long)c * P];
        a0 = fmaf(w[c], v, a0);
        a1 = fmaf(w[C + c], v, a1);
        a2 = fmaf(w[2 * C + c], v, a2);
    }
    part[cq][0][px] = a0; part[cq][1][px] = a1; part[cq][2][px] = a2;
    __syncthreads();
    if (cq != 0 || !live) return;
    a0 = bias[0] + ((part[0][0][px] + part[1][0][px]) + (part[2][0][px] + part[3][0][px]));
    a1 = bias[1] + ((part[0][1][px] + part[1][1][px]) + (part[2][1][px] + part[3][1][px]));
    a2 = bias[2] + ((part[0][2][px] + part[1][2][px]) + (part[2][2][px] + part[3][2][px]));
    float* rp = rgb + b * 3 * P + p;
    if (accumulate) { a0 += rp[0]; a1 += rp[P]; a2 += rp[2 * P]; }
    rp[0] = a0; rp[P] = a1; rp[2 * P] = a2;
    if (img) {
        float* ip = img + b * 3 * P + p;
        a0 = 1.0f / (1.0f + expf(-a0)); a1 = 1.0f / (1.0f + expf(-a1)); a2 = 1.0f / (1.0f + expf(-a2));
        ip[0] = a0; ip[P] = a1; ip[2 * P] = a2;
    }
    if (out) {
        float* op = out + b * 3 * P + p;
        op[0] = a0; op[P] = a1; op[2 * P] = a2;
    }
}

__global__ __launch_bounds__(256) void rgb_bwd_fused_kernel(const float* __restrict__ drgb, const float* __restrict__ net, int C,
                                                            long P, int batch, const float* __restrict__ w,
                                                            float* __restrict__ dnet, int accumulate, int masked,
                                                            float* __restrict__ part) {
    extern __shared__ float wsum[];                              // [4 waves][3][C + 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long quads = (long)batch * P / 4;
    f32x4 g[RGBF_IT][3];
    long base[RGBF_IT];
    bool ok[RGBF_IT];
#pragma unroll
    for (int it = 0; it < RGBF_IT; ++it) {
        const long q = ((long)blockIdx.x * RGBF_IT + it) * 256 + tid;
        ok[it] = q < quads;
        const long p4 = ok[it] ? 4 * q : 0;
        const long b = p4 / P, p = p4 - b * P;
        base[it] = b * C * P + p;
        const float* gp = drgb + b * 3 * P + p;
#pragma unroll
        for (int o = 0; o < 3; ++o) g[it][o] = ok[it] ? *(const f32x4*)(gp + o * P) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    auto wave_sum = [](float v) {
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);
        return v;
    };
    const int cper = (C + (int)gridDim.y - 1) / (int)gridDim.y;
    const int cbeg = (int)blockIdx.y * cper;
    const int cend = cbeg + cper < C ? cbeg + cper : C;
    const bool with_bias = blockIdx.y + 1 == gridDim.y;
    const int cnt = cend > cbeg ? cend - cbeg : 0;
    for (int j = 0; j < cnt + (with_bias ? 1 : 0); ++j) {
        const int c = j < cnt ? cbeg + j : C;
        float a[3] = {0.0f, 0.0f, 0.0f};
        if (c < C) {
            const float w0 = w[c], w1 = w[C + c], w2 = w[2 * C + c];
            f32x4 nv[RGBF_IT], xv[RGBF_IT];
#pragma unroll
            for (int it = 0; it < RGBF_IT; ++it) {
                nv[it] = ok[it] ? *(const f32x4*)(net + base[it] + (long)c * P) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                xv[it] = (accumulate && dnet && ok[it]) ? *(const f32x4*)(dnet + base[it] + (long)c * P) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int it = 0; it < RGBF_IT; ++it) {
                f32x4 v = w0 * g[it][0] + w1 * g[it][1] + w2 * g[it][2];
                if (accumulate) v += xv[it];
                if (masked) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= nv[it][e] > 0.0f ? 1.0f : LEAK;
                }
                if (dnet && ok[it]) *(f32x4*)(dnet + base[it] + (long)c * P) = v;
#pragma unroll
                for (int o = 0; o < 3; ++o)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[o] = fmaf(g[it][o][e], nv[it][e], a[o]);
            }
        } else {
#pragma unroll
            for (int it = 0; it < RGBF_IT; ++it)
#pragma unroll
                for (int o = 0; o < 3; ++o) a[o] += (g[it][o].x + g[it][o].y) + (g[it][o].z + g[it][o].w);
        }
        if (part) {
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float t = wave_sum(a[o]);
                if (lane == 0) wsum[(wave * 3 + o) * (C + 1) + c] = t;
            }
        }
    }
    if (!part) return;
    __syncthreads();
    for (int i = tid; i < 3 * (C + 1); i += 256) {
        const int o = i / (C + 1), c = i - o * (C + 1);
        if (!((c >= cbeg && c < cend) || (with_bias && c == C))) continue;
        const float t = (wsum[(0 * 3 + o) * (C + 1) + c] + wsum[(1 * 3 + o) * (C + 1) + c]) +
                        (wsum[(2 * 3 + o) * (C + 1) + c] + wsum[(3 * 3 + o) * (C + 1) + c]);
        part[((long)c * gridDim.x + blockIdx.x) * 3 + o] = t;
    }
}

// The same pass with the NEXT step folded in: block i's backward needs g = blur^T(dhid), not dhid.  A workgroup owns a tile
// of 16 rows x 64 columns (thread: row tid / 16, four columns 4 (tid % 16) ..); per channel it computes dhid for the tile and
// its one-pixel halo (threads 0 .. 163 take one halo pixel each), parks it in LDS (two buffers: one barrier per channel) and
// applies the adjoint stencil from there with blur_kernel's weights and order -- dhid never goes to memory (saves one write
// and one read of the C/2-channel gradient per block).  Weight-gradient partials as above (centre pixels only).
constexpr int RB_TH = 16, RB_TW = 64, RB_LD = 72, RB_TILE = (RB_TH + 2) * RB_LD;      // LDS row: 3 pad + 66 used + 3

// sum over the 16 lanes of a DPP row, in every lane of the row: xor-1 / xor-2 butterflies inside the quads, then the half-row and
// row mirrors
__device__ __forceinline__ float row_sum16(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});            // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});            // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});           // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});           // row_mirror
    return v;
}
// a, b, c, d: four values that are uniform inside each 16-lane row; returns, in row r of the wave, the sum over the four rows of
// value r (v_permlane16_swap: [a0 b0 a2 b2], [a1 b1 a3 b3]; v_permlane32_swap of the two pair sums: [a01 b01 c01 d01], [a23 ...])
__device__ __forceinline__ float rows_total4(float a, float b, float c, float d) {
    auto sw16 = [](float x, float y) {
        auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
    };
    auto sw32 = [](float x, float y) {
        auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
    };
    return sw32(sw16(a, b), sw16(c, d));
}

__global__ __launch_bounds__(256) void rgb_bwd_blur_kernel(const float* __restrict__ drgb, const float* __restrict__ net, int C,
                                                           int H, int W, int batch, const float* __restrict__ w,
                                                           const float* __restrict__ dnet_in, float* __restrict__ gout,
                                                           float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wsum_floats = (12 * (C + 1) + 3) & ~3;
    float* wsum = sm;                                            // [4 waves][3][C + 1]
    float* tile = sm + wsum_floats;                              // [2][18][RB_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long P = (long)H * W;
    const int tiles_x = W / RB_TW, tiles_y = H / RB_TH;
    const int b = (int)(blockIdx.x / (unsigned)(tiles_x * tiles_y)), trem = (int)(blockIdx.x - (unsigned)b * (tiles_x * tiles_y));
    const int y00 = (trem / tiles_x) * RB_TH, x00 = (trem % tiles_x) * RB_TW;
    const int r = tid >> 4, q = tid & 15, y = y00 + r, x = x00 + 4 * q;
    const long base = (long)b * C * P + (long)y * W + x;
    f32x4 g[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) g[o] = *(const f32x4*)(drgb + ((long)b * 3 + o) * P + (long)y * W + x);
    // halo pixel of this thread (if any): top row, bottom row, left column, right column of the 18 x 66 patch
    int hy = 0, hx = 0;
    bool halo = tid < 2 * (RB_TW + 2) + 2 * RB_TH;
    if (tid < RB_TW + 2) { hy = -1; hx = tid - 1; }
    else if (tid < 2 * (RB_TW + 2)) { hy = RB_TH; hx = tid - (RB_TW + 2) - 1; }
    else if (tid < 2 * (RB_TW + 2) + RB_TH) { hy = tid - 2 * (RB_TW + 2); hx = -1; }
    else { hy = tid - 2 * (RB_TW + 2) - RB_TH; hx = RB_TW; }
    const int gy = y00 + hy, gx = x00 + hx;
    const bool hin = halo && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const long hbase = hin ? (long)b * C * P + (long)gy * W + gx : 0;
    float hg[3] = {0.0f, 0.0f, 0.0f};
    if (hin) {
#pragma unroll
        for (int o = 0; o < 3; ++o) hg[o] = drgb[((long)b * 3 + o) * P + (long)gy * W + gx];
    }
    const int hpos = (hy + 1) * RB_LD + 3 + (hx + 1);
    const int cpos = (r + 1) * RB_LD + 4 + 4 * q;                // the thread's four pixels inside the patch (16-byte aligned)
    // adjoint taps of this thread's outputs (as blur_kernel, adjoint = 1)
    float yl, yc, yr, wl[4], wc[4], wr[4];
    blur_taps(y, H, true, yl, yc, yr);
#pragma unroll
    for (int e = 0; e < 4; ++e) blur_taps(x + e, W, true, wl[e], wc[e], wr[e]);
    auto wave_sum = [](float v) {
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);
        return v;
    };
    const int cper = (C + (int)gridDim.y - 1) / (int)gridDim.y;
    const int cbeg = (int)blockIdx.y * cper;
    const int cend = cbeg + cper < C ? cbeg + cper : C;
    const bool with_bias = blockIdx.y + 1 == gridDim.y;
    // The memory operands of the next channels are requested before channel c is worked on: a workgroup's channels are a serial
    // chain (load -> LDS -> barrier -> stencil -> store).  Round 4: one channel ahead (155 -> 119 us at the 512 x 512 level).  Round 5
    // made the distance a parameter (a ring of RB_AHEAD register sets) and measured 2 and 3: SLOWER at every level (107.6 / 94.8 /
    // 44.8 us at 1, 109.0 / 99.6 / 53.8 at 2, 122.0 / 107.4 / 56.2 at 3) -- with seven workgroups per CU the kernel is not short of
    // bytes in flight; its ~150 VALU / LDS instructions per channel and wave are half of its time.
    f32x4 nvq[RB_AHEAD], dnq[RB_AHEAD];
    float hnq[RB_AHEAD], hdq[RB_AHEAD];
#pragma unroll
    for (int d = 0; d < RB_AHEAD; ++d) { nvq[d] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; dnq[d] = nvq[d]; hnq[d] = 0.0f; hdq[d] = 0.0f; }
    auto fetch = [&](int c, f32x4& nv_n, f32x4& dn_n, float& hn_n, float& hd_n) {
        nv_n = *(const f32x4*)(net + base + (long)c * P);
        if (dnet_in) dn_n = *(const f32x4*)(dnet_in + base + (long)c * P);
        if (hin) {
            hn_n = net[hbase + (long)c * P];
            if (dnet_in) hd_n = dnet_in[hbase + (long)c * P];
        }
    };
#pragma unroll
    for (int d = 0; d < RB_AHEAD; ++d)
        if (cbeg + d < cend) fetch(cbeg + d, nvq[d], dnq[d], hnq[d], hdq[d]);
    int j = 0;
    for (int c0 = cbeg; c0 < cend; c0 += RB_AHEAD) {
#pragma unroll
        for (int d = 0; d < RB_AHEAD; ++d) {
            const int c = c0 + d;
            if (c >= cend) break;                                // (uniform)
            float* tb = tile + (j & 1) * RB_TILE;
            ++j;
            const float w0 = w[c], w1 = w[C + c], w2 = w[2 * C + c];
            const f32x4 nv = nvq[d], dn = dnq[d];
            const float hn = hnq[d], hd = hdq[d];
            if (c + RB_AHEAD < cend) fetch(c + RB_AHEAD, nvq[d], dnq[d], hnq[d], hdq[d]);
            f32x4 v = w0 * g[0] + w1 * g[1] + w2 * g[2];
            if (dnet_in) v += dn;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= nv[e] > 0.0f ? 1.0f : LEAK;
            *(f32x4*)(tb + cpos) = v;
            if (halo) {
                float hv = 0.0f;
                if (hin) {
                    hv = w0 * hg[0] + w1 * hg[1] + w2 * hg[2];
                    if (dnet_in) hv += hd;
                    hv *= hn > 0.0f ? 1.0f : LEAK;
                }
                tb[hpos] = hv;
            }
            if (part) {
                float a[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int o = 0; o < 3; ++o)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[o] = fmaf(g[o][e], nv[e], a[o]);
                // the three wave sums without the LDS pipe (18 ds_bpermute per channel before): inside the 16-lane rows by DPP
                // butterflies, across the four rows by the gfx950 row / half swaps -- row o of the wave ends up with the total of a[o]
#pragma unroll
                for (int o = 0; o < 3; ++o) a[o] = row_sum16(a[o]);
                const float t = rows_total4(a[0], a[1], a[2], 0.0f);
                if ((lane & 15) == 0 && lane < 48) wsum[(wave * 3 + (lane >> 4)) * (C + 1) + c] = t;
            }
            __syncthreads();
            f32x4 rows[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float* rp = tb + (r + k) * RB_LD + 4 + 4 * q;
                const f32x4 cc = *(const f32x4*)rp;
                const float l = rp[-1], rr = rp[4];
                rows[k] = f32x4{wl[0] * l + wc[0] * cc.x + wr[0] * cc.y, wl[1] * cc.x + wc[1] * cc.y + wr[1] * cc.z,
                                wl[2] * cc.y + wc[2] * cc.z + wr[2] * cc.w, wl[3] * cc.z + wc[3] * cc.w + wr[3] * rr};
            }
            *(f32x4*)(gout + base + (long)c * P) = yl * rows[0] + yc * rows[1] + yr * rows[2];
        }
    }
    if (!part) return;
    if (with_bias) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float t = wave_sum((g[o].x + g[o].y) + (g[o].z + g[o].w));
            if (lane == 0) wsum[(wave * 3 + o) * (C + 1) + C] = t;
        }
    }
    __syncthreads();
    for (int i = tid; i < 3 * (C + 1); i += 256) {
        const int o = i / (C + 1), c = i - o * (C + 1);
        if (!((c >= cbeg && c < cend) || (with_bias && c == C))) continue;
        const float t = (wsum[(0 * 3 + o) * (C + 1) + c] + wsum[(1 * 3 + o) * (C + 1) + c]) +
                        (wsum[(2 * 3 + o) * (C + 1) + c] + wsum[(3 * 3 + o) * (C + 1) + c]);
        part[((long)c * gridDim.x + blockIdx.x) * 3 + o] = t;
    }
}

// rgb_wsum_batch_kernel: one wave per (level, channel, output): lane l adds the partials l, l + 64, ... in order, then the
// 64 lane sums are added in a fixed tree.
// Round 5: the (up to) n_blocks + 1 reductions of a backward as ONE launch at its end (each was a 5-9 us launch of mostly
// latency right behind its producer, on the stream's critical path): every level keeps its partials in its own slice of the
// scratch; per (level, channel, output) the same wave, the same order of additions -- the same bits as the per-level launches of rounds 2-4.
struct RgbWsumBatch {
    int n;
    unsigned first[UP_MAX + 2];          // job j owns blocks first[j] .. first[j + 1] - 1
    const float* part[UP_MAX + 1];
    int C[UP_MAX + 1], nwg[UP_MAX + 1];
    float* dw[UP_MAX + 1]; float* db[UP_MAX + 1];
};
__global__ __launch_bounds__(64) void rgb_wsum_batch_kernel(const RgbWsumBatch rb) {
    int j = 0;
    while (j + 1 < rb.n && blockIdx.x >= rb.first[j + 1]) ++j;
    const int i = (int)(blockIdx.x - rb.first[j]), lane = threadIdx.x;
    const int C = rb.C[j], nwg = rb.nwg[j];
    const float* part = rb.part[j];
    const int c = i / 3, o = i - 3 * c;
    float a = 0.0f;
    for (int sp = lane; sp < nwg; sp += 64) a += part[((long)c * nwg + sp) * 3 + o];
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) a += __shfl_xor(a, sft);
    if (lane == 0) {
        if (c < C) { if (rb.dw[j]) rb.dw[j][o * C + c] = a; }
        else if (rb.db[j]) rb.db[j][o] = a;
    }
}

// d(rgb) = d(img) * img * (1 - img)
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ dimg, const float* __restrict__ img,
                                                          float* __restrict__ drgb, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) drgb[i] = dimg[i] * img[i] * (1.0f - img[i]);
}

// ---------------------------------------------------------------------------------------------
// PixelShuffleUpsample tail, backward: du [B][C][2H][2W] ->
//   dpre2(b,k,p) = G(b,k,p) * (bit k&3 of sign(b,k>>2,p) ? 1 : 0.2),  G(b, 4c+2i+j, y*W+x) = du(b, c, 2y+i, 2x+j)
//   dres(b,c,p)  = sum_{q<4} G(b, c + q C, p)                                                    (x.repeat adjoint)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unshuffle_bwd_kernel(const float* __restrict__ du, const unsigned char* __restrict__ sign,
                                                            int C, int H, int W, int batch, float* __restrict__ dpre2,
                                                            float* __restrict__ dres) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const long P = (long)H * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long n1 = (long)batch * C * P;
    if (idx < n1) {
        // one out-channel c and pixel p: its 2x2 block of du is the four in-channels 4c..4c+3 (two 8-byte loads)
        const long b = idx / ((long)C * P), rem = idx - b * (long)C * P;
        const int c = (int)(rem / P);
        const long p = rem - (long)c * P;
        const int y = (int)(p / W), x = (int)(p - (long)y * W);
        const float* src = du + (b * C + c) * 4 * P + (long)(2 * y) * (2 * W) + 2 * x;
        const f32x2 top = *(const f32x2*)src, bot = *(const f32x2*)(src + 2 * W);
        const unsigned nib = sign[idx];
        float* dst = dpre2 + (b * 4 * C + 4 * c) * P + p;
        dst[0] = top.x * ((nib & 1) ? 1.0f : LEAK);
        dst[P] = top.y * ((nib & 2) ? 1.0f : LEAK);
        dst[2 * P] = bot.x * ((nib & 4) ? 1.0f : LEAK);
        dst[3 * P] = bot.y * ((nib & 8) ? 1.0f : LEAK);
    } else if (idx < 2 * n1) {
        // x.repeat adjoint: d(x)(b,c,p) = sum_q G(b, c + q C, p), G(b, k, p) = du(b, k>>2, 2y + ((k>>1)&1), 2x + (k&1))
        const long e = idx - n1;
        const long b = e / ((long)C * P), rem = e - b * (long)C * P;
        const int c = (int)(rem / P);
        const long p = rem - (long)c * P;
        const int y = (int)(p / W), x = (int)(p - (long)y * W);
        auto G = [&](int k) {
            return du[(b * C + (k >> 2)) * 4 * P + (long)(2 * y + ((k >> 1) & 1)) * (2 * W) + 2 * x + (k & 1)];
        };
        dres[e] = (G(c) + G(c + C)) + (G(c + 2 * C) + G(c + 3 * C));
    }
}

// The same for C % 4 == 0 (the 64-channel block at 256 x 256, the largest of the three): in-channel c' + q C is element
// c' % 4 of the 2x2 block of out-channel c'/4 + q C/4, so ONE thread that walks the four out-channels cb + q C/4 of four
// consecutive pixels has every term of dres(4 cb + e) in registers -- du is read once instead of twice, all accesses are
// 16-byte vectors.  Same summation order as above.
__global__ __launch_bounds__(256) void unshuffle_bwd4_kernel(const float* __restrict__ du, const unsigned char* __restrict__ sign,
                                                             int C, int H, int W, int batch, float* __restrict__ dpre2,
                                                             float* __restrict__ dres) {
    const long P = (long)H * W, nq = P / 4;
    const int Cq = C / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * Cq * nq) return;
    const long b = idx / ((long)Cq * nq), rem = idx - b * (long)Cq * nq;
    const int cb = (int)(rem / nq);
    const long p = 4 * (rem - (long)cb * nq);
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    f32x4 G[4][4];                                   // [q][e] over the 4 pixels
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cb + q * Cq;
        const float* src = du + (b * C + c) * 4 * P + (long)(2 * y) * (2 * W) + 2 * x;
        const f32x4 t0 = *(const f32x4*)src, t1 = *(const f32x4*)(src + 4);
        const f32x4 b0 = *(const f32x4*)(src + 2 * W), b1 = *(const f32x4*)(src + 2 * W + 4);
        G[q][0] = f32x4{t0.x, t0.z, t1.x, t1.z};
        G[q][1] = f32x4{t0.y, t0.w, t1.y, t1.w};
        G[q][2] = f32x4{b0.x, b0.z, b1.x, b1.z};
        G[q][3] = f32x4{b0.y, b0.w, b1.y, b1.w};
        const unsigned nib4 = *(const unsigned*)(sign + (b * C + c) * P + p);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = G[q][e][t] * (((nib4 >> (8 * t + e)) & 1u) ? 1.0f : LEAK);
            *(f32x4*)(dpre2 + (b * 4 * C + 4 * c + e) * P + p) = v;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        *(f32x4*)(dres + (b * C + 4 * cb + e) * P + p) = (G[0][e] + G[1][e]) + (G[2][e] + G[3][e]);
}

// x.repeat adjoint from the MASKED gradient (round 4: when the du GEMM's epilogue has already written dpre2 and the channel
// count is no multiple of 4, so that the four terms of an output come from four different row slices of that GEMM):
//   dres(b,c,p) = sum_q G(b, c + q C, p),   G(b,k,p) = dpre2(b,k,p) * (bit k&3 of sign(b,k>>2,p) ? 1 : 1/0.2)
// The un-masking multiplies by 5.0f where the mask multiplied by 0.2f: one rounding (<= 1 ulp of that term) away from the
// two-kernel path, which sums the unmasked du.  Same order of the four terms.  A thread = 4 consecutive pixels.
__global__ __launch_bounds__(256) void unshuffle_dres_kernel(const float* __restrict__ dpre2, const unsigned char* __restrict__ sign,
                                                             int C, long P, int batch, float* __restrict__ dres) {
    const long nq = P / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * C * nq) return;
    const long b = idx / ((long)C * nq), rem = idx - b * (long)C * nq;
    const int c = (int)(rem / nq);
    const long p = 4 * (rem - (long)c * nq);
    f32x4 G[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = c + q * C;
        const f32x4 v = *(const f32x4*)(dpre2 + (b * 4 * C + k) * P + p);
        const unsigned nib4 = *(const unsigned*)(sign + (b * C + (k >> 2)) * P + p);
#pragma unroll
        for (int t = 0; t < 4; ++t) G[q][t] = v[t] * (((nib4 >> (8 * t + (k & 3))) & 1u) ? 1.0f : 1.0f / LEAK);
    }
    *(f32x4*)(dres + (b * C + c) * P + p) = (G[0] + G[1]) + (G[2] + G[3]);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct UpDims {
    int n_blocks, ch[UP_MAX + 1], side[UP_MAX + 1];
};

static int up_dims(const GnrUpsampleProblem* p, UpDims* d) {
    if (!p) return fail("gnr_upsample: problem is NULL");
    if (p->struct_size != sizeof(GnrUpsampleProblem))
        return fail("gnr_upsample: GnrUpsampleProblem.struct_size is %u but this libgnr.so (ABI %d) has sizeof = %zu: the caller "
                    "was built against a different include/gnr.h (or did not set struct_size)", p->struct_size, GNR_ABI_VERSION,
                    sizeof(GnrUpsampleProblem));
    if (p->batch < 1 || p->feat_nc < 1) return fail("gnr_upsample: batch and feat_nc must be >= 1");
    if (p->n_blocks < 1 || p->n_blocks > UP_MAX) return fail("gnr_upsample: n_blocks must be 1..%d", UP_MAX);
    if (p->featmap_size < 16 || (p->featmap_size & (p->featmap_size - 1)))
        return fail("gnr_upsample: featmap_size must be a power of two >= 16 (got %d)", p->featmap_size);
    if (p->min_feat < 1) return fail("gnr_upsample: min_feat must be >= 1");
    if (p->feat_nc < p->min_feat)
        return fail("gnr_upsample: feat_nc (%d) < min_feat (%d): the reference's NeuralRenderer builds its first block for "
                    "max(feat_nc, min_feat) input channels and cannot run this configuration either", p->feat_nc, p->min_feat);
    if (!p->x) return fail("gnr_upsample: x is NULL");
    d->n_blocks = p->n_blocks;
    for (int i = 0; i <= p->n_blocks; ++i) {
        const int c = p->feat_nc >> i;
        d->ch[i] = c > p->min_feat ? c : p->min_feat;       // max(feat_nc // 2^i, min_feat), neural_renderer.py:60-97
        d->side[i] = p->featmap_size << i;
    }
    // limits of the kernels: rgb_bwd_fused_kernel keeps 12 (C + 1) floats in LDS; the GEMMs address an image's operand
    // (up to 4 C channels x P pixels, or C x 4 P) through one 32-bit buffer descriptor
    if (p->feat_nc > 1024) return fail("gnr_upsample: feat_nc = %d exceeds the kernels' limit of 1024 channels", p->feat_nc);
    for (int i = 0; i < p->n_blocks; ++i)
        if (16L * d->ch[i] * d->side[i] * d->side[i] >= (1L << 31))
            return fail("gnr_upsample: block %d (%d channels at %d x %d) exceeds the 2 GiB per-image operand limit of the GEMM "
                        "kernels", i, d->ch[i], d->side[i], d->side[i]);
    return 0;
}

// The three 1x1 convolutions of a block, forward or backward: tile variants + offsets of their packed weights
struct BlockPlans {
    Conv16Plan c1, c2, c3;
    long o1, o2, o3;
    bool blur_fused;                    // forward only: feat_layers reads blur(u) on the fly
    bool unshuffle_fused;               // backward only: the du GEMM's epilogue is the PixelShuffleUpsample tail's adjoint
    UpChainPlan chain;                  // forward only, T1 != 0: layer_1 -> layer_2 as ONE kernel;
    long oc1, oc2;                      // c1 / c2 / o1 / o2 are then unused
};

// Forward GEMMs: a1 = W1 net (2C x C), u = shuffle(W2 a1) (4C x 2C), net' = Wf blur(u) (Cn x C over 4P pixels).
// w == NULL: sizes only.  Returns the floats of the packed operands.
static size_t plan_fwd(const UpDims& d, int B, const GnrUpsampleWeights* w, BlockPlans* bp, Conv16PackJobs* jobs) {
    Conv16PackJobs J{};
    for (int i = 0; i < d.n_blocks; ++i) {
        const int C = d.ch[i], Cn = d.ch[i + 1], S = d.side[i];
        const long px = (long)B * S * S;
        BlockPlans q{};
        // forward chain: a1 = W1 net (K1 = C -> M1 = 2C), u = shuffle(W2 a1) (M2 = 4C)
        q.chain = upchain_plan(C, 2 * C, 4 * C, (long)S * S);
        if (q.chain.T1) {
            upchain_add_jobs(J, q.chain, w ? w->up1_w[i] : nullptr, C, 1, 2 * C, C, w ? w->up2_w[i] : nullptr, 2 * C, 1, 4 * C, &q.oc1, &q.oc2);
        } else {
            q.c1 = conv16_plan(2 * C, C, px, 0);
            q.o1 = conv16_add_job(J, w ? w->up1_w[i] : nullptr, C, 1, 2 * C, C, q.c1);
            q.c2 = conv16_plan(4 * C, 2 * C, px, 0);
            q.o2 = conv16_add_job(J, w ? w->up2_w[i] : nullptr, 2 * C, 1, 4 * C, 2 * C, q.c2);
        }
        q.c3 = conv16_plan(Cn, C, 4 * px, 2 * S);
        q.blur_fused = q.c3.MT != 0;
        if (!q.blur_fused) q.c3 = conv16_plan(Cn, C, 4 * px, 0);
        q.o3 = conv16_add_job(J, w ? w->feat_w[i] : nullptr, C, 1, Cn, C, q.c3);
        if (bp) bp[i] = q;
    }
    size_t total = 0;
    for (int i = 0; i < J.n; ++i) total += (size_t)J.j[i].floats;
    if (jobs) *jobs = J;
    return total;
}

// Backward GEMMs (A = W^T through strides): du = Wf^T g (C x Cn over 4P), dpre1 = W2^T dpre2 (2C x 4C), dnet += W1^T dpre1
// (C x 2C).
static size_t plan_bwd(const UpDims& d, int B, const GnrUpsampleWeights* w, BlockPlans* bp, Conv16PackJobs* jobs) {
    Conv16PackJobs J{};
    for (int i = 0; i < d.n_blocks; ++i) {
        const int C = d.ch[i], Cn = d.ch[i + 1], S = d.side[i];
        const long px = (long)B * S * S;
        BlockPlans q{};
        q.c3 = conv16_plan_unshuffle(C, Cn, S);
        q.unshuffle_fused = q.c3.MT != 0;
        if (!q.unshuffle_fused) q.c3 = conv16_plan(C, Cn, 4 * px, 0);
        q.o3 = conv16_add_job(J, w ? w->feat_w[i] : nullptr, 1, C, C, Cn, q.c3, q.unshuffle_fused && C % 4 == 0 ? 1 : 0);
        q.c2 = conv16_plan(2 * C, 4 * C, px, 0);
        q.o2 = conv16_add_job(J, w ? w->up2_w[i] : nullptr, 1, 2 * C, 2 * C, 4 * C, q.c2);
        q.c1 = conv16_plan(C, 2 * C, px, 0);
        q.o1 = conv16_add_job(J, w ? w->up1_w[i] : nullptr, 1, C, C, 2 * C, q.c1);
        if (bp) bp[i] = q;
    }
    size_t total = 0;
    for (int i = 0; i < J.n; ++i) total += (size_t)J.j[i].floats;
    if (jobs) *jobs = J;
    return total;
}

struct UpSaved {                        // kept for the backward
    float* a1[UP_MAX];                  // [B][2C][P]
    unsigned char* sign2[UP_MAX];       // [B][C][P]    four pre-activation sign bits per (out-channel quad, pixel)
    float* pack;                        // packed A operands of every GEMM of the call in flight (forward, then backward)
    size_t pack_floats;                 // its carved size: the plans are made again after the carve (gnr_set_conv16_tile is
                                        // process-wide and may change in between) and must fit
    float* u[UP_MAX];                   // [B][C][4P]   shuffled map (pre-blur): read by the backward's dWf GEMM, never written by it
    float* net[UP_MAX];                 // [B][C'][4P]  block output
    float* img;                         // [B][3][Pn]
    float* vtmp;                        // [B][C][4P]   blurred map, only for blocks whose feat GEMM cannot blur on the fly
    float* rgb_a; float* rgb_b;         // [B][3][Pn]   scratch
};

static size_t up_carve(const GnrUpsampleProblem* p, const UpDims& d, char* base, UpSaved* s) {
    size_t off = 0;
    int region = 0;
    auto take = [&](size_t bytes) {
        char* q = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        if (CANARY_BYTES) {                                   // experimental builds (gnr_canary.h): a gap behind every region
            if (base) canary_note(base + off, "up_carve", region);
            off += CANARY_BYTES;
        }
        ++region;
        return q;
    };
    UpSaved z{};
    const size_t B = (size_t)p->batch;
    BlockPlans bp[UP_MAX];
    const size_t pk_f = plan_fwd(d, p->batch, nullptr, bp, nullptr), pk_b = plan_bwd(d, p->batch, nullptr, nullptr, nullptr);
    size_t vmax = 0;
    for (int i = 0; i < d.n_blocks; ++i) {
        const size_t C = d.ch[i], Cn = d.ch[i + 1], P = (size_t)d.side[i] * d.side[i];
        z.a1[i] = (float*)take(B * 2 * C * P * 4);
        z.sign2[i] = (unsigned char*)take(B * C * P);
        z.u[i] = (float*)take(B * C * 4 * P * 4);
        z.net[i] = (float*)take(B * Cn * 4 * P * 4);
        if (!bp[i].blur_fused && B * C * 4 * P * 4 > vmax) vmax = B * C * 4 * P * 4;
    }
    const size_t Pn = (size_t)d.side[d.n_blocks] * d.side[d.n_blocks];
    z.img = (float*)take(B * 3 * Pn * 4);
    z.pack_floats = pk_f > pk_b ? pk_f : pk_b;
    z.pack = (float*)take(z.pack_floats * 4);
    z.vtmp = (float*)take(vmax);
    z.rgb_a = (float*)take(B * 3 * Pn * 4);
    z.rgb_b = (float*)take(B * 3 * Pn * 4);
    if (s) *s = z;
    return off;
}

struct UpScratch {                      // backward temporaries
    float* g0; float* g1;               // two buffers of the largest activation size
    float* g2;                          // same size: d(net) alternates between g2 and g1, so the
                                        // saved forward state stays intact and a second backward (retain_graph) is valid
    float* drgb_a; float* drgb_b;       // [B][3][Pn]
    float* colsum;                      // the RGB-branch backward's partial sums, level i at colsum + colsum_off[i]
    size_t colsum_off[UP_MAX + 1];
    float* wg;                          // partial tiles of every weight-gradient GEMM of the call (gnr_wgrad.h)
};

static size_t up_colsum_floats(const UpDims& d, int batch, int level) {
    const size_t px = (size_t)batch * d.side[level] * d.side[level];
    size_t wgs = (size_t)rgbf_workgroups((long)px);
    if (px / (RB_TH * RB_TW) > wgs) wgs = px / (RB_TH * RB_TW);       // rgb_bwd_blur_kernel: one workgroup per 16 x 64 tile
    return (((size_t)(d.ch[level] + 1) * wgs * 3) + 63) & ~(size_t)63;
}

// The arena of the call's queued weight-gradient GEMMs: wgrad_arena_floats()'s bound raised to the largest single need of the three
// products per block (dW feat_layers at the block's output size, dW layer_2, dW layer_1) as launch_wgrad_img's LDS-staged route
// plans them -- the route a product takes when the register-fed kernel (gnr_wgrad16.hip, which fits its split count to the arena)
// declines it.  Sized from the launchers' own plan; tests/test_host_logic.py sweeps the batch on the CPU.
static size_t up_wgrad_arena_floats(const GnrUpsampleProblem* p, const UpDims& d) {
    WgradShape g[3 * UP_MAX];
    int n = 0;
    for (int i = 0; i < d.n_blocks; ++i) {
        const int C = d.ch[i], Cn = d.ch[i + 1];
        const long P = (long)d.side[i] * d.side[i], P4 = 4 * P;
        g[n++] = WgradShape{Cn, Cn, C, C, P4 / CHUNK, P4, 0, 0, 0};
        g[n++] = WgradShape{4 * C, 4 * C, 2 * C, 2 * C, P / CHUNK, P, 0, 0, 0};
        g[n++] = WgradShape{2 * C, 2 * C, C, C, P / CHUNK, P, 0, 0, 0};
    }
    return wgrad_arena_floats_for(g, n, p->batch, 4 * d.ch[0], 2 * d.ch[0]);
}

static size_t up_carve_bwd(const GnrUpsampleProblem* p, const UpDims& d, char* base, UpScratch* s) {
    size_t off = 0;
    int region = 0;
    auto take = [&](size_t bytes) {
        char* q = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        if (CANARY_BYTES) {
            if (base) canary_note(base + off, "up_carve_bwd", region);
            off += CANARY_BYTES;
        }
        ++region;
        return q;
    };
    size_t big = 0;
    const size_t B = (size_t)p->batch;
    for (int i = 0; i < d.n_blocks; ++i) {
        const size_t C = d.ch[i], P = (size_t)d.side[i] * d.side[i];
        if (B * 4 * C * P * 4 > big) big = B * 4 * C * P * 4;       // dpre2 == du size
    }
    const size_t Pn = (size_t)d.side[d.n_blocks] * d.side[d.n_blocks];
    UpScratch z{};
    z.g0 = (float*)take(big);
    z.g1 = (float*)take(big);
    z.g2 = (float*)take(big);        // d(net) of block i+1 becomes block i's X (dhid, du, dpre1: up to 4 C P floats)
    z.drgb_a = (float*)take(B * 3 * Pn * 4);
    z.drgb_b = (float*)take(B * 3 * Pn * 4);
    size_t cs_floats = 0;
    for (int i = 0; i <= d.n_blocks; ++i) {           // the RGB-branch backward's partials, one slice per level: (channels + 1) x workgroups x 3
        z.colsum_off[i] = cs_floats;
        cs_floats += up_colsum_floats(d, p->batch, i);
    }
    z.colsum = (float*)take(cs_floats * 4);
    z.wg = (float*)take(up_wgrad_arena_floats(p, d) * 4);      // layer_2 of block 0 is the largest product
    if (s) *s = z;
    return off;
}

static inline unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }

static int check_up_weights(const GnrUpsampleWeights* w, int n_blocks) {
    if (!w) return fail("gnr_upsample: weights are NULL");
    for (int i = 0; i < n_blocks; ++i)
        if (!w->up1_w[i] || !w->up1_b[i] || !w->up2_w[i] || !w->up2_b[i] || !w->feat_w[i] || !w->feat_b[i])
            return fail("gnr_upsample: a block-%d weight pointer is NULL", i);
    for (int i = 0; i <= n_blocks; ++i)
        if (!w->rgb_w[i] || !w->rgb_b[i]) return fail("gnr_upsample: feat_2_rgb_list.%d pointer is NULL", i);
    return 0;
}

static void launch_rgb_conv(const float* net, int C, long P, int batch, const float* w, const float* bias, float* rgb,
                            int accumulate, float* img, float* out, hipStream_t st) {
    if ((long)batch * P <= 32768 && C >= 32)
        hipLaunchKernelGGL(rgb_conv_split_kernel, dim3((unsigned)(((long)batch * P + 63) / 64)), dim3(256), 0, st, net, C, P, batch, w,
                           bias, rgb, accumulate, img, out);
    else
        hipLaunchKernelGGL(rgb_conv_kernel, dim3(blocks_for((long)batch * P)), dim3(256), 0, st, net, C, P, batch, w, bias, rgb,
                           accumulate, img, out);
}

// out <- blur(bilinear2x(in)) for 3-channel images at side S -> 2S, one kernel (round 4; two before)
static void up_rgb(const float* in, float* out, int batch, int S, hipStream_t st) {       // in != out
    const long planes = (long)batch * 3;
    hipLaunchKernelGGL(bilinear_blur_kernel, dim3(blocks_for(planes * 4L * S * S / 4)), dim3(256), 0, st, in, out, planes, S, S);
}

}  // namespace gnr

using namespace gnr;

extern "C" {

size_t gnr_upsample_workspace_bytes(const GnrUpsampleProblem* p, int kind) {
    UpDims d;
    if (up_dims(p, &d)) return 0;
    if (kind == GNR_UP_WS_FWD) return up_carve(p, d, nullptr, nullptr);
    if (kind == GNR_UP_WS_BWD) return up_carve_bwd(p, d, nullptr, nullptr);
    fail("gnr_upsample_workspace_bytes: unknown kind %d", kind);
    return 0;
}

int gnr_upsample_fwd(const GnrUpsampleProblem* p, const GnrUpsampleWeights* w, float* img, void* workspace,
                     size_t ws_bytes, void* stream) {
    UpDims d;
    if (up_dims(p, &d)) return 1;
    if (check_up_weights(w, d.n_blocks)) return 1;
    if (!img) return fail("gnr_upsample_fwd: output image is NULL");
    const size_t need = up_carve(p, d, nullptr, nullptr);
    if (!workspace || ws_bytes < need) return fail("gnr_upsample_fwd: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    if ((uintptr_t)workspace & 255) return fail("gnr_upsample_fwd: workspace must be 256-byte aligned");
    UpSaved s;
    hipStream_t st = (hipStream_t)stream;
    canary_begin(true);
    up_carve(p, d, (char*)workspace, &s);
    canary_arm(st);
    const int B = p->batch;

    // rgb = up(conv_rgb0(x))
    {
        const long P = (long)d.side[0] * d.side[0];
        launch_rgb_conv(p->x, d.ch[0], P, B, w->rgb_w[0], w->rgb_b[0], s.rgb_a, 0, nullptr, nullptr, st);
        up_rgb(s.rgb_a, s.rgb_b, B, d.side[0], st);
    }
    BlockPlans bp[UP_MAX];
    Conv16PackJobs jobs;
    if (plan_fwd(d, B, w, bp, &jobs) > s.pack_floats)
        return fail("gnr_upsample_fwd: gnr_set_conv16_tile changed while the call was being planned");
    jobs.dst = s.pack;
    launch_conv16_pack(jobs, st);                 // every weight matrix of the call, one launch
    const float* net = p->x;
    float* rgb = s.rgb_b;             // the running RGB image ping-pongs between the two scratch images
    float* rgb_other = s.rgb_a;
    for (int i = 0; i < d.n_blocks; ++i) {
        const int C = d.ch[i], Cn = d.ch[i + 1], S = d.side[i];
        const long P = (long)S * S;
        Conv16Params g{};
        if (bp[i].chain.T1) {
            // round 4: both layers in one kernel -- a1 is written (the backward reads it) but never read back
            UpChainParams c{};
            c.plan = bp[i].chain; c.A1 = s.pack + bp[i].oc1; c.A2 = s.pack + bp[i].oc2; c.B = net; c.b_batch = (long)C * P;
            c.K1 = C; c.M1 = 2 * C; c.M2 = 4 * C; c.P = (int)P; c.batch = B;
            c.bias1 = w->up1_b[i]; c.out1 = s.a1[i]; c.out1_batch = 2L * C * P;
            c.bias2 = w->up2_b[i]; c.res = net; c.res_batch = (long)C * P; c.sign_out = s.sign2[i]; c.sign_batch = (long)C * P;
            c.out2 = s.u[i]; c.out2_batch = 4L * C * P; c.W = S;
            if (launch_upchain(c, st)) return 1;
        } else {
            // a1 = lrelu(W1 net + b1)
            g.plan = bp[i].c1; g.At = s.pack + bp[i].o1; g.B = net; g.b_batch = (long)C * P; g.C = s.a1[i]; g.c_batch = 2L * C * P;
            g.M = 2 * C; g.K = C; g.P = (int)P; g.batch = B; g.bias = w->up1_b[i]; g.leaky = 1;
            if (launch_conv16(g, st)) return 1;
            // u = pixel_shuffle(lrelu(W2 a1 + b2) + repeat(net))
            g = Conv16Params{};
            g.plan = bp[i].c2; g.At = s.pack + bp[i].o2; g.B = s.a1[i]; g.b_batch = 2L * C * P; g.C = s.u[i]; g.c_batch = 4L * C * P;
            g.M = 4 * C; g.K = 2 * C; g.P = (int)P; g.batch = B; g.bias = w->up2_b[i]; g.leaky = 1; g.shuffle = 1; g.W = S;
            g.res = net; g.res_batch = (long)C * P; g.sign_out = s.sign2[i]; g.sign_batch = (long)C * P;
            if (launch_conv16(g, st)) return 1;
        }
        // net' = lrelu(Wf blur(u) + bf): the stencil inside the GEMM's operand load, or as its own kernel
        g = Conv16Params{};
        g.plan = bp[i].c3; g.At = s.pack + bp[i].o3; g.B = s.u[i]; g.b_batch = 4L * C * P; g.C = s.net[i]; g.c_batch = 4L * Cn * P;
        g.M = Cn; g.K = C; g.P = (int)(4 * P); g.batch = B; g.bias = w->feat_b[i]; g.leaky = 1;
        if (bp[i].blur_fused) {
            g.blur = 1; g.W = 2 * S; g.H = 2 * S;
        } else {
            hipLaunchKernelGGL(blur_kernel, dim3(blocks_for((long)B * C * P)), dim3(256), 0, st, s.u[i], s.vtmp, (long)B * C, 2 * S,
                               2 * S, 0);
            g.B = s.vtmp;
        }
        // rgb += conv_rgb(i+1)(net');  last block: img = sigmoid(rgb) (or rgb itself), also straight into the caller's image.
        // Round 4: when one wave holds every output channel of its pixels (one row slice: 129 / 64 / 32 channels) the three
        // dots ride on the GEMM's epilogue; otherwise rgb_conv_kernel re-reads the block output.
        const bool last = i == d.n_blocks - 1;
        const bool rgb_fused = bp[i].c3.slices == 1;
        if (rgb_fused) {
            g.rgb_w = w->rgb_w[i + 1]; g.rgb_bias = w->rgb_b[i + 1]; g.rgb = rgb; g.rgb_accumulate = 1;
            g.rgb_img = last && p->final_sigmoid ? s.img : nullptr; g.rgb_out = last ? img : nullptr;
        }
        if (launch_conv16(g, st)) return 1;
        if (!rgb_fused)
            launch_rgb_conv(s.net[i], Cn, 4 * P, B, w->rgb_w[i + 1], w->rgb_b[i + 1], rgb, 1, last && p->final_sigmoid ? s.img : nullptr,
                            last ? img : nullptr, st);
        if (!last) {
            up_rgb(rgb, rgb_other, B, 2 * S, st);
            float* t = rgb; rgb = rgb_other; rgb_other = t;
        }
        net = s.net[i];
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("gnr_upsample_fwd: launch failed: %s", hipGetErrorString(e));
    if (canary_check(st, "gnr_upsample_fwd")) return 1;
    return 0;
}

int gnr_upsample_bwd(const GnrUpsampleProblem* p, const GnrUpsampleWeights* w, const float* d_img, float* d_x,
                     const GnrUpsampleWeightGrads* dw, void* saved, size_t saved_bytes, void* scratch,
                     size_t scratch_bytes, void* stream) {
    UpDims d;
    if (up_dims(p, &d)) return 1;
    if (check_up_weights(w, d.n_blocks)) return 1;
    if (!d_img) return fail("gnr_upsample_bwd: d_img is NULL");
    const size_t need_s = up_carve(p, d, nullptr, nullptr), need_t = up_carve_bwd(p, d, nullptr, nullptr);
    if (!saved || saved_bytes < need_s) return fail("gnr_upsample_bwd: saved workspace too small (%zu < %zu bytes)", saved_bytes, need_s);
    if (!scratch || scratch_bytes < need_t) return fail("gnr_upsample_bwd: scratch too small (%zu < %zu bytes)", scratch_bytes, need_t);
    if (((uintptr_t)saved & 255) || ((uintptr_t)scratch & 255)) return fail("gnr_upsample_bwd: workspaces must be 256-byte aligned");
    UpSaved s;
    hipStream_t st = (hipStream_t)stream;
    canary_begin(false);                    // the saved workspace's gaps were filled by gnr_upsample_fwd: checked, never refilled
    up_carve(p, d, (char*)saved, &s);
    UpScratch t;
    canary_fill_mode(true);
    up_carve_bwd(p, d, (char*)scratch, &t);
    canary_arm(st);
    GnrUpsampleWeightGrads G{};
    if (dw) G = *dw;
    const int B = p->batch, nb = d.n_blocks;
    const long Pn = (long)d.side[nb] * d.side[nb];

    // d(rgb) at full resolution
    float* drgb = t.drgb_a;
    float* drgb_tmp = t.drgb_b;
    if (p->final_sigmoid)
        hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(blocks_for((long)B * 3 * Pn)), dim3(256), 0, st, d_img, s.img, drgb, (long)B * 3 * Pn);
    else
        (void)hipMemcpyAsync(drgb, d_img, (size_t)B * 3 * Pn * 4, hipMemcpyDeviceToDevice, st);

    BlockPlans bp[UP_MAX];
    Conv16PackJobs jobs;
    if (plan_bwd(d, B, w, bp, &jobs) > s.pack_floats)
        return fail("gnr_upsample_bwd: gnr_set_conv16_tile changed while the call was being planned");
    jobs.dst = s.pack;
    launch_conv16_pack(jobs, st);                 // every transposed weight matrix of the call, one launch

    // the RGB branch's weight-gradient partials of every level are reduced by ONE launch at the end of the call
    RgbWsumBatch wsum{};
    auto wsum_push = [&](const float* part, int C, int nwg, float* dwp, float* dbp) {
        const int j = wsum.n++;
        wsum.part[j] = part; wsum.C[j] = C; wsum.nwg[j] = nwg; wsum.dw[j] = dwp; wsum.db[j] = dbp;
        wsum.first[j + 1] = wsum.first[j] + 3u * (unsigned)(C + 1);
    };
    // the split-K reductions of the nine weight-gradient GEMMs are queued and run as ONE launch at the end of the call
    WgradDefer wd;
    wgrad_defer_init(&wd, t.wg, up_wgrad_arena_floats(p, d));

    // Blur and the 1x1 convolution act on different axes (pixels / channels) and commute: with g = blur^T(dhid)
    //   dWf = g u^T,  dbf = sum g (= sum dhid: blur's rows sum to 1),  du = Wf^T g,
    // so the adjoint stencil runs on the C/2 channels of dhid instead of the C channels of dv, and the forward never has to
    // write blur(u).
    float* dnet_next = nullptr;       // gradient w.r.t. net' of block i coming from block i+1 (its input)
    for (int i = nb - 1; i >= 0; --i) {
        const int C = d.ch[i], Cn = d.ch[i + 1], S = d.side[i];
        const long P = (long)S * S, P4 = 4 * P;
        const float* net_in = i == 0 ? p->x : s.net[i - 1];
        // the RGB branch at this resolution: rgb_i = up(rgb_{i-1}) + conv(net') ...; undo the up() that FOLLOWED block i
        if (i < nb - 1) {
            // drgb currently is at side 4S (block i+1's resolution): adjoint of blur o bilinear
            hipLaunchKernelGGL(blur_bilinear_adj_kernel, dim3(blocks_for((long)B * 3 * P4)), dim3(256), 0, st, drgb, drgb_tmp, (long)B * 3,
                               2 * S, 2 * S);
            float* sw = drgb; drgb = drgb_tmp; drgb_tmp = sw;
        }
        // conv_rgb(i+1): weight / bias gradients and dhid = (dnet' + Wr^T drgb) * lrelu'(net'), one pass over net'
        float* X = dnet_next ? dnet_next : t.g0;       // dhid, later du, later dpre1
        float* Y = X == t.g0 ? t.g1 : t.g0;            // g, later dpre2
        {
            const bool want_w = G.rgb_w[i + 1] || G.rgb_b[i + 1];
            if ((2 * S) % RB_TW == 0 && (2 * S) % RB_TH == 0) {
                // ... and g = blur^T dhid in the same pass (dhid stays in LDS)
                const unsigned nwg = (unsigned)(B * (2 * S / RB_TH) * (2 * S / RB_TW));
                const size_t lds = ((size_t)((12 * (Cn + 1) + 3) & ~3) + 2 * RB_TILE) * sizeof(float);
                hipLaunchKernelGGL(rgb_bwd_blur_kernel, dim3(nwg, rgbf_channel_groups(nwg, Cn)), dim3(256), lds, st, drgb, s.net[i], Cn,
                                   2 * S, 2 * S, B, w->rgb_w[i + 1], dnet_next ? X : (const float*)nullptr, Y,
                                   want_w ? t.colsum + t.colsum_off[i + 1] : (float*)nullptr);
                if (want_w) wsum_push(t.colsum + t.colsum_off[i + 1], Cn, (int)nwg, G.rgb_w[i + 1], G.rgb_b[i + 1]);
            } else {
                const unsigned nwg = (unsigned)rgbf_workgroups((long)B * P4);
                hipLaunchKernelGGL(rgb_bwd_fused_kernel, dim3(nwg, rgbf_channel_groups(nwg, Cn)), dim3(256), (size_t)12 * (Cn + 1) * sizeof(float), st, drgb, s.net[i], Cn,
                                   P4, B, w->rgb_w[i + 1], X, dnet_next ? 1 : 0, 1, want_w ? t.colsum + t.colsum_off[i + 1] : (float*)nullptr);
                if (want_w) wsum_push(t.colsum + t.colsum_off[i + 1], Cn, (int)nwg, G.rgb_w[i + 1], G.rgb_b[i + 1]);
                // g = blur^T dhid
                hipLaunchKernelGGL(blur_kernel, dim3(blocks_for((long)B * Cn * P)), dim3(256), 0, st, X, Y, (long)B * Cn, 2 * S, 2 * S, 1);
            }
        }
        // feat_layers[i]: dWf = g u^T, dbf; du = Wf^T g
        launch_wgrad_img(Y, Cn, Cn, s.u[i], C, C, B, P4, G.feat_w[i], C, G.feat_b[i], 0, t.wg, st, &wd);
        // un-shuffle: dpre2 and the residual part of d(net_in) into backward scratch (block 0: straight into the
        // caller's d_x).  X is g0 (last block) or the previous d(net), Y the other of g0 / g1: d(net) takes g2, g1, g2, ...
        // -- never a buffer of the saved forward workspace (round 3 wrote it over u[i], which made a second backward over
        // the same saved state wrong).
        float* dnet = (i == 0 && d_x) ? d_x : (((nb - 1 - i) & 1) ? t.g1 : t.g2);
        bool dres_folded = false;
        Conv16Params g{};
        g.plan = bp[i].c3; g.At = s.pack + bp[i].o3; g.B = Y; g.b_batch = (long)Cn * P4; g.M = C; g.K = Cn; g.P = (int)P4; g.batch = B;
        if (bp[i].unshuffle_fused) {
            // round 4: du never reaches memory -- the GEMM's epilogue writes dpre2 (-> X; Y = g is still being read) and dres
            g.C = X; g.c_batch = 4L * C * P; g.W = S; g.sign_in = s.sign2[i]; g.sign_batch = (long)C * P;
            if (C % 4 == 0) { g.dres = dnet; g.dres_batch = (long)C * P; }
            if (launch_conv16(g, st)) return 1;
            // C % 4 != 0: the x.repeat adjoint's terms sit in different row slices of this GEMM.  Round 4 collected them from dpre2
            // with unshuffle_dres_kernel; round 5: layer_1's data-gradient GEMM below (which accumulated into that kernel's output)
            // collects them in its own epilogue when it runs a 64-pixel tile -- d(net) is written once and never read back
            dres_folded = C % 4 != 0 && bp[i].c1.NT == 4;
            if (C % 4 && !dres_folded)
                hipLaunchKernelGGL(unshuffle_dres_kernel, dim3(blocks_for((long)B * C * (P / 4))), dim3(256), 0, st, X, s.sign2[i], C, P, B, dnet);
            float* sw = X; X = Y; Y = sw;                                        // Y = dpre2, X free for dpre1
        } else {
            g.C = X; g.c_batch = (long)C * P4;
            if (launch_conv16(g, st)) return 1;                                  // X = du
            if (C % 4 == 0)
                hipLaunchKernelGGL(unshuffle_bwd4_kernel, dim3(blocks_for((long)B * (C / 4) * (P / 4))), dim3(256), 0, st, X, s.sign2[i], C, S, S,
                                   B, Y, dnet);
            else
                hipLaunchKernelGGL(unshuffle_bwd_kernel, dim3(blocks_for((long)B * 2 * C * P)), dim3(256), 0, st, X, s.sign2[i], C, S, S, B, Y,
                                   dnet);
        }
        // layer_2: dW2 = dpre2 a1^T, db2; dpre1 = (W2^T dpre2) * lrelu'(a1)  (-> X)
        launch_wgrad_img(Y, 4 * C, 4 * C, s.a1[i], 2 * C, 2 * C, B, P, G.up2_w[i], 2 * C, G.up2_b[i], 0, t.wg, st, &wd);
        g = Conv16Params{};
        g.plan = bp[i].c2; g.At = s.pack + bp[i].o2; g.B = Y; g.b_batch = 4L * C * P; g.C = X; g.c_batch = 2L * C * P;
        g.M = 2 * C; g.K = 4 * C; g.P = (int)P; g.batch = B; g.mask_ref = s.a1[i]; g.mask_batch = 2L * C * P;
        if (launch_conv16(g, st)) return 1;
        // layer_1: dW1 = dpre1 net_in^T, db1; dnet += W1^T dpre1
        launch_wgrad_img(X, 2 * C, 2 * C, net_in, C, C, B, P, G.up1_w[i], C, G.up1_b[i], 0, t.wg, st, &wd);
        g = Conv16Params{};
        g.plan = bp[i].c1; g.At = s.pack + bp[i].o1; g.B = X; g.b_batch = 2L * C * P; g.C = dnet; g.c_batch = (long)C * P;
        g.M = C; g.K = 2 * C; g.P = (int)P; g.batch = B; g.accumulate = 1;
        if (dres_folded) {
            g.accumulate = 0;
            g.dres_from = Y; g.dres_from_batch = 4L * C * P; g.dres_sign = s.sign2[i]; g.dres_sign_batch = (long)C * P;
        }
        if (launch_conv16(g, st)) return 1;
        dnet_next = dnet;             // block i-1's dhid accumulates into it (its Cn x 4P' is this C x P)
    }
    // rgb_0 = up(conv_rgb0(x)): adjoint of up at side S0 -> 2 S0, then the conv
    {
        const int S = d.side[0];
        const long P = (long)S * S;
        hipLaunchKernelGGL(blur_bilinear_adj_kernel, dim3(blocks_for((long)B * 3 * P)), dim3(256), 0, st, drgb, drgb_tmp, (long)B * 3, S, S);
        drgb = drgb_tmp;
        const bool want_w = G.rgb_w[0] || G.rgb_b[0];
        if (want_w || d_x) {
            const unsigned nwg = (unsigned)rgbf_workgroups((long)B * P);
            hipLaunchKernelGGL(rgb_bwd_fused_kernel, dim3(nwg, rgbf_channel_groups(nwg, d.ch[0])), dim3(256), (size_t)12 * (d.ch[0] + 1) * sizeof(float), st, drgb, p->x,
                               d.ch[0], P, B, w->rgb_w[0], d_x ? dnet_next : (float*)nullptr, 1, 0, want_w ? t.colsum + t.colsum_off[0] : (float*)nullptr);
            if (want_w) wsum_push(t.colsum + t.colsum_off[0], d.ch[0], (int)nwg, G.rgb_w[0], G.rgb_b[0]);
        }
    }
    if (wsum.n) hipLaunchKernelGGL(rgb_wsum_batch_kernel, dim3(wsum.first[wsum.n]), dim3(64), 0, st, wsum);
    wgrad_defer_flush(&wd, st);
    if (wd.failed) return fail("gnr_upsample_bwd: a weight-gradient GEMM needs more split-K scratch than the workspace holds (batch %d)", B);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("gnr_upsample_bwd: launch failed: %s", hipGetErrorString(e));
    if (canary_check(st, "gnr_upsample_bwd")) return 1;
    return 0;
}

}  // extern "C"
