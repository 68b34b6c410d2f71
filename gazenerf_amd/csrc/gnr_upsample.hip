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
// All tensors are the reference's channels-first fp32 images [B][C][H*W].  The 1x1 convolutions are GEMMs
// over pixels (C[M][N] = A[M][K] B[K][N], N = pixels contiguous): conv_gemm_kernel, fp32 MFMA
// (v_mfma_f32_32x32x2_f32, exact fp32), 128x128 tile per 256-thread workgroup through a k-major LDS image
// (every operand read is a conflict-free ds_read_b32), epilogues fused: bias + LeakyReLU, the
// residual-repeat + pixel_shuffle store of the PixelShuffleUpsample tail (with the sign byte the backward
// needs), LeakyReLU-derivative masks and accumulation for the dgrad GEMMs (A = W^T by strides).
// Weight gradients re-use wgrad_kernel (gnr_wgrad.hip) on the image layout; the 3-channel RGB branch and the
// stencils (blur, bilinear, their adjoints) are one-thread-per-pixel HBM-bound kernels.
// Work per 64x64 -> 512x512 image: 19.6 GFLOP forward (9.8 GMAC), ~0.6 GB of activation traffic.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/gnr.h"
#include "gnr_device.h"

namespace gnr {
int fail(const char* fmt, ...);
size_t wgrad_scratch_floats();
void launch_wgrad_img(const float* A, int lda, int n_valid, const float* B, int ldb, int k_valid, int batch,
                      long pixels_per_image, float* dW, int ldw, float* colsum_out, int colsum_ld, float* scratch,
                      hipStream_t stream);

constexpr float LEAK = 0.2f;
constexpr int UP_MAX = GNR_UPSAMPLE_MAX_BLOCKS;

// ---------------------------------------------------------------------------------------------
// C[M][N] = epilogue(A[M][K] B[K][N])
// ---------------------------------------------------------------------------------------------
struct GemmParams {
    const float* A; long a_rs, a_cs;               // A(m,k) = A[m*a_rs + k*a_cs]  (source of the packed copy)
    const float* At; int Mp;                       // packed k-major copy: At[k*Mp + m], zero padded to Kp x Mp
    const float* B; long b_batch;                  // B(b,k,n) = B[b*b_batch + k*N + n]
    float* C; long c_batch;                        // C(b,m,n) = C[b*c_batch + m*N + n]   (plain store)
    int M, K, N;                                   // N % 128 == 0
    const float* bias;                             // [M] or NULL
    int leaky;                                     // LeakyReLU(0.2) on (acc + bias)
    const float* mask_ref; long mask_batch;        // result *= (mask_ref(b,m,n) > 0 ? 1 : 0.2)
    int accumulate;                                // C += result
    int shuffle, W;                                // PixelShuffleUpsample tail: n = y*W + x
    const float* res; long res_batch;              // residual res(b, m % (M/4), n)
    unsigned char* sign_out; long sign_batch;      // shuffle: bit e of byte (b, m/4, n) = (acc + bias > 0) of channel 4(m/4)+e
};

constexpr int GK = 16;             // k per LDS tile: two 4-step groups per lane-half
constexpr int GN = 256;            // pixels per workgroup tile: 2 waves x 128

// At[k/4][m][k%4] = A(m,k) for k < K, m < M, else 0; Kp % 16 == 0, Mp % 128 == 0.  The weights are tiny (<= 2 MB):
// re-laying them out per call makes the A tile of a k-group ONE contiguous block for the LDS-DMA, and a lane's
// ds_read_b128 the four k-steps of its row.
__global__ void pack_a_kernel(const float* __restrict__ A, long a_rs, long a_cs, int M, int K, int Mp, int Kp,
                              float* __restrict__ At) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Mp * Kp) return;
    const int k4 = (int)(i / (4L * Mp)), r = (int)(i - (long)k4 * 4 * Mp), m = r >> 2, k = 4 * k4 + (r & 3);
    At[i] = (m < M && k < K) ? A[(long)m * a_rs + (long)k * a_cs] : 0.0f;
}

typedef int gi32x4 __attribute__((ext_vector_type(4)));

// C[M][N] tile of (64 XM) x 256 per 256-thread workgroup: 2 (M) x 2 (N) waves, a wave = XM row tiles x 4 pixel
// "tiles" of v_mfma_f32_32x32x2_f32.  Operand order: the contraction index k is free, so lane-half h takes the
// 4-step groups 2q + h of a 16-k tile; the A value of 4 consecutive steps is one ds_read_b128 of the packed weights;
// a lane's B operand of one step is a ds_read_b128 of FOUR CONSECUTIVE PIXELS of row k -- pixel 4 li + t feeds column
// li of pixel-tile t, so one read feeds four MFMAs and a lane ends up owning 4 consecutive pixels of each of its
// rows: float4 stores in the epilogue.  10-12 ds_read_b128 per 32 XM MFMAs (was: one ds_read_b32 per MFMA).
// Both operand tiles arrive by LDS-DMA (the B rows are 1 KiB contiguous, the packed A groups 16 bytes x rows), rows
// k >= K read zeros through the buffer descriptor's bound: no staging registers, no VALU, no ds_write.
template <int XM>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const GemmParams gp) {
    constexpr int TM = 64 * XM;
    constexpr int A_BYTES = (GK / 4) * TM * 16, B_BYTES = GK * GN * 4, BUF = A_BYTES + B_BYTES;
    constexpr int PA = A_BYTES / 1024, PB = B_BYTES / 1024;            // 1 KiB DMA pieces per k-tile
    __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * GN, m0 = blockIdx.y * TM, b = blockIdx.z;
    const int nk = (gp.K + GK - 1) / GK;

    auto desc = [&](const float* base, long bytes) {
        const unsigned long long a = (unsigned long long)base;
        if (bytes > 0xFFFFFFFFL) bytes = 0xFFFFFFFFL;
        gi32x4 r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
        r.z = __builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
        r.w = 0x00020000;
        return r;
    };
    // A: packed [Kp/4][Mp][4], tile rows m0.. ; B: image b, [K][N] with the bound at row K
    const gi32x4 rsa = desc(gp.At + (long)m0 * 4, ((long)nk * (GK / 4) * gp.Mp - m0) * 16);
    const gi32x4 rsb = desc(gp.B + (long)b * gp.b_batch + n0, ((long)gp.K * gp.N - n0) * 4);
    const unsigned lds0 = (unsigned)(size_t)&lds[0];
    const unsigned voff = (unsigned)lane * 16u;
    auto dma_tile = [&](int kt, int buf) {
        const unsigned la = lds0 + (unsigned)buf * BUF, lb = la + A_BYTES;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
#pragma unroll
        for (int j = wave; j < PA; j += 4) {      // piece j: k-group j / (TM/64), 64-row slice j % (TM/64)
            const unsigned g = (unsigned)j / (TM / 64), sl = (unsigned)j % (TM / 64);
            const unsigned so = ((unsigned)(kt * (GK / 4) + g) * (unsigned)gp.Mp + sl * 64u) * 16u;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                         :: "v"(voff), "s"(rsa), "s"(la + (unsigned)j * 1024u), "s"(so) : "memory");
        }
#pragma unroll
        for (int j = wave; j < PB; j += 4) {      // piece j: row kt*16 + j, 256 pixels
            const unsigned so = (unsigned)(kt * GK + j) * (unsigned)gp.N * 4u;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                         :: "v"(voff), "s"(rsb), "s"(lb + (unsigned)j * 1024u), "s"(so) : "memory");
        }
        asm volatile("s_mov_b32 m0, %0" :: "s"(keep));
    };

    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[XM][4];
#pragma unroll
    for (int x = 0; x < XM; ++x)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[x][t] = zero;
    // byte offsets inside a buffer: A group g, row r -> (g*TM + r)*16;  B row k, pixel p -> A_BYTES + (k*256 + p)*4
    const int a_off = (lh * TM + wm * 32 * XM + li) * 16;                  // + (2q*TM + 32x)*16
    const int b_off = A_BYTES + (4 * lh * GN + wn * 128 + 4 * li) * 4;    // + ((8q + s)*GN)*4

    dma_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) dma_tile(kt + 1, buf ^ 1);
        const char* base = lds + buf * BUF;
        f32x4 av[XM][2], bv[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int x = 0; x < XM; ++x) av[x][q] = *(const f32x4*)(base + a_off + (2 * q * TM + 32 * x) * 16);
#pragma unroll
            for (int sst = 0; sst < 4; ++sst) bv[q][sst] = *(const f32x4*)(base + b_off + (8 * q + sst) * GN * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int sst = 0; sst < 4; ++sst)
#pragma unroll
                for (int x = 0; x < XM; ++x)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[x][t] = mfma32(av[x][q][sst], bv[q][sst][t], acc[x][t]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const int Cq = gp.M / 4;
    const int n = n0 + 128 * wn + 4 * li;                                   // this lane's 4 consecutive pixels
    if (gp.shuffle) {
        // PixelShuffleUpsample tail.  A lane's registers 4q..4q+3 are in-channels 4c..4c+3 of its 4 pixels, i.e. the
        // 2x2 output blocks of out-channel c at 4 consecutive x: two rows of 8 consecutive floats (float4 stores); the
        // four pre-activation signs per pixel go into one nibble, four pixels into one 32-bit store.
        const int py = n / gp.W, px = n - py * gp.W;
#pragma unroll
        for (int x = 0; x < XM; ++x)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mb = m0 + 32 * XM * wm + 32 * x + 8 * q + 4 * lh;      // multiple of 4
                if (mb >= gp.M) continue;
                float v[4][4];                                                   // [e: channel][t: pixel]
                unsigned nib = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = mb + e;
                    const float bias = gp.bias[m];
                    // x.repeat(1,4,1,1): in-channel m reads x channel m % (M/4)
                    const f32x4 res = *(const f32x4*)(gp.res + (long)b * gp.res_batch + (long)(m % Cq) * gp.N + n);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float u = acc[x][t][4 * q + e] + bias;
                        nib |= (u > 0.0f ? 1u : 0u) << (8 * t + e);
                        u = u > 0.0f ? u : LEAK * u;
                        v[e][t] = u + res[t];
                    }
                }
                *(unsigned*)(gp.sign_out + (long)b * gp.sign_batch + (long)(mb >> 2) * gp.N + n) = nib;
                // pixel_shuffle(2): in-channel 4c + 2i + j -> out (c, 2y+i, 2x+j)
                float* dst = gp.C + (long)b * gp.c_batch + (long)(mb >> 2) * (4L * gp.N) + (long)(2 * py) * (2 * gp.W) + 2 * px;
                *(f32x4*)dst = f32x4{v[0][0], v[1][0], v[0][1], v[1][1]};
                *(f32x4*)(dst + 4) = f32x4{v[0][2], v[1][2], v[0][3], v[1][3]};
                *(f32x4*)(dst + 2 * gp.W) = f32x4{v[2][0], v[3][0], v[2][1], v[3][1]};
                *(f32x4*)(dst + 2 * gp.W + 4) = f32x4{v[2][2], v[3][2], v[2][3], v[3][3]};
            }
        return;
    }
#pragma unroll
    for (int x = 0; x < XM; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * XM * wm + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= gp.M) continue;
            f32x4 v = {acc[x][0][r], acc[x][1][r], acc[x][2][r], acc[x][3][r]};
            if (gp.bias) v += gp.bias[m];
            if (gp.leaky) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = v[t] > 0.0f ? v[t] : LEAK * v[t];
            }
            if (gp.mask_ref) {
                const f32x4 mk = *(const f32x4*)(gp.mask_ref + (long)b * gp.mask_batch + (long)m * gp.N + n);
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] *= mk[t] > 0.0f ? 1.0f : LEAK;
            }
            float* dst = gp.C + (long)b * gp.c_batch + (long)m * gp.N + n;
            if (gp.accumulate) v += *(const f32x4*)dst;
            *(f32x4*)dst = v;
        }
}

static size_t pack_floats(int M, int K) {      // padded size of one packed operand, either orientation
    auto one = [](int m, int k) { return (size_t)((m + 127) / 128 * 128) * ((k + GK - 1) / GK * GK); };
    const size_t a = one(M, K), b = one(K, M);
    return a > b ? a : b;
}

static void launch_gemm(GemmParams gp, int batch, float* pack, hipStream_t st) {
    gp.Mp = (gp.M + 127) / 128 * 128;
    const int Kp = (gp.K + GK - 1) / GK * GK;
    hipLaunchKernelGGL(pack_a_kernel, dim3((unsigned)(((long)gp.Mp * Kp + 255) / 256)), dim3(256), 0, st, gp.A, gp.a_rs, gp.a_cs,
                       gp.M, gp.K, gp.Mp, Kp, pack);
    gp.At = pack;
    // 64-row tiles where they waste fewer padded rows (M = 129, 258, 516: one channel past a multiple of 128); their
    // MFMA-per-read ratio is lower, hence the 1.1
    const long w128 = (long)((gp.M + 127) / 128) * 128, w64 = (long)((gp.M + 63) / 64) * 64;
    if (w64 * 11 < w128 * 10)
        hipLaunchKernelGGL((conv_gemm_kernel<1>), dim3(gp.N / GN, (gp.M + 63) / 64, batch), dim3(256), 0, st, gp);
    else
        hipLaunchKernelGGL((conv_gemm_kernel<2>), dim3(gp.N / GN, (gp.M + 127) / 128, batch), dim3(256), 0, st, gp);
}

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
    auto row = [&](int yy) {
        const float* r = p + (long)yy * W;
        const f32x4 c = *(const f32x4*)(r + x);
        const float l = r[xm], rr = r[xp];
        return f32x4{wl[0] * l + wc[0] * c.x + wr[0] * c.y, wl[1] * c.x + wc[1] * c.y + wr[1] * c.z,
                     wl[2] * c.y + wc[2] * c.z + wr[2] * c.w, wl[3] * c.z + wc[3] * c.w + wr[3] * rr};
    };
    const f32x4 a = row(y0), b = row(y), c = row(y2);
    *(f32x4*)(out + (q / ((long)(W / 4) * H)) * ((long)H * W) + (long)y * W + x) = yl * a + yc * b + yr * c;
}

// bilinear x2, align_corners=False: out[2m] = .25 in[m-1] + .75 in[m] (m = 0: in[0]); out[2m+1] = .75 in[m] + .25 in[m+1]
__global__ __launch_bounds__(256) void bilinear2x_kernel(const float* __restrict__ in, float* __restrict__ out, long planes,
                                                         int H, int W) {
    const int H2 = 2 * H, W2 = 2 * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * H2 * W2) return;
    const int ox = (int)(idx % W2), oy = (int)((idx / W2) % H2);
    const float* p = in + (idx / ((long)H2 * W2)) * ((long)H * W);
    auto taps = [](int o, int n, int& i0, int& i1, float& w0, float& w1) {
        const int m = o >> 1;
        if (o & 1) { i0 = m; i1 = m + 1 < n ? m + 1 : m; w0 = 0.75f; w1 = 0.25f; }
        else { i0 = m >= 1 ? m - 1 : 0; i1 = m; w0 = m >= 1 ? 0.25f : 0.0f; w1 = m >= 1 ? 0.75f : 1.0f; }
    };
    int x0, x1, y0, y1;
    float wx0, wx1, wy0, wy1;
    taps(ox, W, x0, x1, wx0, wx1);
    taps(oy, H, y0, y1, wy0, wy1);
    out[idx] = wy0 * (wx0 * p[(long)y0 * W + x0] + wx1 * p[(long)y0 * W + x1]) +
               wy1 * (wx0 * p[(long)y1 * W + x0] + wx1 * p[(long)y1 * W + x1]);
}

// adjoint: din[m] = sum_o w(o,m) dout[o] with o in {2m-1, 2m, 2m+1, 2m+2}
__global__ __launch_bounds__(256) void bilinear2x_adj_kernel(const float* __restrict__ dout, float* __restrict__ din,
                                                             long planes, int H, int W) {
    const int H2 = 2 * H, W2 = 2 * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * H * W) return;
    const int x = (int)(idx % W), y = (int)((idx / W) % H);
    const float* p = dout + (idx / ((long)H * W)) * ((long)H2 * W2);
    auto taps = [](int m, int n, int (&o)[4], float (&w)[4]) {
        o[0] = 2 * m - 1; w[0] = m >= 1 ? 0.25f : 0.0f;                 // odd output of m-1 reads in[m]
        o[1] = 2 * m;     w[1] = m >= 1 ? 0.75f : 1.0f;
        o[2] = 2 * m + 1; w[2] = m + 1 < n ? 0.75f : 1.0f;
        o[3] = 2 * m + 2; w[3] = m + 1 < n ? 0.25f : 0.0f;              // even output of m+1 reads in[m]
        if (o[0] < 0) o[0] = 0;
        if (o[3] > 2 * n - 1) o[3] = 2 * n - 1;
    };
    int ox[4], oy[4];
    float wx[4], wy[4];
    taps(x, W, ox, wx);
    taps(y, H, oy, wy);
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float r = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) r += wx[c] * p[(long)oy[a] * W2 + ox[c]];
        acc += wy[a] * r;
    }
    din[idx] = acc;
}

// ---------------------------------------------------------------------------------------------
// the 3-channel RGB branch
// ---------------------------------------------------------------------------------------------
// rgb(b,o,p) = [rgb(b,o,p) +] sum_c W[o][c] net(b,c,p) + bias[o];  img = sigmoid(rgb) if wanted
__global__ __launch_bounds__(256) void rgb_conv_kernel(const float* __restrict__ net, int C, long P, int batch,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ rgb, int accumulate, float* __restrict__ img) {
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
        ip[0] = 1.0f / (1.0f + expf(-a0)); ip[P] = 1.0f / (1.0f + expf(-a1)); ip[2 * P] = 1.0f / (1.0f + expf(-a2));
    }
}

// dnet(b,c,p) = ([dnet(b,c,p)] + sum_o W[o][c] drgb(b,o,p)) * (net(b,c,p) > 0 ? 1 : 0.2)   [mask optional]
__global__ __launch_bounds__(256) void rgb_conv_bwd_data_kernel(const float* __restrict__ drgb, int C, long P, int batch,
                                                                const float* __restrict__ w, float* __restrict__ dnet,
                                                                int accumulate, const float* __restrict__ act) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * P) return;
    const long b = idx / P, p = idx - b * P;
    const float* gp = drgb + b * 3 * P + p;
    const float g0 = gp[0], g1 = gp[P], g2 = gp[2 * P];
    // 8 channels per round: their dnet / act loads are issued before the first store (every channel is independent)
    for (int c0 = 0; c0 < C; c0 += 8) {
        float dv[8], av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long o = b * C * P + (long)(c0 + u) * P + p;
            dv[u] = (accumulate && c0 + u < C) ? dnet[o] : 0.0f;
            av[u] = (act && c0 + u < C) ? act[o] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            if (c < C) {
                float v = w[c] * g0 + w[C + c] * g1 + w[2 * C + c] * g2;
                if (accumulate) v += dv[u];
                if (act) v *= av[u] > 0.0f ? 1.0f : LEAK;
                dnet[b * C * P + (long)c * P + p] = v;
            }
        }
    }
}

// dW[o][c] = sum_{b,p} drgb(b,o,p) net(b,c,p);  db[o] = sum drgb(b,o,p)  (channel c == C stands for the bias).
// Grid (C+1, RGBW_SPLITS): each block reduces a contiguous slice of the (b,p) range; rgb_wsum_kernel adds the
// slices in a fixed order (deterministic).
constexpr int RGBW_SPLITS = 32;

__global__ __launch_bounds__(256) void rgb_conv_bwd_weight_kernel(const float* __restrict__ drgb, const float* __restrict__ net,
                                                                  int C, long P, int batch, float* __restrict__ part) {
    __shared__ float red[3][256];
    const int c = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x;
    const long total = (long)batch * P, per = (total + RGBW_SPLITS - 1) / RGBW_SPLITS;
    const long i0 = (long)sp * per, i1 = i0 + per < total ? i0 + per : total;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (long i = i0 + tid; i < i1; i += 256) {
        const long b = i / P, p = i - b * P;
        const float v = c < C ? net[b * C * P + (long)c * P + p] : 1.0f;
        const float* gp = drgb + b * 3 * P + p;
        a0 = fmaf(gp[0], v, a0); a1 = fmaf(gp[P], v, a1); a2 = fmaf(gp[2 * P], v, a2);
    }
    red[0][tid] = a0; red[1][tid] = a1; red[2][tid] = a2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; red[2][tid] += red[2][tid + s]; }
        __syncthreads();
    }
    if (tid < 3) part[((long)c * RGBW_SPLITS + sp) * 3 + tid] = red[tid][0];
}

__global__ void rgb_wsum_kernel(const float* __restrict__ part, int C, float* __restrict__ dw, float* __restrict__ db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (c, o)
    if (i >= (C + 1) * 3) return;
    const int c = i / 3, o = i - 3 * c;
    float a = 0.0f;
    for (int sp = 0; sp < RGBW_SPLITS; ++sp) a += part[((long)c * RGBW_SPLITS + sp) * 3 + o];
    if (c < C) { if (dw) dw[o * C + c] = a; }
    else if (db) db[o] = a;
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

// out[n] = sum_b in[b][n]
__global__ void sum_batch_kernel(const float* __restrict__ in, int batch, int n, int ld, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.0f;
    for (int b = 0; b < batch; ++b) a += in[(long)b * ld + i];
    out[i] = a;
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
    return 0;
}

struct UpSaved {                        // kept for the backward
    float* a1[UP_MAX];                  // [B][2C][P]
    unsigned char* sign2[UP_MAX];       // [B][C][P]    four pre-activation sign bits per (out-channel quad, pixel)
    float* pack;                        // packed A operand of the GEMM in flight
    float* v[UP_MAX];                   // [B][C][4P]   blurred
    float* net[UP_MAX];                 // [B][C'][4P]  block output
    float* img;                         // [B][3][Pn]
    float* u;                           // [B][C][4P]   scratch (largest block)
    float* rgb_a; float* rgb_b;         // [B][3][Pn]   scratch
};

static size_t up_carve(const GnrUpsampleProblem* p, const UpDims& d, char* base, UpSaved* s) {
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return q; };
    UpSaved z{};
    size_t umax = 0;
    const size_t B = (size_t)p->batch;
    for (int i = 0; i < d.n_blocks; ++i) {
        const size_t C = d.ch[i], Cn = d.ch[i + 1], P = (size_t)d.side[i] * d.side[i];
        z.a1[i] = (float*)take(B * 2 * C * P * 4);
        z.sign2[i] = (unsigned char*)take(B * C * P);
        z.v[i] = (float*)take(B * C * 4 * P * 4);
        z.net[i] = (float*)take(B * Cn * 4 * P * 4);
        if (B * C * 4 * P * 4 > umax) umax = B * C * 4 * P * 4;
    }
    const size_t Pn = (size_t)d.side[d.n_blocks] * d.side[d.n_blocks];
    z.img = (float*)take(B * 3 * Pn * 4);
    size_t pk = 0;
    for (int i = 0; i < d.n_blocks; ++i) {
        const int C = d.ch[i], Cn = d.ch[i + 1];
        const size_t c3[3] = {pack_floats(2 * C, C), pack_floats(4 * C, 2 * C), pack_floats(Cn, C)};
        for (size_t v : c3) if (v > pk) pk = v;
    }
    z.pack = (float*)take(pk * 4);
    z.u = (float*)take(umax);
    z.rgb_a = (float*)take(B * 3 * Pn * 4);
    z.rgb_b = (float*)take(B * 3 * Pn * 4);
    if (s) *s = z;
    return off;
}

struct UpScratch {                      // backward temporaries
    float* g0; float* g1;               // two buffers of the largest activation size
    float* drgb_a; float* drgb_b;       // [B][3][Pn]
    float* colsum;                      // [B][max M]
    float* wg;                          // wgrad partial tiles
};

static size_t up_carve_bwd(const GnrUpsampleProblem* p, const UpDims& d, char* base, UpScratch* s) {
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return q; };
    size_t big = 0;
    const size_t B = (size_t)p->batch;
    int mmax = 0;
    for (int i = 0; i < d.n_blocks; ++i) {
        const size_t C = d.ch[i], P = (size_t)d.side[i] * d.side[i];
        if (B * 4 * C * P * 4 > big) big = B * 4 * C * P * 4;       // dpre2 == du size
        if (4 * d.ch[i] > mmax) mmax = 4 * d.ch[i];
    }
    const size_t Pn = (size_t)d.side[d.n_blocks] * d.side[d.n_blocks];
    UpScratch z{};
    z.g0 = (float*)take(big);
    z.g1 = (float*)take(big);
    z.drgb_a = (float*)take(B * 3 * Pn * 4);
    z.drgb_b = (float*)take(B * 3 * Pn * 4);
    size_t cs_floats = B * (size_t)(mmax + 128);
    if (cs_floats < (size_t)(mmax + 1) * RGBW_SPLITS * 3) cs_floats = (size_t)(mmax + 1) * RGBW_SPLITS * 3;
    z.colsum = (float*)take(cs_floats * 4);
    z.wg = (float*)take(wgrad_scratch_floats() * 4);
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

// rgb <- blur(bilinear2x(rgb_in)) for 3-channel images at side S -> 2S; tmp holds the bilinear result
static void up_rgb(const float* in, float* tmp, float* out, int batch, int S, hipStream_t st) {
    const long planes = (long)batch * 3;
    hipLaunchKernelGGL(bilinear2x_kernel, dim3(blocks_for(planes * 4L * S * S)), dim3(256), 0, st, in, tmp, planes, S, S);
    hipLaunchKernelGGL(blur_kernel, dim3(blocks_for(planes * 4L * S * S / 4)), dim3(256), 0, st, tmp, out, planes, 2 * S, 2 * S, 0);
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
    up_carve(p, d, (char*)workspace, &s);
    hipStream_t st = (hipStream_t)stream;
    const int B = p->batch;

    // rgb = up(conv_rgb0(x))
    {
        const long P = (long)d.side[0] * d.side[0];
        hipLaunchKernelGGL(rgb_conv_kernel, dim3(blocks_for((long)B * P)), dim3(256), 0, st, p->x, d.ch[0], P, B, w->rgb_w[0],
                           w->rgb_b[0], s.rgb_a, 0, (float*)nullptr);
        up_rgb(s.rgb_a, s.rgb_b, s.rgb_a, B, d.side[0], st);          // in -> tmp -> out: in may be overwritten
    }
    const float* net = p->x;
    float* rgb = s.rgb_a;
    float* rgb_tmp = s.rgb_b;
    for (int i = 0; i < d.n_blocks; ++i) {
        const int C = d.ch[i], Cn = d.ch[i + 1], S = d.side[i];
        const long P = (long)S * S;
        GemmParams g{};
        // a1 = lrelu(W1 net + b1)
        g.A = w->up1_w[i]; g.a_rs = C; g.a_cs = 1; g.B = net; g.b_batch = (long)C * P; g.C = s.a1[i]; g.c_batch = 2L * C * P;
        g.M = 2 * C; g.K = C; g.N = (int)P; g.bias = w->up1_b[i]; g.leaky = 1;
        launch_gemm(g, B, s.pack, st);
        // u = pixel_shuffle(lrelu(W2 a1 + b2) + repeat(net))
        g = GemmParams{};
        g.A = w->up2_w[i]; g.a_rs = 2 * C; g.a_cs = 1; g.B = s.a1[i]; g.b_batch = 2L * C * P; g.C = s.u; g.c_batch = 4L * C * P;
        g.M = 4 * C; g.K = 2 * C; g.N = (int)P; g.bias = w->up2_b[i]; g.leaky = 1; g.shuffle = 1; g.W = S;
        g.res = net; g.res_batch = (long)C * P; g.sign_out = s.sign2[i]; g.sign_batch = (long)C * P;
        launch_gemm(g, B, s.pack, st);
        // v = blur(u)
        hipLaunchKernelGGL(blur_kernel, dim3(blocks_for((long)B * C * P)), dim3(256), 0, st, s.u, s.v[i], (long)B * C, 2 * S,
                           2 * S, 0);
        // net' = lrelu(Wf v + bf)
        g = GemmParams{};
        g.A = w->feat_w[i]; g.a_rs = C; g.a_cs = 1; g.B = s.v[i]; g.b_batch = 4L * C * P; g.C = s.net[i]; g.c_batch = 4L * Cn * P;
        g.M = Cn; g.K = C; g.N = (int)(4 * P); g.bias = w->feat_b[i]; g.leaky = 1;
        launch_gemm(g, B, s.pack, st);
        // rgb += conv_rgb(i+1)(net');  last block: img = sigmoid(rgb) (or rgb itself)
        const bool last = i == d.n_blocks - 1;
        hipLaunchKernelGGL(rgb_conv_kernel, dim3(blocks_for((long)B * 4 * P)), dim3(256), 0, st, s.net[i], Cn, 4 * P, B,
                           w->rgb_w[i + 1], w->rgb_b[i + 1], rgb, 1, last && p->final_sigmoid ? s.img : (float*)nullptr);
        if (!last) up_rgb(rgb, rgb_tmp, rgb, B, 2 * S, st);
        net = s.net[i];
    }
    const size_t out_bytes = (size_t)B * 3 * d.side[d.n_blocks] * d.side[d.n_blocks] * 4;
    (void)hipMemcpyAsync(img, p->final_sigmoid ? s.img : rgb, out_bytes, hipMemcpyDeviceToDevice, st);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("gnr_upsample_fwd: launch failed: %s", hipGetErrorString(e));
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
    up_carve(p, d, (char*)saved, &s);
    UpScratch t;
    up_carve_bwd(p, d, (char*)scratch, &t);
    GnrUpsampleWeightGrads G{};
    if (dw) G = *dw;
    hipStream_t st = (hipStream_t)stream;
    const int B = p->batch, nb = d.n_blocks;
    const long Pn = (long)d.side[nb] * d.side[nb];

    // d(rgb) at full resolution
    float* drgb = t.drgb_a;
    float* drgb_tmp = t.drgb_b;
    if (p->final_sigmoid)
        hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(blocks_for((long)B * 3 * Pn)), dim3(256), 0, st, d_img, s.img, drgb, (long)B * 3 * Pn);
    else
        (void)hipMemcpyAsync(drgb, d_img, (size_t)B * 3 * Pn * 4, hipMemcpyDeviceToDevice, st);

    float* dnet_next = nullptr;       // gradient w.r.t. net' of block i coming from block i+1 (its input)
    for (int i = nb - 1; i >= 0; --i) {
        const int C = d.ch[i], Cn = d.ch[i + 1], S = d.side[i];
        const long P = (long)S * S, P4 = 4 * P;
        const float* net_in = i == 0 ? p->x : s.net[i - 1];
        // the RGB branch at this resolution: rgb_i = up(rgb_{i-1}) + conv(net') ...; undo the up() that FOLLOWED block i
        if (i < nb - 1) {
            // drgb currently is at side 4S (block i+1's resolution): adjoint of blur o bilinear
            hipLaunchKernelGGL(blur_kernel, dim3(blocks_for((long)B * 3 * 4 * P)), dim3(256), 0, st, drgb, drgb_tmp, (long)B * 3, 4 * S,
                               4 * S, 1);
            hipLaunchKernelGGL(bilinear2x_adj_kernel, dim3(blocks_for((long)B * 3 * P4)), dim3(256), 0, st, drgb_tmp, drgb, (long)B * 3,
                               2 * S, 2 * S);
        }
        // conv_rgb(i+1): weight/bias gradients, then dhid = (dnet' + Wr^T drgb) * lrelu'(net')
        if (G.rgb_w[i + 1] || G.rgb_b[i + 1]) {
            hipLaunchKernelGGL(rgb_conv_bwd_weight_kernel, dim3(Cn + 1, RGBW_SPLITS), dim3(256), 0, st, drgb, s.net[i], Cn, P4, B, t.colsum);
            hipLaunchKernelGGL(rgb_wsum_kernel, dim3((3 * (Cn + 1) + 63) / 64), dim3(64), 0, st, t.colsum, Cn, G.rgb_w[i + 1], G.rgb_b[i + 1]);
        }
        float* dhid = dnet_next ? dnet_next : t.g0;
        hipLaunchKernelGGL(rgb_conv_bwd_data_kernel, dim3(blocks_for((long)B * P4)), dim3(256), 0, st, drgb, Cn, P4, B, w->rgb_w[i + 1],
                           dhid, dnet_next ? 1 : 0, s.net[i]);
        float* other = dhid == t.g0 ? t.g1 : t.g0;
        // feat_layers[i]: dWf = dhid v^T, dbf; dv = Wf^T dhid
        launch_wgrad_img(dhid, Cn, Cn, s.v[i], C, C, B, P4, G.feat_w[i], C, t.colsum, Cn + 128, t.wg, st);
        if (G.feat_b[i]) hipLaunchKernelGGL(sum_batch_kernel, dim3((Cn + 63) / 64), dim3(64), 0, st, t.colsum, B, Cn, Cn + 128, G.feat_b[i]);
        GemmParams g{};
        g.A = w->feat_w[i]; g.a_rs = 1; g.a_cs = C; g.B = dhid; g.b_batch = (long)Cn * P4; g.C = other; g.c_batch = (long)C * P4;
        g.M = C; g.K = Cn; g.N = (int)P4;
        launch_gemm(g, B, s.pack, st);                                           // other = dv
        // du = blur^T dv  (into dhid's buffer)
        hipLaunchKernelGGL(blur_kernel, dim3(blocks_for((long)B * C * P)), dim3(256), 0, st, other, dhid, (long)B * C, 2 * S, 2 * S, 1);
        float* du = dhid;
        // un-shuffle: dpre2 (-> other) and the residual part of d(net_in) (-> s.u, free in the backward)
        float* dpre2 = other;
        float* dnet = s.u;
        hipLaunchKernelGGL(unshuffle_bwd_kernel, dim3(blocks_for((long)B * 2 * C * P)), dim3(256), 0, st, du, s.sign2[i], C, S, S, B, dpre2,
                           dnet);
        // layer_2: dW2 = dpre2 a1^T, db2; dpre1 = (W2^T dpre2) * lrelu'(a1)  (-> du's buffer)
        launch_wgrad_img(dpre2, 4 * C, 4 * C, s.a1[i], 2 * C, 2 * C, B, P, G.up2_w[i], 2 * C, t.colsum, 4 * C + 128, t.wg, st);
        if (G.up2_b[i]) hipLaunchKernelGGL(sum_batch_kernel, dim3((4 * C + 63) / 64), dim3(64), 0, st, t.colsum, B, 4 * C, 4 * C + 128, G.up2_b[i]);
        float* dpre1 = du;
        g = GemmParams{};
        g.A = w->up2_w[i]; g.a_rs = 1; g.a_cs = 2 * C; g.B = dpre2; g.b_batch = 4L * C * P; g.C = dpre1; g.c_batch = 2L * C * P;
        g.M = 2 * C; g.K = 4 * C; g.N = (int)P; g.mask_ref = s.a1[i]; g.mask_batch = 2L * C * P;
        launch_gemm(g, B, s.pack, st);
        // layer_1: dW1 = dpre1 net_in^T, db1; dnet += W1^T dpre1
        launch_wgrad_img(dpre1, 2 * C, 2 * C, net_in, C, C, B, P, G.up1_w[i], C, t.colsum, 2 * C + 128, t.wg, st);
        if (G.up1_b[i]) hipLaunchKernelGGL(sum_batch_kernel, dim3((2 * C + 63) / 64), dim3(64), 0, st, t.colsum, B, 2 * C, 2 * C + 128, G.up1_b[i]);
        g = GemmParams{};
        g.A = w->up1_w[i]; g.a_rs = 1; g.a_cs = C; g.B = dpre1; g.b_batch = 2L * C * P; g.C = dnet; g.c_batch = (long)C * P;
        g.M = C; g.K = 2 * C; g.N = (int)P; g.accumulate = 1;
        launch_gemm(g, B, s.pack, st);
        // hand d(net_in) to block i-1 in a buffer that survives: g0/g1 are free again -> copy into the one not used next
        if (i > 0) {
            (void)hipMemcpyAsync(t.g0, dnet, (size_t)B * C * P * 4, hipMemcpyDeviceToDevice, st);
            dnet_next = t.g0;
        } else {
            // block 0: + conv_rgb0 path below, then out
            dnet_next = dnet;
        }
    }
    // rgb_0 = up(conv_rgb0(x)): adjoint of up at side S0 -> 2 S0, then the conv
    {
        const int S = d.side[0];
        const long P = (long)S * S;
        hipLaunchKernelGGL(blur_kernel, dim3(blocks_for((long)B * 3 * P)), dim3(256), 0, st, drgb, drgb_tmp, (long)B * 3, 2 * S, 2 * S, 1);
        hipLaunchKernelGGL(bilinear2x_adj_kernel, dim3(blocks_for((long)B * 3 * P)), dim3(256), 0, st, drgb_tmp, drgb, (long)B * 3, S, S);
        if (G.rgb_w[0] || G.rgb_b[0]) {
            hipLaunchKernelGGL(rgb_conv_bwd_weight_kernel, dim3(d.ch[0] + 1, RGBW_SPLITS), dim3(256), 0, st, drgb, p->x, d.ch[0], P, B, t.colsum);
            hipLaunchKernelGGL(rgb_wsum_kernel, dim3((3 * (d.ch[0] + 1) + 63) / 64), dim3(64), 0, st, t.colsum, d.ch[0], G.rgb_w[0], G.rgb_b[0]);
        }
        hipLaunchKernelGGL(rgb_conv_bwd_data_kernel, dim3(blocks_for((long)B * P)), dim3(256), 0, st, drgb, d.ch[0], P, B, w->rgb_w[0],
                           dnet_next, 1, (const float*)nullptr);
        if (d_x) (void)hipMemcpyAsync(d_x, dnet_next, (size_t)B * d.ch[0] * P * 4, hipMemcpyDeviceToDevice, st);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("gnr_upsample_bwd: launch failed: %s", hipGetErrorString(e));
    return 0;
}

}  // extern "C"
