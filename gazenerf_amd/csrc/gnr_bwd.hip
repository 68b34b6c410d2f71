// gnr_bwd.hip -- backward of the hot path (gfx950).
//
// Per stream (face, then eyes; the dY scratch is re-used):
//   1. gt_kernel          upstream d(feat_out) [B,C,N_r] -> row-major [ray][288]
//   2. comp_bwd_kernel    CalcRayColor backward (utils/model_utils.py:498-534) per ray:
//                         s_i = g . feat_i, w_i, dL/dsigma_raw_i, sum_i dL/ddelta_i * delta_i
//   3. packT16_kernel     W^T as MFMA A-fragments in the chain's k-order            (gnr_bwd16.hip; bf16x3: gnr_bwd3.hip)
//   4. bwd16_chain_kernel register-chained dgrad through RGB2..L0 (same structure and FLOPs as the
//                         forward kernel), ReLU masks from the saved sign bits, every layer's dY
//                         dumped chunk-channel-major for the weight gradients, d(encoding) -> d(pts) partials
//   5. launch_wgrad x12   dW = dY^T X  (gnr_wgrad.hip), colsum for the biases
//   6. latent_kernel      per-image bias sums -> d(shape,gaze,appea) and the latent columns of dW
// Once per call: geo_kernel  d(pts), dL/dl partials -> dR, dT (GenSamplePoints backward).
#include <atomic>

#include "gnr_bwd_common.h"
#include "gnr_canary.h"
#include "gnr_wgrad.h"

namespace gnr {
int fail(const char* fmt, ...);
size_t carve_fwd(const GnrProblem* p, int n_streams, bool save, char* base, FwdParams* fp);
size_t wgrad_arena_floats(int batch, int max_m, int max_k);
void launch_wgrad(const float* A, int lda, int n_valid, const float* B, int ldb, int k_valid, int batch,
                  long chunks_per_image, float* dW, int ldw, int col_off, int enc_map, float* colsum_out,
                  int colsum_ld, const float* vec, float* vec_out, float* scratch, hipStream_t stream, bool bf16x3,
                  int n_crop = -1, int k_crop = -1, bool small_tiles = false, WgradDefer* defer = nullptr);
void launch_vecsum(const float* v, int batch, long per_image, float* out, int out_stride, hipStream_t stream);
void stage_mark(int stage, int which, hipStream_t st);
struct VdBwdScratch { float* d_rb[2]; float* dR_part; float* dW_part; float* dR_extra; int bpi, segs; };
size_t vd_bwd_floats(const GnrProblem* p, int n_streams);
void vd_carve_bwd(const GnrProblem* p, int n_streams, float* base, VdBwdScratch* sc);
void launch_vd_bwd(const GnrProblem& p, int n_streams, const GnrWeights* const* w, const GnrWeightGrads* const* dw,
                   const float* embed, const VdBwdScratch& sc, bool want_dR, hipStream_t st);
void launch_packT16(const PackTParams& pt, hipStream_t stream);
void launch_bwd16_chain(const BwdParams& bp, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// upstream gradient transpose: g[b][n][ray] -> gT[ray_g][288] (zero padded / zero when NULL)
// ---------------------------------------------------------------------------------------------
__global__ void gt_kernel(const float* __restrict__ g, int feat_nc, int n_rays, long n_rays_total,
                          float* __restrict__ gT) {
    __shared__ float tile[32][33];
    const long ray0 = (long)blockIdx.x * 32;
    const int n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 8 rows at a time
    for (int yy = ty; yy < 32; yy += 8) {
        const int n = n0 + yy;
        const long rg = ray0 + tx;
        float v = 0.0f;
        if (g && n < feat_nc && rg < n_rays_total) {
            const long b = rg / n_rays, r = rg - b * n_rays;
            v = g[(b * feat_nc + n) * n_rays + r];
        }
        tile[yy][tx] = v;
    }
    __syncthreads();
    for (int yy = ty; yy < 32; yy += 8) {
        const long rg = ray0 + yy;
        if (rg < n_rays_total) gT[rg * FEAT_PAD + n0 + tx] = tile[tx][yy];
    }
}

// ---------------------------------------------------------------------------------------------
// CalcRayColor backward.  One wavefront per ray.
//   out_n = sum_i w_i f_i[n], bg = 1 - sum_i w_i, w_i = a_i T_i, T_i = prod_{j<i} x_j, x = 1-a+1e-10
//   q_i  = g.f_i - g_bg ;  dL/da_i = q_i T_i - (sum_{j>i} q_j w_j) / x_i
//   a = 1 - exp(-sigma delta): dL/dsigma = dL/da delta e, dL/ddelta = dL/da sigma e,  e = exp(-sigma delta)
// ---------------------------------------------------------------------------------------------
struct CompBwdParams {
    GnrProblem prob;
    int chunks_per_ray;
    const float* gT;          // [rays][288]
    const float* g_bg;        // [B,1,N_r] or NULL
    const float* act_feat;    // [M][288]
    const float* sigma_raw;   // [M]
    const float* delta;       // [M]
    float* wglob;             // [M]  w_i
    float* dsig;              // [M]  dL/dsigma_raw_i
    float* csum;              // [rays] sum_i dL/ddelta_i * delta_i
    float* dsig_ray;          // [rays] sum_i dL/dsigma_raw_i  (density bias gradient)
    int accumulate;
};

constexpr int CB_MAX = 512;

__global__ __launch_bounds__(256) void comp_bwd_kernel(const CompBwdParams cp) {
    __shared__ float sh_q[4][CB_MAX], sh_x[4][CB_MAX], sh_T[4][CB_MAX], sh_S[4][CB_MAX], sh_g[4][FEAT_PAD];
    const GnrProblem& p = cp.prob;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long n_rays_total = (long)p.batch * p.n_rays;
    const long ray = (long)blockIdx.x * 4 + wave;
    const bool live = ray < n_rays_total;
    const int np = p.n_samples, cpr = cp.chunks_per_ray;
    float* q = sh_q[wave]; float* xx = sh_x[wave]; float* TT = sh_T[wave]; float* SS = sh_S[wave];
    const long row0 = live ? ray * cpr * CHUNK : 0;       // padded rows of this ray are contiguous

    // q_i = g . feat_i - g_bg: the features are CCM ([chunk][288][32]), so a lane owns one sample and
    // walks the channels; each load instruction covers two full 128-byte rows (two chunks).
    float* gsh = sh_g[wave];
    if (live)
        for (int c = lane; c < FEAT_PAD; c += 64) gsh[c] = cp.gT[ray * FEAT_PAD + c];
    const float gbg = (live && cp.g_bg) ? cp.g_bg[ray] : 0.0f;
    __syncthreads();
    if (live) {
        for (int i = lane; i < cpr * CHUNK; i += 64) {
            const float* fr = cp.act_feat + (row0 / CHUNK + i / CHUNK) * (long)(CHUNK * FEAT_PAD) + (i % CHUNK);
            // 16 loads in flight per lane (round 3; 4 before: 16 waves x 1 KiB per CU did not cover the HBM latency --
            // 0.68 of the 8 TB/s spec); four accumulator chains as before, so the summation order is unchanged
            float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
            static_assert(FEAT_PAD % 16 == 0, "unroll");
            // round 5: only the 16-channel groups that hold real channels (258 -> 272 of the layout's 288; the upstream
            // gradient of a padded channel is zero, so the skipped terms were fmaf(f, 0, v) = v: same bits, 5.6 % fewer bytes)
            const int nf = (p.feat_nc + 15) & ~15;
            for (int n = 0; n < nf; n += 16) {
                float f[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) f[u] = __builtin_nontemporal_load(fr + (n + u) * CHUNK);
#pragma unroll
                for (int u = 0; u < 16; u += 4) {
                    v0 = fmaf(f[u + 0], gsh[n + u + 0], v0);
                    v1 = fmaf(f[u + 1], gsh[n + u + 1], v1);
                    v2 = fmaf(f[u + 2], gsh[n + u + 2], v2);
                    v3 = fmaf(f[u + 3], gsh[n + u + 3], v3);
                }
            }
            if (i < np) q[i] = ((v0 + v1) + (v2 + v3)) - gbg;
        }
    }
    __syncthreads();
    if (live) {
        for (int i = lane; i < np; i += 64) {
            const float sr = cp.sigma_raw[row0 + i];
            const float sg = fmaxf(sr, 0.0f);
            const float e = expf(-sg * cp.delta[row0 + i]);
            xx[i] = ((1.0f - (1.0f - e))) + 1e-10f;      // x = 1 - alpha + 1e-10 with alpha = 1 - e
        }
    }
    __syncthreads();
    if (live && lane == 0) {
        float T = 1.0f;
        for (int i = 0; i < np; ++i) { TT[i] = T; T *= xx[i]; }
    }
    __syncthreads();
    if (live) {
        for (int i = lane; i < np; i += 64) {
            const float sr = cp.sigma_raw[row0 + i];
            const float e = expf(-fmaxf(sr, 0.0f) * cp.delta[row0 + i]);
            const float alpha = 1.0f - e;
            SS[i] = q[i] * alpha * TT[i];                 // q_i w_i
        }
    }
    __syncthreads();
    if (live && lane == 0) {
        float S = 0.0f;
        for (int i = np - 1; i >= 0; --i) { const float t = SS[i]; SS[i] = S; S += t; }
    }
    __syncthreads();
    float cs = 0.0f, dsum = 0.0f;
    if (live) {
        for (int i = lane; i < np; i += 64) {
            const float sr = cp.sigma_raw[row0 + i];
            const float sg = fmaxf(sr, 0.0f);
            const float dl = cp.delta[row0 + i];
            const float e = expf(-sg * dl);
            const float alpha = 1.0f - e;
            const float dalpha = q[i] * TT[i] - SS[i] / xx[i];
            cp.wglob[row0 + i] = alpha * TT[i];
            const float dsr = sr > 0.0f ? dalpha * dl * e : 0.0f;
            cp.dsig[row0 + i] = dsr;
            dsum += dsr;
            cs = fmaf(dalpha * sg * e, dl, cs);
        }
        // padded tail rows (i >= np) carry no gradient
        for (int i = np + lane; i < cpr * CHUNK; i += 64) { cp.wglob[row0 + i] = 0.0f; cp.dsig[row0 + i] = 0.0f; }
    }
    cs += __shfl_xor(cs, 32);
    cs += __shfl_xor(cs, 16);
    cs += __shfl_xor(cs, 8);
    cs += __shfl_xor(cs, 4);
    cs += __shfl_xor(cs, 2);
    cs += __shfl_xor(cs, 1);
    dsum += __shfl_xor(dsum, 32);
    dsum += __shfl_xor(dsum, 16);
    dsum += __shfl_xor(dsum, 8);
    dsum += __shfl_xor(dsum, 4);
    dsum += __shfl_xor(dsum, 2);
    dsum += __shfl_xor(dsum, 1);
    if (live && lane == 0) {
        cp.csum[ray] = cp.accumulate ? cp.csum[ray] + cs : cs;
        cp.dsig_ray[ray] = dsum;
    }
}

// ---------------------------------------------------------------------------------------------
// geometry: per-ray partials -> dR, dT.   pts = T + m z, m = -u/u_z, l = -|u|/u_z, u = R Kinv [x y 1]
// ---------------------------------------------------------------------------------------------
struct GeoParams {
    GnrProblem prob;
    int chunks_per_ray;
    const float* geo_chunk;   // [n_chunks][8]
    const float* csum;        // [rays]
    float* part;              // [B][blocks][12]
    int blocks_per_image;
};

__global__ __launch_bounds__(256) void geo_kernel(const GeoParams gp) {
    __shared__ float red[12][256];
    const GnrProblem& p = gp.prob;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int ray = blockIdx.x * 256 + tid;
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = 0.0f;
    if (ray < p.n_rays) {
        const long rg = (long)b * p.n_rays + ray;
        const Ray r = make_ray(p, b, ray);
        float A0 = 0, A1 = 0, A2 = 0, B0 = 0, B1 = 0;
        for (int c = 0; c < gp.chunks_per_ray; ++c) {
            const float* gc = gp.geo_chunk + (rg * gp.chunks_per_ray + c) * 8;
            A0 += gc[0]; A1 += gc[1]; A2 += gc[2]; B0 += gc[3]; B1 += gc[4];
        }
        const float un = 1.0f / r.inv_n;                    // |u|
        const float iuz = 1.0f / r.uz;
        const float dLdl = gp.csum[rg] / r.l;               // sum_i dL/ddelta_i (z_{i+1}-z_i)
        const float mx = -r.ux * iuz, my = -r.uy * iuz;
        // dL/du
        const float dux = -B0 * iuz - dLdl * r.ux / (un * r.uz);
        const float duy = -B1 * iuz - dLdl * r.uy / (un * r.uz);
        const float duz = (B0 * r.ux + B1 * r.uy) * iuz * iuz + dLdl * (-1.0f / un + un * iuz * iuz);
        // v = Kinv [x y 1]
        const float x = p.xy[((long)b * 2 + 0) * p.n_rays + ray], y = p.xy[((long)b * 2 + 1) * p.n_rays + ray];
        const float* K = p.Kinv + b * 9;
        const float v0 = fmaf(K[2], 1.0f, fmaf(K[1], y, K[0] * x));
        const float v1 = fmaf(K[5], 1.0f, fmaf(K[4], y, K[3] * x));
        const float v2 = fmaf(K[8], 1.0f, fmaf(K[7], y, K[6] * x));
        v[0] = dux * v0; v[1] = dux * v1; v[2] = dux * v2;
        v[3] = duy * v0; v[4] = duy * v1; v[5] = duy * v2;
        v[6] = duz * v0; v[7] = duz * v1; v[8] = duz * v2;
        v[9] = A0; v[10] = A1;
        // z edges move with T_z (plane sweep, its jitter, FineSample's merged edges): dpts/dT_z += m, i.e.
        // dT_z = m . A (m_z = -1); explicit constant edges (edges_follow_T == 0): dT_z = A2
        v[11] = A2 + ((p.z_edges && !p.edges_follow_T) ? 0.0f : (mx * A0 + my * A1 - A2));
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) red[k][tid] = v[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
#pragma unroll
            for (int k = 0; k < 12; ++k) red[k][tid] += red[k][tid + s];
        __syncthreads();
    }
    if (tid < 12) gp.part[((long)b * gp.blocks_per_image + blockIdx.x) * 12 + tid] = red[tid][0];
}

__global__ void geo_final_kernel(const float* part, int blocks_per_image, float* dR, float* dT, const float* dR_extra) {
    const int b = blockIdx.x, k = threadIdx.x;
    if (k >= 12) return;
    float acc = 0.0f;
    for (int i = 0; i < blocks_per_image; ++i) acc += part[((long)b * blocks_per_image + i) * 12 + k];
    if (k < 9) { if (dR) dR[b * 9 + k] = dR_extra ? acc + dR_extra[b * 9 + k] : acc; }      // + the view-direction part (gnr_vd.hip)
    else if (dT) dT[b * 3 + (k - 9)] = acc;
}

// ---------------------------------------------------------------------------------------------
// latent codes: per-image bias-gradient sums -> d(shape, gaze, appea), latent columns of dW, biases
// ---------------------------------------------------------------------------------------------
struct LatentParams {
    GnrProblem prob;
    GnrWeights w;
    GnrWeightGrads dw;
    const float* dbias;       // [N_CHAIN + 1][B][H]  per-image column sums of dY (row N_CHAIN: dsig)
    float* dshape; float* dgaze; float* dappea;
    int accumulate;           // second stream adds to the first stream's latent gradients
};

// grid.x = job: 0 -> d(ext codes) from L0 and L5; 1 -> d(appea); 2.. -> weight columns / biases
__global__ void latent_codes_kernel(const LatentParams lp) {
    const GnrProblem& p = lp.prob;
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    const int ext = p.shape_dims + p.gaze_dims, vp = ENC_CH + ext;
    const int Hh = p.hidden, Hh2 = Hh / 2;                    // logical widths; dbias rows keep the stride H
    const float* db0 = lp.dbias + ((long)0 * p.batch + b) * H;
    const float* db5 = lp.dbias + ((long)5 * p.batch + b) * H;
    const float* dbr1 = lp.dbias + ((long)LR1 * p.batch + b) * H;
    // (loads issued 8 rows at a time, FMAs in row order: the serial version spent 240 us in dependent-latency loads)
    if (c < ext) {
        float acc = 0.0f;
        const float* w0 = lp.w.fea_w[0] + ENC_CH + c;
        const float* w5 = lp.w.fea_w[5] + ENC_CH + c;
        const long ld5 = vp + Hh;
        int n = 0;
        for (; n + 8 <= Hh; n += 8) {
            float a[8], b5[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a[u] = w0[(long)(n + u) * vp]; b5[u] = w5[(long)(n + u) * ld5]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc = fmaf(a[u], db0[n + u], acc);
                acc = fmaf(b5[u], db5[n + u], acc);
            }
        }
        for (; n < Hh; ++n) {
            acc = fmaf(w0[(long)n * vp], db0[n], acc);
            acc = fmaf(w5[(long)n * ld5], db5[n], acc);
        }
        float* dst = c < p.shape_dims ? (lp.dshape ? lp.dshape + b * p.shape_dims + c : nullptr)
                                      : (lp.dgaze ? lp.dgaze + b * p.gaze_dims + (c - p.shape_dims) : nullptr);
        if (dst) *dst = lp.accumulate ? *dst + acc : acc;
    }
    if (c < p.appea_dims && lp.dappea) {
        float acc = 0.0f;
        const int ld1 = Hh + p.vd_dims + p.appea_dims, a0 = Hh + p.vd_dims;      // appearance columns of RGB_layer_1
        const float* w1 = lp.w.rgb_w[1] + a0 + c;
        int n = 0;
        for (; n + 8 <= Hh2; n += 8) {
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = w1[(long)(n + u) * ld1];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(a[u], dbr1[n + u], acc);
        }
        for (; n < Hh2; ++n) acc = fmaf(w1[(long)n * ld1], dbr1[n], acc);
        float* dst = lp.dappea + b * p.appea_dims + c;
        *dst = lp.accumulate ? *dst + acc : acc;
    }
}

// latent columns of dW0 / dW5 / dW_r1 (rank-B outer products) and all bias gradients
__global__ void latent_weights_kernel(const LatentParams lp) {
    const GnrProblem& p = lp.prob;
    const int n = blockIdx.x, t = threadIdx.x;               // block per output row n < H
    const int ext = p.shape_dims + p.gaze_dims, vp = ENC_CH + ext;
    const int Hh = p.hidden, Hh2 = Hh / 2;                    // rows beyond the network's width have no parameters
    auto code = [&](int b, int c) {
        return c < p.shape_dims ? p.shape_code[b * p.shape_dims + c] : p.gaze[b * p.gaze_dims + (c - p.shape_dims)];
    };
    for (int c = t; c < ext && n < Hh; c += blockDim.x) {
        float a0 = 0.0f, a5 = 0.0f;
        for (int b = 0; b < p.batch; ++b) {
            const float cv = code(b, c);
            a0 = fmaf(lp.dbias[((long)0 * p.batch + b) * H + n], cv, a0);
            a5 = fmaf(lp.dbias[((long)5 * p.batch + b) * H + n], cv, a5);
        }
        if (lp.dw.fea_w[0]) lp.dw.fea_w[0][(long)n * vp + ENC_CH + c] = a0;
        if (lp.dw.fea_w[5]) lp.dw.fea_w[5][(long)n * (vp + Hh) + ENC_CH + c] = a5;
    }
    if (n < Hh2 && lp.dw.rgb_w[1])
        for (int c = t; c < p.appea_dims; c += blockDim.x) {
            float a = 0.0f;
            for (int b = 0; b < p.batch; ++b)
                a = fmaf(lp.dbias[((long)LR1 * p.batch + b) * H + n], p.appea_code[b * p.appea_dims + c], a);
            lp.dw.rgb_w[1][(long)n * (Hh + p.vd_dims + p.appea_dims) + Hh + p.vd_dims + c] = a;
        }
    if (n < Hh2 && lp.dw.rgb_w[1])      // view-direction columns: their gradient flows through ray_bias to the caller's fold
        for (int c = t; c < p.vd_dims; c += blockDim.x) lp.dw.rgb_w[1][(long)n * (Hh + p.vd_dims + p.appea_dims) + Hh + c] = 0.0f;
    if (t == 0) {
        auto bsum = [&](int l) {
            float a = 0.0f;
            for (int b = 0; b < p.batch; ++b) a += lp.dbias[((long)l * p.batch + b) * H + n];
            return a;
        };
        for (int l = 0; l < 8; ++l)
            if (n < Hh && lp.dw.fea_b[l]) lp.dw.fea_b[l][n] = bsum(l);
        if (n < Hh && lp.dw.rgb_b[0]) lp.dw.rgb_b[0][n] = bsum(LR0);
        if (n < Hh2 && lp.dw.rgb_b[1]) lp.dw.rgb_b[1][n] = bsum(LR1);
        if (n < p.feat_nc && lp.dw.rgb_b[2]) lp.dw.rgb_b[2][n] = bsum(LR2);
        if (n == 0 && lp.dw.density_b) lp.dw.density_b[0] = bsum(N_CHAIN);
    }
}

// ---------------------------------------------------------------------------------------------
// d loss / d ray_bias[ray][c] = sum over the ray's samples of dY of RGB_layer_1 (GnrProblem.ray_bias is added to that
// layer's bias for every sample of the ray).  One wave per ray over the saved dY_r1 dump: the fp32 kernels' S16 layout
// (16 contiguous samples per channel and sub-chunk), or the bf16x3 path's QHL quads (16 bytes {hi01, hi23, lo01, lo23} per sample; the
// slot permutation and the half swap do not matter to a sum over all 32 slots except for which half is hi).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ray_bias_grad_kernel(const float* __restrict__ dY_r1, int chunks_per_ray, long n_rays_total,
                                                            int n_out, int qhl, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays_total) return;
    if (!qhl) {
        // fp32 kernels: S16 layout (gnr_chain16.h) -- per 16-sample sub-chunk, channel c in the 64-byte row s16_row(c)
        for (int c = lane; c < n_out; c += 64) {
            float acc = 0.0f;
            for (int sb = 0; sb < 2 * chunks_per_ray; ++sb) {
                const f32x4* src = (const f32x4*)(dY_r1 + (ray * 2 * chunks_per_ray + sb) * (long)(16 * H2) + s16_row(c) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const f32x4 v = src[q]; acc += (v.x + v.y) + (v.z + v.w); }
            }
            out[ray * n_out + c] = acc;
        }
    } else {
        for (int quad = lane; 4 * quad < n_out; quad += 64) {
            float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const bool swapped = (quad >> 2) & 1;
            for (int ch = 0; ch < chunks_per_ray; ++ch) {
                const uint4* src = (const uint4*)((const char*)dY_r1 + (ray * chunks_per_ray + ch) * (long)(CHUNK * H2 * 4) + quad * 512);
                for (int j = 0; j < CHUNK; ++j) {
                    const uint4 e = src[j];
                    const unsigned h01 = swapped ? e.z : e.x, h23 = swapped ? e.w : e.y, l01 = swapped ? e.x : e.z, l23 = swapped ? e.y : e.w;
                    a[0] += __uint_as_float(h01 << 16) + __uint_as_float(l01 << 16);
                    a[1] += __uint_as_float(h01 & 0xffff0000u) + __uint_as_float(l01 & 0xffff0000u);
                    a[2] += __uint_as_float(h23 << 16) + __uint_as_float(l23 << 16);
                    a[3] += __uint_as_float(h23 & 0xffff0000u) + __uint_as_float(l23 & 0xffff0000u);
                }
            }
            for (int e = 0; e < 4; ++e)
                if (4 * quad + e < n_out) out[ray * n_out + 4 * quad + e] = a[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct BwdScratch {
    float *dsig_ray, *packedT, *gT, *wglob, *dsig, *dY_h, *dY_r0, *dY_r1, *dfeat, *geo_chunk, *csum, *geo_part;
    float *dbias, *cs_part, *wg_part, *vd;
    int geo_blocks;
};

// The arena of one weight set's queued weight-gradient GEMMs: wgrad_arena_floats()'s bound raised to the largest single need
// of the GEMMs run_bwd launches (step 5) -- in either precision, since the workspace is sized before the precision is known.
// Sized from the launchers' own plan; tests/test_host_logic.py sweeps the batch on the CPU.
static size_t bwd_wgrad_arena_floats(const GnrProblem* p) {
    const long cpi = (long)p->n_rays * ((p->n_samples + CHUNK - 1) / CHUNK);
    WgradShape g[16];
    int n = 0;
    auto add = [&](int lda, int nv, int ldb, int kv, int vec, int x3, int small) { g[n++] = WgradShape{lda, nv, ldb, kv, cpi, 0, vec, x3, small}; };
    for (int x3 = 0; x3 < 2; ++x3) {
        if (!x3 && p->feat_nc > 192 && p->feat_nc <= 288) {
            add(FEAT_PAD, 192, H2, H2, 0, 0, 0);
            add(FEAT_PAD, p->feat_nc - 192, H2, H2, 0, 0, 1);
        } else {
            add(FEAT_PAD, p->feat_nc, H2, H2, 0, x3, 0);
        }
        add(H2, H2, H, H, 0, x3, 0);            // RGB_layer_1
        add(H, H, H, H, 1, x3, 0);              // RGB_layer_0 with the density-head rider
        add(H, H, H, H, 0, x3, 0);              // the trunk's 384 x 384 layers
        add(H, H, ENC_PAD, ENC_PAD, 0, x3, 0);  // the encoding columns of layers 0 and 5
    }
    return wgrad_arena_floats_for(g, n, p->batch, H, H);
}

static size_t carve_bwd(const GnrProblem* p, char* base, BwdScratch* sc) {
    const int cpr = (p->n_samples + CHUNK - 1) / CHUNK;
    const size_t n_rays_total = (size_t)p->batch * p->n_rays;
    const size_t n_chunks = n_rays_total * cpr, M = n_chunks * CHUNK;
    size_t off = 0;
    int region = 0;
    auto take = [&](size_t floats) {
        float* ptr = base ? (float*)(base + off) : nullptr;
        off += align_up(floats * sizeof(float));
        if (CANARY_BYTES) {                                   // experimental builds (gnr_canary.h): a gap behind every region
            if (base) canary_note(base + off, "carve_bwd", region);
            off += CANARY_BYTES;
        }
        ++region;
        return ptr;
    };
    BwdScratch s{};
    s.packedT = take(PACKEDT_FLOATS + 16 * 256);      // + one ring of padding past the last row
    s.gT = take(n_rays_total * FEAT_PAD);
    s.wglob = take(M);
    s.dsig = take(M);
    s.dY_h = take((size_t)8 * M * H);
    s.dY_r0 = take(M * H);
    s.dY_r1 = take(M * H2);
    s.dfeat = take(M * FEAT_PAD);
    s.geo_chunk = take(2 * n_chunks * 8);             // fp32 kernels: one partial per 16-sample sub-chunk
    s.csum = take(n_rays_total);
    s.dsig_ray = take(n_rays_total);
    s.geo_blocks = (p->n_rays + 255) / 256;
    s.geo_part = take((size_t)p->batch * s.geo_blocks * 12);
    s.dbias = take((size_t)(N_CHAIN + 1) * p->batch * H);
    s.cs_part = nullptr;
    s.wg_part = take(bwd_wgrad_arena_floats(p));           // partial tiles of every weight-gradient GEMM of one weight set (gnr_wgrad.h)
    s.vd = vd_on_device(p) ? take(vd_bwd_floats(p, 2)) : nullptr;
    if (sc) *sc = s;
    return off;
}

size_t bwd_scratch_bytes(const GnrProblem* p, int) { return carve_bwd(p, nullptr, nullptr); }

int run_bwd(const GnrProblem* p, int n_streams, const GnrWeights* const* w, const GnrOutputGrads* dout,
            const GnrInputGrads* din, const GnrWeightGrads* const* dw, void* saved, size_t saved_bytes,
            void* scratch, size_t scratch_bytes, hipStream_t st, bool bf16x3) {
    FwdParams fp{};
    const size_t need_saved = carve_fwd(p, n_streams, true, nullptr, nullptr);
    if (!saved || saved_bytes < need_saved)
        return fail("gnr_bwd: saved workspace too small (%zu < %zu bytes); run gnr_fwd with save_for_backward",
                    saved_bytes, need_saved);
    const size_t need_scr = carve_bwd(p, nullptr, nullptr);
    if (!scratch || scratch_bytes < need_scr)
        return fail("gnr_bwd: scratch too small (%zu < %zu bytes)", scratch_bytes, need_scr);
    if (((uintptr_t)saved & 255) || ((uintptr_t)scratch & 255)) return fail("gnr_bwd: workspaces must be 256-byte aligned");
    canary_begin(false);                    // the saved workspace's gaps were filled by gnr_fwd: checked, never refilled
    carve_fwd(p, n_streams, true, (char*)saved, &fp);
    BwdScratch sc{};
    canary_fill_mode(true);
    carve_bwd(p, (char*)scratch, &sc);
    canary_arm(st);

    const int cpr = fp.chunks_per_ray;
    const long n_rays_total = (long)p->batch * p->n_rays;
    const long M = fp.M;
    const int vp = ENC_CH + p->shape_dims + p->gaze_dims;
    const int Hh = p->hidden, Hh2 = Hh / 2;
    GnrInputGrads dinz{};
    if (din) dinz = *din;
    const bool vdev = vd_on_device(p);
    VdBwdScratch vsc{};
    if (vdev) vd_carve_bwd(p, n_streams, sc.vd, &vsc);

    for (int s = 0; s < n_streams; ++s) {
        const StreamWs& ws = fp.ws[s];
        const GnrWeights& W = *w[s];
        GnrWeightGrads DW{};
        if (dw[s]) DW = *dw[s];

        // 1. upstream gradient, row-major per ray
        hipLaunchKernelGGL(gt_kernel, dim3((unsigned)((n_rays_total + 31) / 32), FEAT_PAD / 32), dim3(256), 0, st,
                           dout->feat[s], p->feat_nc, p->n_rays, n_rays_total, sc.gT);
        // 2. compositing backward
        CompBwdParams cb{};
        cb.prob = *p; cb.chunks_per_ray = cpr; cb.gT = sc.gT; cb.g_bg = dout->bg_alpha[s];
        cb.act_feat = ws.act_feat; cb.sigma_raw = ws.sigma_raw; cb.delta = fp.delta;
        cb.wglob = sc.wglob; cb.dsig = sc.dsig; cb.csum = sc.csum; cb.dsig_ray = sc.dsig_ray; cb.accumulate = s > 0;
        if (s == 0) stage_mark(GNR_STAGE_COMP_BWD, 0, st);
        hipLaunchKernelGGL(comp_bwd_kernel, dim3((unsigned)((n_rays_total + 3) / 4)), dim3(256), 0, st, cb);
        if (s == 0) stage_mark(GNR_STAGE_COMP_BWD, 1, st);
        // 3. transposed weight stream
        PackTParams pt{};
        auto setl = [&](int l, const float* wp, int ld, int n_valid, int col0, int k_valid, int enc) {
            pt.w[l] = wp; pt.ld[l] = ld; pt.n_valid[l] = n_valid; pt.col0[l] = col0; pt.k_valid[l] = k_valid; pt.enc[l] = enc;
        };
        // Hh = the network's own width (<= H): rows / columns beyond it are packed as zeros (as in launch_prep)
        setl(0, W.rgb_w[2], Hh2, p->feat_nc, 0, Hh2, 0);
        setl(1, W.rgb_w[1], Hh + p->vd_dims + p->appea_dims, Hh2, 0, Hh, 0);
        setl(2, W.rgb_w[0], Hh, Hh, 0, Hh, 0);
        setl(3, W.fea_w[7], Hh, Hh, 0, Hh, 0);
        setl(4, W.fea_w[6], Hh, Hh, 0, Hh, 0);
        setl(5, W.fea_w[5], vp + Hh, Hh, 0, ENC_PAD, 1);
        setl(6, W.fea_w[5], vp + Hh, Hh, vp, Hh, 0);
        setl(7, W.fea_w[4], Hh, Hh, 0, Hh, 0);
        setl(8, W.fea_w[3], Hh, Hh, 0, Hh, 0);
        setl(9, W.fea_w[2], Hh, Hh, 0, Hh, 0);
        setl(10, W.fea_w[1], Hh, Hh, 0, Hh, 0);
        setl(11, W.fea_w[0], vp, Hh, 0, ENC_PAD, 1);
        pt.packed = sc.packedT;
        if (bf16x3) launch_packT3(pt, st);
        else launch_packT16(pt, st);
        // 4. dgrad chain
        BwdParams bp{};
        bp.prob = *p; bp.chunks_per_ray = cpr; bp.n_chunks = fp.n_chunks; bp.M = M;
        bp.packedT = sc.packedT; bp.wsig = ws.wsig; bp.gT = sc.gT; bp.wglob = sc.wglob; bp.dsig = sc.dsig;
        bp.relu_bits = ws.relu_bits; bp.enc = fp.enc; bp.zval = fp.zval;
        bp.dY_h = sc.dY_h; bp.dY_r0 = sc.dY_r0; bp.dY_r1 = sc.dY_r1; bp.dfeat = sc.dfeat;
        bp.geo_chunk = sc.geo_chunk; bp.accumulate_geo = s > 0;
        bp.clk = clock_probe_slot(GNR_STAGE_DGRAD);
        if (s == 0) stage_mark(GNR_STAGE_DGRAD, 0, st);
        if (bf16x3)
            launch_bwd3_chain(bp, st);
        else
            launch_bwd16_chain(bp, st);
        if (s == 0) stage_mark(GNR_STAGE_DGRAD, 1, st);

        // 5. weight gradients dW = dY^T X; the same kernels emit the per-image column sums of dY
        //    (bias / latent gradients) and, for RGB_layer_0, the density-head gradient dsig^T h7.
        const float* hact = ws.act_h;
        auto hptr = [&](int l) { return hact + (size_t)l * M * H; };
        auto dyh = [&](int l) { return sc.dY_h + (size_t)l * M * H; };
        auto dbl = [&](int l) { return sc.dbias + (size_t)l * p->batch * H; };
        const long cpi = (long)p->n_rays * cpr;                       // chunks per image
        if (s == 0) stage_mark(GNR_STAGE_WGRAD, 0, st);
        // the split-K reductions of the GEMMs below are queued and run as ONE launch behind the last GEMM (gnr_wgrad.h)
        WgradDefer wd;
        wgrad_defer_init(&wd, sc.wg_part, bwd_wgrad_arena_floats(p));
        // GEMM shapes = the kernels' (H, H2); the last two arguments crop the written gradient to the network's width
        if (!bf16x3 && p->feat_nc > 192 && p->feat_nc <= 288) {
            // RGB_layer_2 (258 x 192): as ONE product its rows pad to two 192-row tiles, the second two-thirds empty
            // (1.04 ms: 0.6 of the clock's peak).  Rows 0..191 are a full 192 x 192 tile (all 256 CUs, one round); the 66 rows
            // beyond go to a 96-row tile of wgrad_kernel -- same dumps, the operand base moved 192 channels (S16 rows of 16 floats) into the sub-chunk.
            launch_wgrad(sc.dfeat, FEAT_PAD, 192, ws.act_y1, H2, H2, p->batch, cpi, DW.rgb_w[2], Hh2, 0, 0,
                         dbl(LR2), H, nullptr, nullptr, sc.wg_part, st, false, 192, Hh2, false, &wd);
            launch_wgrad(sc.dfeat + 192 * 16, FEAT_PAD, p->feat_nc - 192, ws.act_y1, H2, H2, p->batch, cpi,      // S16: 16 floats per row
                         DW.rgb_w[2] ? DW.rgb_w[2] + (size_t)192 * Hh2 : nullptr, Hh2, 0, 0, dbl(LR2) + 192, H, nullptr, nullptr,
                         sc.wg_part, st, false, p->feat_nc - 192, Hh2, true, &wd);
        } else {
            launch_wgrad(sc.dfeat, FEAT_PAD, p->feat_nc, ws.act_y1, H2, H2, p->batch, cpi, DW.rgb_w[2], Hh2, 0, 0,
                         dbl(LR2), H, nullptr, nullptr, sc.wg_part, st, bf16x3, p->feat_nc, Hh2, false, &wd);
        }
        launch_wgrad(sc.dY_r1, H2, H2, ws.act_y0, H, H, p->batch, cpi, DW.rgb_w[1], Hh + p->vd_dims + p->appea_dims, 0, 0,
                     dbl(LR1), H, nullptr, nullptr, sc.wg_part, st, bf16x3, Hh2, Hh, false, &wd);
        launch_wgrad(sc.dY_r0, H, H, hptr(7), H, H, p->batch, cpi, DW.rgb_w[0], Hh, 0, 0, dbl(LR0), H,
                     sc.dsig, DW.density_w, sc.wg_part, st, bf16x3, Hh, Hh, false, &wd);
        for (int l = 7; l >= 1; --l) {
            if (l == 5) {
                launch_wgrad(dyh(5), H, H, hptr(4), H, H, p->batch, cpi, DW.fea_w[5], vp + Hh, vp, 0, dbl(5), H,
                             nullptr, nullptr, sc.wg_part, st, bf16x3, Hh, Hh, false, &wd);
                if (DW.fea_w[5])
                    launch_wgrad(dyh(5), H, H, bf16x3 ? fp.enc3 : fp.enc, ENC_PAD, ENC_PAD, p->batch, cpi, DW.fea_w[5], vp + Hh, 0,
                                 bf16x3 ? 2 : 1, nullptr, 0, nullptr, nullptr, sc.wg_part, st, bf16x3, Hh, ENC_PAD, false, &wd);
            } else {
                launch_wgrad(dyh(l), H, H, hptr(l - 1), H, H, p->batch, cpi, DW.fea_w[l], Hh, 0, 0, dbl(l), H,
                             nullptr, nullptr, sc.wg_part, st, bf16x3, Hh, Hh, false, &wd);
            }
        }
        launch_wgrad(dyh(0), H, H, bf16x3 ? fp.enc3 : fp.enc, ENC_PAD, ENC_PAD, p->batch, cpi, DW.fea_w[0], vp, 0, bf16x3 ? 2 : 1,
                     dbl(0), H, nullptr, nullptr, sc.wg_part, st, bf16x3, Hh, ENC_PAD, false, &wd);
        wgrad_defer_flush(&wd, st);
        if (wd.failed) return fail("gnr_bwd: a weight-gradient GEMM needs more split-K scratch than the workspace holds (batch %d)", p->batch);
        if (s == 0) stage_mark(GNR_STAGE_WGRAD, 1, st);
        launch_vecsum(sc.dsig_ray, p->batch, p->n_rays, dbl(N_CHAIN), H, st);

        if (dinz.ray_bias[s] || vdev)        // device-side view direction: the per-ray sums feed launch_vd_bwd below
            hipLaunchKernelGGL(ray_bias_grad_kernel, dim3((unsigned)((n_rays_total + 3) / 4)), dim3(256), 0, st, sc.dY_r1, cpr,
                               n_rays_total, Hh2, bf16x3 ? 1 : 0, vdev ? vsc.d_rb[s] : dinz.ray_bias[s]);

        // 6. latent gradients from the per-image bias sums
        LatentParams lp{};
        lp.prob = *p; lp.w = W; lp.dw = DW; lp.dbias = sc.dbias;
        lp.dshape = dinz.shape_code; lp.dgaze = dinz.gaze; lp.dappea = dinz.appea_code; lp.accumulate = s > 0;
        const int maxc = (p->shape_dims + p->gaze_dims) > p->appea_dims ? (p->shape_dims + p->gaze_dims) : p->appea_dims;
        if (maxc > 0)
            hipLaunchKernelGGL(latent_codes_kernel, dim3((maxc + 63) / 64, p->batch), dim3(64), 0, st, lp);
        hipLaunchKernelGGL(latent_weights_kernel, dim3(H), dim3(64), 0, st, lp);
    }

    // view direction folded by the library: d W1[:, H:H+vd] (after latent_weights_kernel zeroed those columns) and the
    // direction's share of dR
    if (vdev) launch_vd_bwd(*p, n_streams, w, dw, fp.vd_embed, vsc, dinz.R != nullptr, st);
    // geometry: dR, dT
    if (dinz.R || dinz.T) {
        GeoParams gp{};
        gp.prob = *p; gp.chunks_per_ray = bf16x3 ? cpr : 2 * cpr;   /* fp32 kernels: one partial per 16-sample sub-chunk */ gp.geo_chunk = sc.geo_chunk; gp.csum = sc.csum;
        gp.part = sc.geo_part; gp.blocks_per_image = sc.geo_blocks;
        hipLaunchKernelGGL(geo_kernel, dim3(sc.geo_blocks, p->batch), dim3(256), 0, st, gp);
        hipLaunchKernelGGL(geo_final_kernel, dim3(p->batch), dim3(64), 0, st, sc.geo_part, sc.geo_blocks, dinz.R, dinz.T,
                           (vdev && dinz.R) ? vsc.dR_extra : (const float*)nullptr);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("gnr_bwd: launch failed: %s", hipGetErrorString(e));
    if (canary_check(st, "gnr_bwd")) return 1;
    return 0;
}

}  // namespace gnr
