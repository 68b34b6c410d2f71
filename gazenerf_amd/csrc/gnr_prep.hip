// gnr_prep.hip -- small memory-bound kernels around the fused MLP kernel (gfx950):
//   pack_kernel      weights [out,in] -> MFMA A-fragment stream in the chain's k-order
//   bias_kernel      per-image biases with the latent codes folded in (models/gaze_nerf.py:136-143:
//                    181 of layer-0/5 inputs and 127 of RGB_layer_1 inputs are per-image constants)
//   combine_kernel   chunk partials -> CalcRayColor outputs (utils/model_utils.py:516-534),
//                    channels-first [B,C,N_r] via an LDS transpose
//   resample_kernel  FineSample.forward (utils/model_utils.py:413-490)
//   zvals_kernel     left sample edges [B,N_r,N_p]
#include "gnr_device.h"

namespace gnr {

struct PackParams {
    const float* w[N_CHAIN];
    int ld[N_CHAIN];        // row stride of the source matrix
    int n_out[N_CHAIN];     // valid output rows
    int hcol[N_CHAIN];      // first source column of the hidden-activation inputs
    int kh[N_CHAIN];        // number of valid hidden input channels
    float* packed;
};

__global__ void pack_kernel(const PackParams pp) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < PACKED_FLOATS;
         e += (size_t)gridDim.x * blockDim.x) {
        int l = 0;
        size_t off = 0;
        while (l + 1 < N_CHAIN && e >= off + layer_packed_floats(l)) { off += layer_packed_floats(l); ++l; }
        const size_t loc = e - off;
        const int nt_n = layer_nt(l);
        const int sg = (int)(loc / ((size_t)nt_n * 256));
        const int rem = (int)(loc % ((size_t)nt_n * 256));
        const int nt = rem / 256, lane = (rem % 256) / 4, jj = rem % 4;
        const int step = 4 * sg + jj, h = lane >> 5, n = 32 * nt + (lane & 31);
        const int es = layer_enc_steps(l);
        int col = -1;
        if (step < es) {
            col = enc_channel(step, h);                       // encoding occupies source columns 0..62
        } else {
            const int k = dlayout_channel(step - es, h);
            if (k < pp.kh[l]) col = pp.hcol[l] + k;
        }
        float v = 0.0f;
        if (n < pp.n_out[l] && col >= 0) v = pp.w[l][(size_t)n * pp.ld[l] + col];
        pp.packed[e] = v;
    }
}

struct BiasParams {
    GnrProblem prob;
    GnrWeights w;
    float* bias;     // [N_CHAIN][B][H]
    float* wsig;     // [H+4]
};

__global__ void bias_kernel(const BiasParams bp) {
    const int l = blockIdx.x, b = blockIdx.y, n = threadIdx.x;     // blockDim = H
    const GnrProblem& p = bp.prob;
    const int ext = p.shape_dims + p.gaze_dims;
    const int vp = ENC_CH + ext;
    float v = 0.0f;
    if (l <= 7) {
        v = bp.w.fea_b[l][n];
        if (l == 0 || l == 5) {
            const int ld = (l == 0) ? vp : vp + H;
            const float* wr = bp.w.fea_w[l] + (size_t)n * ld + ENC_CH;
            for (int c = 0; c < p.shape_dims; ++c) v = fmaf(wr[c], p.shape_code[b * p.shape_dims + c], v);
            for (int c = 0; c < p.gaze_dims; ++c) v = fmaf(wr[p.shape_dims + c], p.gaze[b * p.gaze_dims + c], v);
        }
    } else if (l == LR0) {
        v = bp.w.rgb_b[0][n];
    } else if (l == LR1) {
        if (n < H2) {
            v = bp.w.rgb_b[1][n];
            const float* wr = bp.w.rgb_w[1] + (size_t)n * (H + p.appea_dims) + H;
            for (int c = 0; c < p.appea_dims; ++c) v = fmaf(wr[c], p.appea_code[b * p.appea_dims + c], v);
        }
    } else {
        v = n < p.feat_nc ? bp.w.rgb_b[2][n] : 0.0f;
    }
    bp.bias[((size_t)l * p.batch + b) * H + n] = v;
    if (l == 0 && b == 0) {
        bp.wsig[n] = bp.w.density_w[n];
        if (n == 0) bp.wsig[H] = bp.w.density_b[0];
    }
}

void launch_prep(const GnrProblem& p, int n_streams, const GnrWeights* const* w, StreamWs* ws,
                 hipStream_t stream, bool pack_fp32) {
    const int vp = ENC_CH + p.shape_dims + p.gaze_dims;
    for (int s = 0; s < n_streams; ++s) {
        PackParams pp;
        for (int l = 0; l < N_CHAIN; ++l) {
            if (l <= 7) {
                pp.w[l] = w[s]->fea_w[l];
                pp.ld[l] = (l == 0) ? vp : (l == 5 ? vp + H : H);
                pp.n_out[l] = H;
                pp.hcol[l] = (l == 5) ? vp : 0;
                pp.kh[l] = (l == 0) ? 0 : H;
            } else if (l == LR0) {
                pp.w[l] = w[s]->rgb_w[0]; pp.ld[l] = H; pp.n_out[l] = H; pp.hcol[l] = 0; pp.kh[l] = H;
            } else if (l == LR1) {
                pp.w[l] = w[s]->rgb_w[1]; pp.ld[l] = H + p.appea_dims; pp.n_out[l] = H2; pp.hcol[l] = 0; pp.kh[l] = H;
            } else {
                pp.w[l] = w[s]->rgb_w[2]; pp.ld[l] = H2; pp.n_out[l] = p.feat_nc; pp.hcol[l] = 0; pp.kh[l] = H2;
            }
        }
        pp.packed = ws[s].packed;
        if (pack_fp32) hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(256), 0, stream, pp);
        BiasParams bp;
        bp.prob = p; bp.w = *w[s]; bp.bias = ws[s].bias; bp.wsig = ws[s].wsig;
        hipLaunchKernelGGL(bias_kernel, dim3(N_CHAIN, p.batch), dim3(H), 0, stream, bp);
    }
}

// ---------------------------------------------------------------------------------------------
// combine: out[n] = sum_c Tpre_c * partial_c[n];  Tpre_c = prod_{c'<c} P_c'
// ---------------------------------------------------------------------------------------------
constexpr int RB = 32;          // rays per block
constexpr int MAX_CPR = 16;     // chunks per ray supported (N_p <= 512)

__global__ __launch_bounds__(256) void combine_kernel(const CombineParams cp) {
    __shared__ float tile[FEAT_PAD][RB + 1];
    __shared__ float tpre[RB][MAX_CPR];
    const GnrProblem& p = cp.prob;
    const int cpr = cp.chunks_per_ray;
    const long n_rays_total = (long)p.batch * p.n_rays;
    const long ray0 = (long)blockIdx.x * RB;
    const int s = blockIdx.y;
    const int tid = threadIdx.x;
    const float* pf = cp.part_feat[s];
    const float* ps = cp.part_sc[s];

    if (tid < RB) {
        const long rg = ray0 + tid;
        float T = 1.0f, accw = 0.0f, dsum = 0.0f;
        if (rg < n_rays_total) {
            for (int c = 0; c < cpr; ++c) {
                const f32x4 sc = *(const f32x4*)(ps + (rg * cpr + c) * 4);
                tpre[tid][c] = T;
                accw = fmaf(T, sc.y, accw);
                dsum = fmaf(T, sc.z, dsum);
                T *= sc.x;
            }
            const int b = (int)(rg / p.n_rays);
            const long r = rg - (long)b * p.n_rays;
            cp.out.bg_alpha[s][(long)b * p.n_rays + r] = 1.0f - accw;
            if (cp.out.depth[s]) cp.out.depth[s][(long)b * p.n_rays + r] = dsum;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < RB * FEAT_PAD; idx += 256) {
        const int rl = idx / FEAT_PAD, n = idx - rl * FEAT_PAD;
        const long rg = ray0 + rl;
        float acc = 0.0f;
        if (rg < n_rays_total)
            for (int c = 0; c < cpr; ++c) acc = fmaf(tpre[rl][c], pf[(rg * cpr + c) * FEAT_PAD + n], acc);
        tile[n][rl] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < p.feat_nc * RB; idx += 256) {
        const int n = idx / RB, rl = idx - n * RB;
        const long rg = ray0 + rl;
        if (rg < n_rays_total) {
            const int b = (int)(rg / p.n_rays);
            const long r = rg - (long)b * p.n_rays;
            cp.out.feat[s][((long)b * p.feat_nc + n) * p.n_rays + r] = tile[n][rl];
        }
    }
    if (cp.out.weights[s]) {
        const int np = p.n_samples;
        for (int idx = tid; idx < RB * np; idx += 256) {
            const int rl = idx / np, i = idx - rl * np;
            const long rg = ray0 + rl;
            if (rg < n_rays_total) {
                const int c = i / CHUNK;
                cp.out.weights[s][rg * np + i] =
                    cp.wl[s][(rg * cpr + c) * CHUNK + (i - c * CHUNK)] * tpre[rl][c];
            }
        }
    }
}

void launch_combine(const CombineParams& cp, hipStream_t stream) {
    const long n_rays_total = (long)cp.prob.batch * cp.prob.n_rays;
    const unsigned gx = (unsigned)((n_rays_total + RB - 1) / RB);
    hipLaunchKernelGGL(combine_kernel, dim3(gx, cp.n_streams), dim3(256), 0, stream, cp);
}

// ---------------------------------------------------------------------------------------------
// zvals: left edges of every sample, [B,N_r,N_p]
// ---------------------------------------------------------------------------------------------
__global__ void zvals_kernel(const GnrProblem p, float* out) {
    const long total = (long)p.batch * p.n_rays * p.n_samples;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long rg = e / p.n_samples;
        const int i = (int)(e - rg * p.n_samples);
        const int b = (int)(rg / p.n_rays);
        out[e] = sample_edge(p, p.T[b * 3 + 2], rg, i);
    }
}

void launch_zvals(const GnrProblem& p, float* out, hipStream_t stream) {
    const long total = (long)p.batch * p.n_rays * p.n_samples;
    const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(zvals_kernel, dim3(grid ? grid : 1), dim3(256), 0, stream, p, out);
}

// ---------------------------------------------------------------------------------------------
// resample: FineSample.forward, utils/model_utils.py:413-490.  One 64-lane wave per ray.
//   pdf  = w[1:-1] / sum(w[1:-1] + 1e-5);  cdf = [0, cumsum(pdf)]          (N_c - 1 entries)
//   inds = searchsorted(cdf, u, right=True); below = max(0, inds-1); above = min(N_c-2, inds)
//   bins = midpoints of the coarse z;  z_f = bins[below] + t (bins[above]-bins[below])
//   out  = sort(cat[coarse z, z_f])
// ---------------------------------------------------------------------------------------------
constexpr int RS_MAX = 512;     // max merged edges per ray

__global__ __launch_bounds__(64) void resample_kernel(const float* __restrict__ w,
                                                      const float* __restrict__ cz,
                                                      const float* __restrict__ u, long n_rays,
                                                      int nc, int nf, float* __restrict__ zout) {
    __shared__ float cdf[RS_MAX];
    __shared__ float bins[RS_MAX];
    __shared__ float merged[RS_MAX];
    const long ray = blockIdx.x;
    if (ray >= n_rays) return;
    const int lane = threadIdx.x;
    const int nc2 = nc - 2, nfs = nf + 1, total = nc + nfs;
    const float* wr = w + ray * nc;
    const float* zr = cz + ray * nc;
    // sum(w' + 1e-5): sequential fp32 sum to mirror torch.sum on a short row is not reproducible
    // bit-for-bit anyway; one lane does it in index order.
    if (lane == 0) {
        float ssum = 0.0f;
        for (int k = 0; k < nc2; ++k) ssum += wr[1 + k] + 1e-5f;
        float c = 0.0f;
        cdf[0] = 0.0f;
        for (int k = 0; k < nc2; ++k) { c += wr[1 + k] / ssum; cdf[k + 1] = c; }
    }
    for (int k = lane; k < nc - 1; k += 64) bins[k] = 0.5f * (zr[k + 1] + zr[k]);
    for (int k = lane; k < nc; k += 64) merged[k] = zr[k];
    __syncthreads();
    const float ustep = 1.0f / (float)(nfs - 1);
    for (int q = lane; q < nfs; q += 64) {
        float uq;
        if (u) uq = u[ray * nfs + q];
        else uq = (q < nfs / 2) ? ustep * (float)q : 1.0f - ustep * (float)(nfs - 1 - q);
        // searchsorted(right=True): first index with cdf[idx] > uq, over nc2+1 entries
        int lo = 0, hi = nc2 + 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= uq) lo = mid + 1; else hi = mid; }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < nc2 ? lo : nc2;
        float denom = cdf[above] - cdf[below];
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (uq - cdf[below]) / denom;
        merged[nc + q] = bins[below] + t * (bins[above] - bins[below]);
    }
    __syncthreads();
    // rank sort (stable for ties by index): total <= 512, 64 lanes
    for (int a = lane; a < total; a += 64) {
        const float va = merged[a];
        int rank = 0;
        for (int b2 = 0; b2 < total; ++b2) {
            const float vb = merged[b2];
            rank += (vb < va) || (vb == va && b2 < a);
        }
        zout[ray * total + rank] = va;
    }
}

void launch_resample(const float* w, const float* cz, const float* u, long n_rays, int nc, int nf,
                     float* zout, hipStream_t stream) {
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)n_rays), dim3(64), 0, stream, w, cz, u, n_rays, nc, nf, zout);
}

}  // namespace gnr
