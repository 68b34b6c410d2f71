// gnr_prep.hip -- small memory-bound kernels around the fused MLP kernel (gfx950):
//   pack16_kernel    weights [out,in] -> MFMA A-fragment stream in the chain's k-order
//   bias_kernel      per-image biases with the latent codes folded in (models/gaze_nerf.py:136-143:
//                    181 of layer-0/5 inputs and 127 of RGB_layer_1 inputs are per-image constants)
//   combine_kernel   chunk partials -> CalcRayColor outputs (utils/model_utils.py:516-534),
//                    channels-first [B,C,N_r] via an LDS transpose
//   resample_kernel  FineSample.forward (utils/model_utils.py:413-490)
//   zvals_kernel     left sample edges [B,N_r,N_p]
#include "gnr_chain16.h"

namespace gnr {

struct PackParams {
    const float* w[N_CHAIN];
    int ld[N_CHAIN];        // row stride of the source matrix
    int n_out[N_CHAIN];     // valid output rows
    int hcol[N_CHAIN];      // first source column of the hidden-activation inputs
    int kh[N_CHAIN];        // number of valid hidden input channels
    float* packed;
};

// The weight stream of the 16x16x4 chain (gnr_chain16.h): rows (k-group of 16 input channels, n-tile of 16 outputs),
// lane l = (output row l&15, k = l>>4), component e = k-step.
__global__ void pack16_kernel(const PackParams pp) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < PACKED_FLOATS;
         e += (size_t)gridDim.x * blockDim.x) {
        int l = 0;
        size_t off = 0;
        while (l + 1 < N_CHAIN && e >= off + layer_packed_floats(l)) { off += layer_packed_floats(l); ++l; }
        const size_t loc = e - off;
        const int nt_n = 2 * layer_nt(l);
        const int rowi = (int)(loc / 256), rem = (int)(loc % 256);
        const int kg = rowi / nt_n, nt = rowi % nt_n;
        const int lane = rem / 4, ee = rem % 4;
        const int n = 16 * nt + (lane & 15), gk = lane >> 4;
        const int ekg = layer_enc_steps(l) ? NT16_E : 0;
        int col = -1;
        if (kg < ekg) {
            col = enc16_channel(4 * kg + ee, gk);             // encoding occupies source columns 0..62
        } else {
            const int k = d16_channel(kg - ekg, ee, gk);
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

// v + sum_c w[c] x[c] as one fmaf chain in index order; the loads are issued 8 at a time (a serial load-fma-load-fma
// chain over the 181 latent columns cost 55 us per weight set and call: 1-3 % of a 64 x 64-ray inference)
__device__ __forceinline__ float dot_in_order(const float* __restrict__ w, const float* __restrict__ x, int n, float v) {
    int c = 0;
    for (; c + 8 <= n; c += 8) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = w[c + u]; b[u] = x[c + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) v = fmaf(a[u], b[u], v);
    }
    for (; c < n; ++c) v = fmaf(w[c], x[c], v);
    return v;
}

__global__ void bias_kernel(const BiasParams bp) {
    const int l = blockIdx.x, b = blockIdx.y, n = threadIdx.x;     // blockDim = H
    const GnrProblem& p = bp.prob;
    const int ext = p.shape_dims + p.gaze_dims;
    const int vp = ENC_CH + ext;
    const int Hh = p.hidden, Hh2 = Hh / 2;      // channels beyond the network's width: bias 0 (their weights are packed as 0)
    float v = 0.0f;
    if (l <= 7) {
        if (n < Hh) {
            v = bp.w.fea_b[l][n];
            if (l == 0 || l == 5) {
                const int ld = (l == 0) ? vp : vp + Hh;
                const float* wr = bp.w.fea_w[l] + (size_t)n * ld + ENC_CH;
                v = dot_in_order(wr, p.shape_code + b * p.shape_dims, p.shape_dims, v);
                v = dot_in_order(wr + p.shape_dims, p.gaze + b * p.gaze_dims, p.gaze_dims, v);
            }
        }
    } else if (l == LR0) {
        if (n < Hh) v = bp.w.rgb_b[0][n];
    } else if (l == LR1) {
        if (n < Hh2) {
            v = bp.w.rgb_b[1][n];
            // columns [Hh, Hh + vd_dims) belong to the view-direction embedding: GnrProblem.ray_bias carries them
            const float* wr = bp.w.rgb_w[1] + (size_t)n * (Hh + p.vd_dims + p.appea_dims) + Hh + p.vd_dims;
            v = dot_in_order(wr, p.appea_code + b * p.appea_dims, p.appea_dims, v);
        }
    } else {
        v = n < p.feat_nc ? bp.w.rgb_b[2][n] : 0.0f;
    }
    bp.bias[((size_t)l * p.batch + b) * H + n] = v;
    if (l == 0 && b == 0) {
        bp.wsig[n] = n < Hh ? bp.w.density_w[n] : 0.0f;
        if (n == 0) bp.wsig[H] = bp.w.density_b[0];
    }
}

void launch_prep(const GnrProblem& p, int n_streams, const GnrWeights* const* w, StreamWs* ws,
                 hipStream_t stream, bool pack_fp32) {
    const int vp = ENC_CH + p.shape_dims + p.gaze_dims;
    const int Hh = p.hidden, Hh2 = Hh / 2;      // the network's own width; rows / columns beyond it are packed as zeros
    for (int s = 0; s < n_streams; ++s) {
        PackParams pp;
        for (int l = 0; l < N_CHAIN; ++l) {
            if (l <= 7) {
                pp.w[l] = w[s]->fea_w[l];
                pp.ld[l] = (l == 0) ? vp : (l == 5 ? vp + Hh : Hh);
                pp.n_out[l] = Hh;
                pp.hcol[l] = (l == 5) ? vp : 0;
                pp.kh[l] = (l == 0) ? 0 : Hh;
            } else if (l == LR0) {
                pp.w[l] = w[s]->rgb_w[0]; pp.ld[l] = Hh; pp.n_out[l] = Hh; pp.hcol[l] = 0; pp.kh[l] = Hh;
            } else if (l == LR1) {
                pp.w[l] = w[s]->rgb_w[1]; pp.ld[l] = Hh + p.vd_dims + p.appea_dims; pp.n_out[l] = Hh2; pp.hcol[l] = 0; pp.kh[l] = Hh;
            } else {
                pp.w[l] = w[s]->rgb_w[2]; pp.ld[l] = Hh2; pp.n_out[l] = p.feat_nc; pp.hcol[l] = 0; pp.kh[l] = Hh2;
            }
        }
        pp.packed = ws[s].packed;
        if (pack_fp32) hipLaunchKernelGGL(pack16_kernel, dim3(1024), dim3(256), 0, stream, pp);
        BiasParams bp;
        bp.prob = p; bp.w = *w[s]; bp.bias = ws[s].bias; bp.wsig = ws[s].wsig;
        hipLaunchKernelGGL(bias_kernel, dim3(N_CHAIN, p.batch), dim3(H), 0, stream, bp);
    }
}

// ---------------------------------------------------------------------------------------------
// combine: out[n] = sum_c Tpre_c * partial_c[n];  Tpre_c = prod_{c'<c} P_c'
// ---------------------------------------------------------------------------------------------
constexpr int RB = 32;          // rays per block
constexpr int MAX_CPR = 32;     // (sub-)chunks per ray supported (N_p <= 512, 16-sample sub-chunks)

__global__ __launch_bounds__(256) void combine_kernel(const CombineParams cp) {
    __shared__ float tile[FEAT_PAD][RB + 1];
    __shared__ float tpre[RB][MAX_CPR];
    const GnrProblem& p = cp.prob;
    const int cpr = cp.chunks_per_ray, clen = cp.chunk_len;      // partials per ray, samples per partial
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
                const int c = i / clen;
                cp.out.weights[s][rg * np + i] =
                    cp.wl[s][(rg * cpr + c) * clen + (i - c * clen)] * tpre[rl][c];
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
// resample: FineSample.forward, utils/model_utils.py:413-490.
//   pdf  = w[1:-1] / sum(w[1:-1] + 1e-5);  cdf = [0, cumsum(pdf)]          (N_c - 1 entries)
//   inds = searchsorted(cdf, u, right=True); below = max(0, inds-1); above = min(N_c-2, inds)
//   bins = midpoints of the coarse z;  z_f = bins[below] + t (bins[above]-bins[below])
//   out  = sort(cat[coarse z, z_f])
//
// HBM-bound by nature (reads 2 N_c floats, writes N_c + N_f + 1 per ray: 1.3 KB at 64 + 128), so the kernel is
// organised around coalesced rows and no serial lane:
//   * a wave takes a PASS of up to 64 consecutive rays.  Their weight rows are one contiguous block: it is
//     streamed coalesced into an LDS image with an odd row stride, so that in the next step LANE r walks ROW r
//     conflict-free;
//   * cdf: torch.cumsum adds in index order and t = (u - cdf[below]) / (cdf[above] - cdf[below]) divides by
//     bin masses down to 1e-5, so an ulp of cdf is 6e-3 of a bin: the running sum must keep the sequential
//     fp32 order.  It is sequential per ray and parallel over the 64 rays of the pass (one ray per lane);
//   * then ray by ray with the whole wave: the fine samples by binary search in the ray's cdf row, and the
//     sort as a MERGE of two sorted lists instead of a sort: the coarse z are sorted, a fine sample's position
//     among them follows from its bin (z_f lies between two bin midpoints: one or two comparisons), its
//     position among the fine samples is its index (u ascending: the inverse cdf is monotone in fp32 too) or
//     its rank among the u's (random u); a 65-bin histogram of the fine samples' coarse positions + one wave
//     scan gives every coarse z its place.  The merged row leaves through LDS as coalesced stores.
// ---------------------------------------------------------------------------------------------
constexpr int RS_WAVES = 4;          // waves per workgroup (each on its own pass)
constexpr int RS_ROW_FLOATS = 2048;  // LDS floats per wave for the weight / cdf rows: 32 rays per pass at N_c = 64,
                                     // ~10 KB per wave -> 16 waves per CU (64 rays per pass left 8: latency-bound)

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");     // LDS traffic of this wave is in order; make it visible
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64 * RS_WAVES) void resample_kernel(const float* __restrict__ w,
                                                                 const float* __restrict__ cz,
                                                                 const float* __restrict__ u, long n_rays, int nc,
                                                                 int nf, int rpp, int stride,
                                                                 float* __restrict__ zout) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc2 = nc - 2, nfs = nf + 1, total = nc + nfs;
    const int per_wave = rpp * stride + nc + (nc + 1) + total + nfs;
    float* rows = smem + (size_t)wave * per_wave;       // [rpp][stride]: w' then cdf, row r = ray r of the pass
    float* zs = rows + rpp * stride;                    // [nc]      coarse z of the current ray
    int* cnt = (int*)(zs + nc);                         // [nc + 1]  histogram of coarse positions
    float* merged = (float*)(cnt + nc + 1);             // [total]
    float* us = merged + total;                         // [nfs]     u of the current ray (random mode)
    const long pass = (long)blockIdx.x * RS_WAVES + wave;
    const long ray0 = pass * rpp;
    if (ray0 >= n_rays) return;
    const int nr = (int)((n_rays - ray0) < rpp ? (n_rays - ray0) : rpp);

    // 1. weight rows of the pass: one contiguous block, coalesced, into the odd-stride image
    {
        const float* src = w + ray0 * nc;
        const int n = nr * nc;
        for (int e = lane; e < n; e += 64) {
            const int r = e / nc, k = e - r * nc;
            if (k >= 1 && k <= nc2) rows[r * stride + (k - 1)] = src[e];
        }
    }
    wave_sync();
    // 2. lane r: sum and sequential cumsum of row r, in place (cdf[k + 1] replaces w'[k + 1] after it was read)
    if (lane < nr) {
        float* row = rows + lane * stride;
        float ssum = 0.0f;
        for (int k = 0; k < nc2; ++k) ssum += row[k] + 1e-5f;
        float c = 0.0f, cur = row[0];
        row[0] = 0.0f;
        for (int k = 0; k < nc2; ++k) {
            const float wk = cur;
            if (k + 1 < nc2) cur = row[k + 1];
            c += wk / ssum;
            row[k + 1] = c;
        }
    }
    wave_sync();
    // 3. ray by ray, the whole wave
    const float ustep = 1.0f / (float)(nfs - 1);
    for (int r = 0; r < nr; ++r) {
        const long ray = ray0 + r;
        const float* cdf = rows + r * stride;           // nc2 + 1 entries
        for (int k = lane; k < nc; k += 64) zs[k] = cz[ray * nc + k];
        for (int k = lane; k <= nc; k += 64) cnt[k] = 0;
        if (u)
            for (int q = lane; q < nfs; q += 64) us[q] = u[ray * nfs + q];
        wave_sync();
        for (int q = lane; q < nfs; q += 64) {
            float uq;
            if (u) uq = us[q];
            else uq = (q < nfs / 2) ? ustep * (float)q : 1.0f - ustep * (float)(nfs - 1 - q);     // torch.linspace
            // searchsorted(right=True): first index with cdf[idx] > uq, over nc2 + 1 entries
            int lo = 0, hi = nc2 + 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= uq) lo = mid + 1; else hi = mid; }
            const int below = lo - 1 > 0 ? lo - 1 : 0;
            const int above = lo < nc2 ? lo : nc2;
            const float cb = cdf[below];
            float denom = cdf[above] - cb;
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (uq - cb) / denom;
            const float bb = __fmul_rn(0.5f, __fadd_rn(zs[below + 1], zs[below]));
            const float ba = __fmul_rn(0.5f, __fadd_rn(zs[above + 1], zs[above]));
            const float zf = __fadd_rn(bb, __fmul_rn(t, __fsub_rn(ba, bb)));
            // position among the coarse z = #{k : z[k] <= zf}: z[0..below] <= bins[below] <= zf <= bins[above]
            // <= z[above + 1], so at most two more comparisons succeed (more only for duplicate coarse z)
            int cq = below + 1;
            while (cq < nc && zs[cq] <= zf) ++cq;
            // position among the fine samples
            int rq = q;
            if (u) {
                rq = 0;
                for (int j = 0; j < nfs; ++j) {
                    const float uj = us[j];
                    rq += (uj < uq) || (uj == uq && j < q);
                }
            }
            atomicAdd(&cnt[cq], 1);
            merged[cq + rq] = zf;
        }
        wave_sync();
        // coarse k goes to k + #{fine with coarse position <= k}: inclusive scan of the histogram
        int carry = 0;
        for (int k0 = 0; k0 < nc; k0 += 64) {
            const int k = k0 + lane;
            int v = k < nc ? cnt[k] : 0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(v, d);
                if (lane >= d) v += o;
            }
            if (k < nc) merged[k + carry + v] = zs[k];
            carry += __shfl(v, 63);
        }
        wave_sync();
        for (int i = lane; i < total; i += 64) zout[ray * total + i] = merged[i];
        wave_sync();
    }
}

void launch_resample(const float* w, const float* cz, const float* u, long n_rays, int nc, int nf,
                     float* zout, hipStream_t stream) {
    const int stride = (nc - 1) | 1;                      // odd: lane r walking row r touches every bank once
    int rpp = RS_ROW_FLOATS / stride;
    if (rpp > 64) rpp = 64;
    const int total = nc + nf + 1;
    const size_t lds = (size_t)RS_WAVES * ((size_t)rpp * stride + nc + (nc + 1) + total + (nf + 1)) * sizeof(float);
    const long passes = (n_rays + rpp - 1) / rpp;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((passes + RS_WAVES - 1) / RS_WAVES)), dim3(64 * RS_WAVES), lds,
                       stream, w, cz, u, n_rays, nc, nf, rpp, stride, zout);
}

}  // namespace gnr
