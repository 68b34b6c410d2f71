// gnr_vd.hip -- the reference's view-direction option on the device (gfx950).
//
// `include_vd` (models/gaze_nerf.py:70-80, 140-143, 240-243): RGB_layer_1 also sees vd_encoder(dirs), the Embedder
// (utils/model_utils.py:253-280; N_freqs = 4, include_input -> 27 channels) of the normalised ray direction
// (utils/model_utils.py:366-369, 317), between the hidden and the appearance columns.  The direction is constant along
// a ray, so W[:, H:H+27] . embed(dir) is a per-ray bias of that layer; the chain kernels take it as such
// (GnrProblem.ray_bias / StreamWs.ray_bias).  Rounds 1-2 had the CALLER compute it (torch autograd); here the C ABI
// does it alone when vd_dims > 0 and GnrProblem.ray_bias is NULL:
//   vd_fwd_kernel       dirs -> embedding [rays][28] (kept for the backward) -> ray_bias [rays][H/2] per weight set
//   vd_bwd_ray_kernel   d ray_bias -> d embedding -> d direction -> d(R Kinv [x y 1]) -> per-block partial sums of dR
//   vd_bwd_w_kernel     d W[:, H:H+27] = sum over rays of d ray_bias (x) embedding, partial sums per ray segment
//   vd_bwd_final_kernel fixed-order sums of both (deterministic)
// A few MFLOP per call: HBM / latency bound, off the critical path.
#include "gnr_device.h"

namespace gnr {
int fail(const char* fmt, ...);

constexpr int VD_PAD = 28;          // floats per ray in the saved embedding (27 used with the reference's 4 frequencies)
constexpr int VD_MAX = 51;          // 3 + 6 * 8 frequencies at most
constexpr int VD_RPB = 16;          // rays per block (forward / ray backward)
constexpr int VD_SEG_RAYS = 1024;   // rays per partial sum of the weight gradient

struct VdParams {
    GnrProblem prob;
    int n_streams;
    const float* w1[2];          // RGB_layer_1.weight [H/2][ld] of each weight set
    int ld;                      // hidden + vd_dims + appea_dims
    float* embed;                // [rays][VD_PAD]
    float* ray_bias[2];          // [rays][hidden/2]                       (forward: written)
    const float* d_ray_bias[2];  // [rays][hidden/2]                       (backward: read)
    float* dR_part;              // [B][blocks_per_image][9]
    float* dW_part;              // [n_streams][segs][vd_dims][hidden/2]
    float* dW1[2];               // gradient of RGB_layer_1.weight (columns hidden .. hidden + vd_dims are written)
    float* dR_extra;             // [B][9]
    int blocks_per_image, segs;
};

// Embedder: [d(3) | sin(2^f d)(3) | cos(2^f d)(3), f = 0..nf-1]
__device__ __forceinline__ void vd_embed(float dx, float dy, float dz, int nf, float* e) {
    const float d[3] = {dx, dy, dz};
    e[0] = dx; e[1] = dy; e[2] = dz;
    for (int f = 0; f < nf; ++f) {
        const float sc = (float)(1 << f);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float s, c;
            sincosf(d[a] * sc, &s, &c);
            e[3 + 6 * f + a] = s;
            e[3 + 6 * f + 3 + a] = c;
        }
    }
}

__global__ __launch_bounds__(256) void vd_fwd_kernel(const VdParams vp) {
    __shared__ float emb[VD_RPB][VD_MAX + 1];
    const GnrProblem& p = vp.prob;
    const int vd = p.vd_dims, nf = (vd - 3) / 6, Hh = p.hidden, nout = Hh / 2;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int ray0 = blockIdx.x * VD_RPB;
    if (tid < VD_RPB && ray0 + tid < p.n_rays) {
        const Ray r = make_ray(p, b, ray0 + tid);
        float e[VD_MAX];
        vd_embed(r.dx, r.dy, r.dz, nf, e);
        float* dst = vp.embed + ((long)b * p.n_rays + ray0 + tid) * VD_PAD;
        for (int k = 0; k < vd; ++k) {
            emb[tid][k] = e[k];
            if (k < VD_PAD) dst[k] = e[k];
        }
    }
    __syncthreads();
    if (tid >= nout) return;
    for (int s = 0; s < vp.n_streams; ++s) {
        float w[VD_MAX];
        const float* wr = vp.w1[s] + (long)tid * vp.ld + Hh;
        for (int k = 0; k < vd; ++k) w[k] = wr[k];
        for (int rr = 0; rr < VD_RPB && ray0 + rr < p.n_rays; ++rr) {
            float acc = 0.0f;
            for (int k = 0; k < vd; ++k) acc = fmaf(w[k], emb[rr][k], acc);
            vp.ray_bias[s][((long)b * p.n_rays + ray0 + rr) * nout + tid] = acc;
        }
    }
}

// d ray_bias [rays][nout] (both weight sets) -> d embedding -> d direction -> dR partial of this block
__global__ __launch_bounds__(256) void vd_bwd_ray_kernel(const VdParams vp) {
    __shared__ float drb[VD_RPB][H2 + 1];
    __shared__ float de[VD_RPB][VD_MAX + 1];
    __shared__ float red[VD_RPB][9];
    const GnrProblem& p = vp.prob;
    const int vd = p.vd_dims, nf = (vd - 3) / 6, Hh = p.hidden, nout = Hh / 2;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int ray0 = blockIdx.x * VD_RPB;
    for (int i = tid; i < VD_RPB * (VD_MAX + 1); i += 256) (&de[0][0])[i] = 0.0f;
    __syncthreads();
    for (int s = 0; s < vp.n_streams; ++s) {
        for (int i = tid; i < VD_RPB * nout; i += 256) {
            const int rr = i / nout, c = i - rr * nout;
            drb[rr][c] = ray0 + rr < p.n_rays ? vp.d_ray_bias[s][((long)b * p.n_rays + ray0 + rr) * nout + c] : 0.0f;
        }
        __syncthreads();
        // thread (rr, k): d embed[k] += sum_c d_rb[rr][c] W[c][H + k]   (fixed order in c: deterministic)
        for (int i = tid; i < VD_RPB * vd; i += 256) {
            const int rr = i / vd, k = i - rr * vd;
            const float* wc = vp.w1[s] + Hh + k;
            float acc = 0.0f;
            for (int c = 0; c < nout; ++c) acc = fmaf(drb[rr][c], wc[(long)c * vp.ld], acc);
            de[rr][k] += acc;
        }
        __syncthreads();
    }
    if (tid < VD_RPB) {
        float o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (ray0 + tid < p.n_rays) {
            const int ray = ray0 + tid;
            const Ray r = make_ray(p, b, ray);
            const float* e = vp.embed + ((long)b * p.n_rays + ray) * VD_PAD;
            const float d[3] = {r.dx, r.dy, r.dz};
            float dd[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float acc = de[tid][a];
                for (int f = 0; f < nf; ++f) {
                    const float sc = (float)(1 << f);
                    const float sv = 3 + 6 * f + a < VD_PAD ? e[3 + 6 * f + a] : sinf(d[a] * sc);
                    const float cv = 3 + 6 * f + 3 + a < VD_PAD ? e[3 + 6 * f + 3 + a] : cosf(d[a] * sc);
                    acc += sc * (cv * de[tid][3 + 6 * f + a] - sv * de[tid][3 + 6 * f + 3 + a]);
                }
                dd[a] = acc;
            }
            // d = u / |u|:  du = (dd - d (d . dd)) / |u|;  u = R v, v = Kinv [x y 1]:  dR[i][j] += du[i] v[j]
            const float dot = d[0] * dd[0] + d[1] * dd[1] + d[2] * dd[2];
            const float du[3] = {(dd[0] - d[0] * dot) * r.inv_n, (dd[1] - d[1] * dot) * r.inv_n, (dd[2] - d[2] * dot) * r.inv_n};
            const float x = p.xy[((long)b * 2 + 0) * p.n_rays + ray], y = p.xy[((long)b * 2 + 1) * p.n_rays + ray];
            const float* K = p.Kinv + b * 9;
            const float v[3] = {fmaf(K[2], 1.0f, fmaf(K[1], y, K[0] * x)), fmaf(K[5], 1.0f, fmaf(K[4], y, K[3] * x)),
                                fmaf(K[8], 1.0f, fmaf(K[7], y, K[6] * x))};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int jx = 0; jx < 3; ++jx) o[3 * i + jx] = du[i] * v[jx];
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) red[tid][q] = o[q];
    }
    __syncthreads();
    if (tid < 9) {
        float acc = 0.0f;
        for (int rr = 0; rr < VD_RPB; ++rr) acc += red[rr][tid];
        vp.dR_part[((long)b * vp.blocks_per_image + blockIdx.x) * 9 + tid] = acc;
    }
}

// dW_part[s][seg][k][c] = sum over the segment's rays of d_rb[s][ray][c] * embed[ray][k]
__global__ __launch_bounds__(256) void vd_bwd_w_kernel(const VdParams vp) {
    const GnrProblem& p = vp.prob;
    const int vd = p.vd_dims, nout = p.hidden / 2;
    const int k = blockIdx.x, seg = blockIdx.y, s = blockIdx.z, c = threadIdx.x;
    const long n_total = (long)p.batch * p.n_rays;
    const long r0 = (long)seg * VD_SEG_RAYS, r1 = r0 + VD_SEG_RAYS < n_total ? r0 + VD_SEG_RAYS : n_total;
    if (c >= nout) return;
    float acc = 0.0f;
    const int nf = (vd - 3) / 6;
    for (long ray = r0; ray < r1; ++ray) {
        float e;
        if (k < VD_PAD) {
            e = vp.embed[ray * VD_PAD + k];
        } else {            // more than 4 frequencies: recompute the entries the saved rows do not hold
            const int b = (int)(ray / p.n_rays);
            const Ray r = make_ray(p, b, (int)(ray - (long)b * p.n_rays));
            float ee[VD_MAX];
            vd_embed(r.dx, r.dy, r.dz, nf, ee);
            e = ee[k];
        }
        acc = fmaf(vp.d_ray_bias[s][ray * nout + c], e, acc);
    }
    vp.dW_part[(((long)s * vp.segs + seg) * vd + k) * nout + c] = acc;
}

__global__ void vd_bwd_final_kernel(const VdParams vp) {
    const GnrProblem& p = vp.prob;
    const int vd = p.vd_dims, nout = p.hidden / 2, Hh = p.hidden;
    const long n_w = (long)vp.n_streams * vd * nout;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_w) {
        const int s = (int)(i / ((long)vd * nout)), rem = (int)(i - (long)s * vd * nout), k = rem / nout, c = rem - k * nout;
        if (vp.dW1[s]) {
            float acc = 0.0f;
            for (int seg = 0; seg < vp.segs; ++seg) acc += vp.dW_part[(((long)s * vp.segs + seg) * vd + k) * nout + c];
            vp.dW1[s][(long)c * vp.ld + Hh + k] = acc;
        }
    } else if (i < n_w + (long)p.batch * 9 && vp.dR_extra) {
        const int q = (int)(i - n_w), b = q / 9, e = q - 9 * b;
        float acc = 0.0f;
        for (int blk = 0; blk < vp.blocks_per_image; ++blk) acc += vp.dR_part[((long)b * vp.blocks_per_image + blk) * 9 + e];
        vp.dR_extra[b * 9 + e] = acc;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------
static inline size_t vd_align(size_t x) { return (x + 255) & ~(size_t)255; }

bool vd_on_device(const GnrProblem* p) { return p->vd_dims > 0 && !p->ray_bias[0] && !p->ray_bias[1]; }

// floats the forward workspace keeps for the device-side view direction: embedding + one ray_bias per weight set
size_t vd_fwd_floats(const GnrProblem* p, int n_streams) {
    if (!vd_on_device(p)) return 0;
    const size_t rays = (size_t)p->batch * p->n_rays;
    return vd_align(rays * VD_PAD * 4) / 4 + (size_t)n_streams * (vd_align(rays * (p->hidden / 2) * 4) / 4);
}
void vd_carve_fwd(const GnrProblem* p, int n_streams, float* base, float** embed, float** rb) {
    const size_t rays = (size_t)p->batch * p->n_rays;
    *embed = base;
    float* q = base + vd_align(rays * VD_PAD * 4) / 4;
    for (int s = 0; s < n_streams; ++s) { rb[s] = q; q += vd_align(rays * (p->hidden / 2) * 4) / 4; }
}

int vd_check(const GnrProblem* p) {
    if ((p->vd_dims - 3) % 6 != 0 || p->vd_dims < 3 || p->vd_dims > VD_MAX)
        return fail("gnr: vd_dims = %d: the device-side view-direction encoder needs 3 + 6 n_freqs with n_freqs <= 8 "
                    "(the reference: 27); pass ray_bias to fold another encoding yourself", p->vd_dims);
    return 0;
}

void launch_vd_fwd(const GnrProblem& p, int n_streams, const GnrWeights* const* w, float* embed, float* const* rb, hipStream_t st) {
    VdParams vp{};
    vp.prob = p; vp.n_streams = n_streams; vp.ld = p.hidden + p.vd_dims + p.appea_dims; vp.embed = embed;
    for (int s = 0; s < n_streams; ++s) { vp.w1[s] = w[s]->rgb_w[1]; vp.ray_bias[s] = rb[s]; }
    hipLaunchKernelGGL(vd_fwd_kernel, dim3((unsigned)((p.n_rays + VD_RPB - 1) / VD_RPB), (unsigned)p.batch), dim3(256), 0, st, vp);
}

// scratch floats of the backward: d_rb per weight set + partial sums + dR_extra
size_t vd_bwd_floats(const GnrProblem* p, int n_streams) {
    if (!vd_on_device(p)) return 0;
    const size_t rays = (size_t)p->batch * p->n_rays;
    const size_t bpi = (p->n_rays + VD_RPB - 1) / VD_RPB, segs = (rays + VD_SEG_RAYS - 1) / VD_SEG_RAYS;
    return (size_t)n_streams * (vd_align(rays * (p->hidden / 2) * 4) / 4) + vd_align((size_t)p->batch * bpi * 9 * 4) / 4 +
           vd_align((size_t)n_streams * segs * p->vd_dims * (p->hidden / 2) * 4) / 4 + vd_align((size_t)p->batch * 9 * 4) / 4;
}
struct VdBwdScratch { float* d_rb[2]; float* dR_part; float* dW_part; float* dR_extra; int bpi, segs; };
void vd_carve_bwd(const GnrProblem* p, int n_streams, float* base, VdBwdScratch* sc) {
    const size_t rays = (size_t)p->batch * p->n_rays;
    sc->bpi = (int)((p->n_rays + VD_RPB - 1) / VD_RPB);
    sc->segs = (int)((rays + VD_SEG_RAYS - 1) / VD_SEG_RAYS);
    float* q = base;
    for (int s = 0; s < 2; ++s) sc->d_rb[s] = nullptr;
    for (int s = 0; s < n_streams; ++s) { sc->d_rb[s] = q; q += vd_align(rays * (p->hidden / 2) * 4) / 4; }
    sc->dR_part = q; q += vd_align((size_t)p->batch * sc->bpi * 9 * 4) / 4;
    sc->dW_part = q; q += vd_align((size_t)n_streams * sc->segs * p->vd_dims * (p->hidden / 2) * 4) / 4;
    sc->dR_extra = q;
}

// after every weight set's d_rb has been written: d W1[:, H:H+vd] (written) and dR_extra [B][9] (for geo_final_kernel)
void launch_vd_bwd(const GnrProblem& p, int n_streams, const GnrWeights* const* w, const GnrWeightGrads* const* dw,
                   const float* embed, const VdBwdScratch& sc, bool want_dR, hipStream_t st) {
    VdParams vp{};
    vp.prob = p; vp.n_streams = n_streams; vp.ld = p.hidden + p.vd_dims + p.appea_dims; vp.embed = (float*)embed;
    vp.dR_part = sc.dR_part; vp.dW_part = sc.dW_part; vp.dR_extra = want_dR ? sc.dR_extra : nullptr;
    vp.blocks_per_image = sc.bpi; vp.segs = sc.segs;
    bool any_w = false;
    for (int s = 0; s < n_streams; ++s) {
        vp.w1[s] = w[s]->rgb_w[1]; vp.d_ray_bias[s] = sc.d_rb[s];
        vp.dW1[s] = dw[s] ? dw[s]->rgb_w[1] : nullptr;
        any_w = any_w || vp.dW1[s];
    }
    if (want_dR)
        hipLaunchKernelGGL(vd_bwd_ray_kernel, dim3((unsigned)sc.bpi, (unsigned)p.batch), dim3(256), 0, st, vp);
    if (any_w)
        hipLaunchKernelGGL(vd_bwd_w_kernel, dim3((unsigned)p.vd_dims, (unsigned)sc.segs, (unsigned)n_streams), dim3(256), 0, st, vp);
    const long n = (long)n_streams * p.vd_dims * (p.hidden / 2) + (long)p.batch * 9;
    hipLaunchKernelGGL(vd_bwd_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, vp);
}

}  // namespace gnr
