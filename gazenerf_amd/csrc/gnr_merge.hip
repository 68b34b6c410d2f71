// gnr_merge.hip -- feature-map merge after the volumetric hot path, gfx950.
//
// Replaces models/gaze_nerf.py:175-203 + utils/model_utils.py:11-46 (rotate / rotation_matrix_2d):
//   merge_face = feat_face + bg_alpha_face * bg_featmap
//   merge_eyes = feat_eyes + bg_alpha_eyes * bg_featmap
//   eyes_planes[3g+c'] = sum_c merge_eyes[3g+c] * Rot[c][c'],  Rot = M2(yaw) M1(pitch)  (per image)
//   merge = max(merge_face, eyes_planes)
// All maps are the reference's channels-first [B, C, n_pix] (C = 3*86), so the two outputs of the
// render op feed it without a layout change.  HBM-bound elementwise work: one thread per pixel walks
// the images and channel triplets; every access is coalesced across the pixels of a wave.
// Algorithmic bytes per pixel per image: (2*C + 2) read + C (bg, shared) + 3*C written = 6.2 KB.
// Backward recomputes the forward quantities; d(gaze) goes through the 9 entries of Rot with a
// fixed-order two-stage reduction (deterministic); torch.maximum's tie rule (half each) is kept.
#include "gnr_internal.h"

namespace gnr {
int fail(const char* fmt, ...);

struct MergeParams {
    GnrMergeProblem p;
    float* merge_face; float* eyes_planes; float* merge;
};

struct Rot3 { float m[3][3]; };

__device__ __forceinline__ Rot3 make_rot(const float* gaze, int b) {
    const float c0 = cosf(gaze[2 * b]), s0 = sinf(gaze[2 * b]), c1 = cosf(gaze[2 * b + 1]), s1 = sinf(gaze[2 * b + 1]);
    Rot3 r;
    r.m[0][0] = c1;   r.m[0][1] = s1 * s0; r.m[0][2] = s1 * c0;
    r.m[1][0] = 0.0f; r.m[1][1] = c0;      r.m[1][2] = -s0;
    r.m[2][0] = -s1;  r.m[2][1] = c1 * s0; r.m[2][2] = c1 * c0;
    return r;
}

__global__ __launch_bounds__(256) void merge_fwd_kernel(const MergeParams mp) {
    const GnrMergeProblem& p = mp.p;
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= p.n_pix) return;
    const int C = p.feat_nc, G = C / 3;
    // grid.y = image, grid.z = slice of the channel triplets: a 64 x 64 map is only 16 blocks of pixels, and one thread
    // walking all images and 86 triplets took 124 us (1-3 % of a 64 x 64-ray inference)
    const int g0 = (int)((long)G * blockIdx.z / gridDim.z), g1 = (int)((long)G * (blockIdx.z + 1) / gridDim.z);
    {
        const int b = blockIdx.y;
        const Rot3 R = make_rot(p.gaze, b);
        const float af = p.bg_alpha_face[(long)b * p.n_pix + pix], ae = p.bg_alpha_eyes[(long)b * p.n_pix + pix];
        for (int g = g0; g < g1; ++g) {
            float mf[3], me[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const long o = ((long)b * C + 3 * g + c) * p.n_pix + pix;
                const float bg = p.bg_featmap[(long)(3 * g + c) * p.n_pix + pix];
                mf[c] = p.feat_face[o] + af * bg;
                me[c] = p.feat_eyes[o] + ae * bg;
            }
#pragma unroll
            for (int c2 = 0; c2 < 3; ++c2) {
                const long o = ((long)b * C + 3 * g + c2) * p.n_pix + pix;
                const float ep = me[0] * R.m[0][c2] + me[1] * R.m[1][c2] + me[2] * R.m[2][c2];
                if (mp.merge_face) mp.merge_face[o] = mf[c2];
                if (mp.eyes_planes) mp.eyes_planes[o] = ep;
                if (mp.merge) mp.merge[o] = fmaxf(mf[c2], ep);
            }
        }
    }
}

struct MergeBwdParams {
    GnrMergeProblem p;
    const float* g_mf; const float* g_ep; const float* g_m;
    float* d_ff; float* d_af; float* d_fe; float* d_ae; float* d_bg;
    float* rot_part;          // [B][blocks * slices][9]
    float* af_part;           // [slices][B][n_pix] partial d(bg_alpha_face) per channel slice (slices > 1), then the eyes'
    float* ae_part;
    int blocks;
};

// channel-triplet slices of the backward grid: a 64 x 64 map is 16 blocks of pixels on 256 CUs, and one thread walking
// all images and 86 triplets took 228 us
static inline int merge_slices(const GnrMergeProblem* p) {
    const long blocks = (p->n_pix + 255) / 256;
    long z = 256 / blocks;
    const int G = p->feat_nc / 3;
    if (z > 16) z = 16;
    if (z > G) z = G;
    return z < 1 ? 1 : (int)z;
}

__global__ __launch_bounds__(256) void merge_bwd_kernel(const MergeBwdParams mp) {
    __shared__ float red[9][256];
    const GnrMergeProblem& p = mp.p;
    const int tid = threadIdx.x;
    const long pix = (long)blockIdx.x * 256 + tid;
    const bool live = pix < p.n_pix;
    const int C = p.feat_nc, G = C / 3;
    const int zs = gridDim.y, z = blockIdx.y;
    const int gbeg = (int)((long)G * z / zs), gend = (int)((long)G * (z + 1) / zs);
    for (int b = 0; b < p.batch; ++b) {
        float dR[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) dR[k] = 0.0f;
        if (live) {
            const Rot3 R = make_rot(p.gaze, b);
            const float af = p.bg_alpha_face[(long)b * p.n_pix + pix], ae = p.bg_alpha_eyes[(long)b * p.n_pix + pix];
            float daf = 0.0f, dae = 0.0f;
            for (int g = gbeg; g < gend; ++g) {
                float mf[3], me[3], bg[3], ep[3], Gmf[3], Gep[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const long o = ((long)b * C + 3 * g + c) * p.n_pix + pix;
                    bg[c] = p.bg_featmap[(long)(3 * g + c) * p.n_pix + pix];
                    mf[c] = p.feat_face[o] + af * bg[c];
                    me[c] = p.feat_eyes[o] + ae * bg[c];
                }
#pragma unroll
                for (int c2 = 0; c2 < 3; ++c2) {
                    const long o = ((long)b * C + 3 * g + c2) * p.n_pix + pix;
                    ep[c2] = me[0] * R.m[0][c2] + me[1] * R.m[1][c2] + me[2] * R.m[2][c2];
                    const float gm = mp.g_m ? mp.g_m[o] : 0.0f;
                    // torch.maximum backward: the larger input takes the gradient, ties split it
                    const float wf = mf[c2] > ep[c2] ? 1.0f : (mf[c2] == ep[c2] ? 0.5f : 0.0f);
                    Gmf[c2] = (mp.g_mf ? mp.g_mf[o] : 0.0f) + gm * wf;
                    Gep[c2] = (mp.g_ep ? mp.g_ep[o] : 0.0f) + gm * (1.0f - wf);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const long o = ((long)b * C + 3 * g + c) * p.n_pix + pix;
                    const float Gme = Gep[0] * R.m[c][0] + Gep[1] * R.m[c][1] + Gep[2] * R.m[c][2];
                    if (mp.d_ff) mp.d_ff[o] = Gmf[c];
                    if (mp.d_fe) mp.d_fe[o] = Gme;
                    daf = fmaf(Gmf[c], bg[c], daf);
                    dae = fmaf(Gme, bg[c], dae);
                    if (mp.d_bg) {
                        float* d = mp.d_bg + (long)(3 * g + c) * p.n_pix + pix;      // this thread owns it
                        const float v = af * Gmf[c] + ae * Gme;
                        *d = b == 0 ? v : *d + v;
                    }
#pragma unroll
                    for (int c2 = 0; c2 < 3; ++c2) dR[3 * c + c2] = fmaf(me[c], Gep[c2], dR[3 * c + c2]);
                }
            }
            if (zs == 1) {
                if (mp.d_af) mp.d_af[(long)b * p.n_pix + pix] = daf;
                if (mp.d_ae) mp.d_ae[(long)b * p.n_pix + pix] = dae;
            } else {                         // merge_alpha_kernel adds the slices in order
                mp.af_part[((long)z * p.batch + b) * p.n_pix + pix] = daf;
                mp.ae_part[((long)z * p.batch + b) * p.n_pix + pix] = dae;
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) red[k][tid] = dR[k];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s)
#pragma unroll
                for (int k = 0; k < 9; ++k) red[k][tid] += red[k][tid + s];
            __syncthreads();
        }
        if (tid < 9) mp.rot_part[((long)b * mp.blocks * zs + (long)blockIdx.x * zs + z) * 9 + tid] = red[tid][0];
        __syncthreads();
    }
}

// d(gaze) from dRot: Rot = [[c1, s1 s0, s1 c0], [0, c0, -s0], [-s1, c1 s0, c1 c0]]
// d(bg_alpha)[b][pix] = sum over the channel slices, in slice order
__global__ void merge_alpha_kernel(const float* __restrict__ part, int slices, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.0f;
    for (int zz = 0; zz < slices; ++zz) a += part[(long)zz * n + i];       // <= 16 independent loads, unrolled by hipcc
    out[i] = a;
}

__global__ void merge_gaze_kernel(const float* rot_part, int blocks, const float* gaze, float* d_gaze) {
    __shared__ float dsum[9];
    const int b = blockIdx.x;
    if (threadIdx.x < 9) {          // one thread per matrix entry, loads 16 at a time, additions in block order
        const float* src = rot_part + (long)b * blocks * 9 + threadIdx.x;
        float a = 0.0f;
        int i = 0;
        for (; i + 16 <= blocks; i += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[(long)(i + u) * 9];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += v[u];
        }
        for (; i < blocks; ++i) a += src[(long)i * 9];
        dsum[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float d[9];
    for (int k = 0; k < 9; ++k) d[k] = dsum[k];
    const float c0 = cosf(gaze[2 * b]), s0 = sinf(gaze[2 * b]), c1 = cosf(gaze[2 * b + 1]), s1 = sinf(gaze[2 * b + 1]);
    // pitch (index 0): d/d0 of s0 = c0, of c0 = -s0
    const float g0 = d[1] * (s1 * c0) + d[2] * (-s1 * s0) + d[4] * (-s0) + d[5] * (-c0) + d[7] * (c1 * c0) + d[8] * (-c1 * s0);
    // yaw (index 1): d/d1 of s1 = c1, of c1 = -s1
    const float g1 = d[0] * (-s1) + d[1] * (c1 * s0) + d[2] * (c1 * c0) + d[6] * (-c1) + d[7] * (-s1 * s0) + d[8] * (-s1 * c0);
    d_gaze[2 * b] = g0;
    d_gaze[2 * b + 1] = g1;
}

static int check_merge(const GnrMergeProblem* p) {
    if (!p) return fail("gnr_merge: problem is NULL");
    if (p->struct_size != sizeof(GnrMergeProblem))
        return fail("gnr_merge: GnrMergeProblem.struct_size is %u but this libgnr.so (ABI %d) has sizeof = %zu: the caller was "
                    "built against a different include/gnr.h (or did not set struct_size)", p->struct_size, GNR_ABI_VERSION,
                    sizeof(GnrMergeProblem));
    if (p->batch < 1 || p->n_pix < 1) return fail("gnr_merge: empty problem");
    if (p->feat_nc < 3 || p->feat_nc % 3) return fail("gnr_merge: feat_nc must be a multiple of 3 (got %d)", p->feat_nc);
    if (!p->feat_face || !p->bg_alpha_face || !p->feat_eyes || !p->bg_alpha_eyes || !p->bg_featmap || !p->gaze)
        return fail("gnr_merge: NULL input pointer");
    return 0;
}

}  // namespace gnr

using namespace gnr;

extern "C" {

size_t gnr_merge_scratch_bytes(const GnrMergeProblem* p) {
    if (check_merge(p)) return 0;
    const size_t blocks = ((size_t)p->n_pix + 255) / 256, zs = (size_t)merge_slices(p);
    const size_t rot = ((size_t)p->batch * blocks * zs * 9 * sizeof(float) + 255) & ~(size_t)255;
    const size_t alpha = zs > 1 ? 2 * (((size_t)zs * p->batch * p->n_pix * sizeof(float) + 255) & ~(size_t)255) : 0;
    return rot + alpha;
}

int gnr_merge_fwd(const GnrMergeProblem* p, float* merge_face, float* eyes_planes, float* merge, void* stream) {
    if (check_merge(p)) return 1;
    if (!merge_face && !eyes_planes && !merge) return fail("gnr_merge_fwd: no output requested");
    MergeParams mp{*p, merge_face, eyes_planes, merge};
    const unsigned pb = (unsigned)((p->n_pix + 255) / 256);
    const unsigned gz = pb * p->batch >= 1024 ? 1u : (pb * p->batch >= 256 ? 4u : 16u);      // enough blocks for 256 CUs
    hipLaunchKernelGGL(merge_fwd_kernel, dim3(pb, (unsigned)p->batch, gz), dim3(256), 0, (hipStream_t)stream, mp);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("gnr_merge_fwd: launch failed: %s", hipGetErrorString(e));
    return 0;
}

int gnr_merge_bwd(const GnrMergeProblem* p, const float* g_merge_face, const float* g_eyes_planes, const float* g_merge,
                  float* d_feat_face, float* d_bg_alpha_face, float* d_feat_eyes, float* d_bg_alpha_eyes,
                  float* d_bg_featmap, float* d_gaze, void* scratch, size_t scratch_bytes, void* stream) {
    if (check_merge(p)) return 1;
    const size_t need = gnr_merge_scratch_bytes(p);
    if (!scratch || scratch_bytes < need) return fail("gnr_merge_bwd: scratch too small (%zu < %zu bytes)", scratch_bytes, need);
    MergeBwdParams mp{};
    mp.p = *p; mp.g_mf = g_merge_face; mp.g_ep = g_eyes_planes; mp.g_m = g_merge;
    mp.d_ff = d_feat_face; mp.d_af = d_bg_alpha_face; mp.d_fe = d_feat_eyes; mp.d_ae = d_bg_alpha_eyes; mp.d_bg = d_bg_featmap;
    mp.rot_part = (float*)scratch;
    mp.blocks = (int)((p->n_pix + 255) / 256);
    const int zs = merge_slices(p);
    const size_t rot = ((size_t)p->batch * mp.blocks * zs * 9 * sizeof(float) + 255) & ~(size_t)255;
    const size_t aslab = ((size_t)zs * p->batch * p->n_pix * sizeof(float) + 255) & ~(size_t)255;
    mp.af_part = (float*)((char*)scratch + rot);
    mp.ae_part = (float*)((char*)scratch + rot + aslab);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(merge_bwd_kernel, dim3(mp.blocks, zs), dim3(256), 0, st, mp);
    if (zs > 1) {
        const long n = (long)p->batch * p->n_pix;
        const unsigned nb = (unsigned)((n + 255) / 256);
        if (d_bg_alpha_face) hipLaunchKernelGGL(merge_alpha_kernel, dim3(nb), dim3(256), 0, st, mp.af_part, zs, n, d_bg_alpha_face);
        if (d_bg_alpha_eyes) hipLaunchKernelGGL(merge_alpha_kernel, dim3(nb), dim3(256), 0, st, mp.ae_part, zs, n, d_bg_alpha_eyes);
    }
    if (d_gaze) hipLaunchKernelGGL(merge_gaze_kernel, dim3(p->batch), dim3(64), 0, st, mp.rot_part, mp.blocks * zs, p->gaze, d_gaze);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("gnr_merge_bwd: launch failed: %s", hipGetErrorString(e));
    return 0;
}

}  // extern "C"
