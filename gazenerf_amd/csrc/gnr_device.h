// gnr_device.h -- device helpers shared by the forward and backward kernels (gfx950).
#pragma once
#include "gnr_internal.h"

namespace gnr {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    // v_mfma_f32_32x32x2_f32: D[i][j] += sum_k A[i][k] B[k][j]; lane l holds A[i=l&31][k=l>>5],
    // B[k=l>>5][j=l&31]; D reg r: row i = (r&3) + 8(r>>2) + 4(l>>5), col j = l&31.  Exact fp32 fma chain.
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Ray geometry: GenSamplePoints.forward, utils/model_utils.py:364-372.
// ---------------------------------------------------------------------------------------------
struct Ray {
    float ox, oy, oz;     // ray origin = camera centre T
    float dx, dy, dz;     // unit direction
    float l;              // -1/dz: one unit of z-val == one unit of world z (model_utils.py:369)
    float ux, uy, uz;     // un-normalised direction R Kinv [x y 1]   (kept for backward)
    float inv_n;          // 1/|u|
};

__device__ __forceinline__ Ray make_ray(const GnrProblem& p, int b, int ray) {
    const float x = p.xy[((long)b * 2 + 0) * p.n_rays + ray];
    const float y = p.xy[((long)b * 2 + 1) * p.n_rays + ray];
    const float* K = p.Kinv + b * 9;
    const float* R = p.R + b * 9;
    const float* T = p.T + b * 3;
    // bmm as an fma chain over k = 0,1,2 (F.pad appends the 1.0 row, model_utils.py:365)
    const float v0 = fmaf(K[2], 1.0f, fmaf(K[1], y, K[0] * x));
    const float v1 = fmaf(K[5], 1.0f, fmaf(K[4], y, K[3] * x));
    const float v2 = fmaf(K[8], 1.0f, fmaf(K[7], y, K[6] * x));
    Ray r;
    r.ux = fmaf(R[2], v2, fmaf(R[1], v1, R[0] * v0));
    r.uy = fmaf(R[5], v2, fmaf(R[4], v1, R[3] * v0));
    r.uz = fmaf(R[8], v2, fmaf(R[7], v1, R[6] * v0));
    const float n = sqrtf(fmaf(r.uz, r.uz, fmaf(r.uy, r.uy, r.ux * r.ux)));
    r.inv_n = 1.0f / n;
    r.dx = r.ux / n;
    r.dy = r.uy / n;
    r.dz = r.uz / n;
    r.l = -1.0f / r.dz;
    r.ox = T[0];
    r.oy = T[1];
    r.oz = T[2];
    return r;
}

// Edge i (0..N_p) of the plane sweep, utils/model_utils.py:339-357.  Un-fused mul/add so the
// rounding sequence equals the reference's elementwise ops.
__device__ __forceinline__ float sweep_edge(const GnrProblem& p, float oz, int i) {
    const int steps = p.n_samples + 1;
    const float step = 1.0f / (float)(steps - 1);
    // torch.linspace: symmetric fill from both ends
    const float t = (i < steps / 2) ? __fmul_rn(step, (float)i)
                                    : __fsub_rn(1.0f, __fmul_rn(step, (float)(steps - 1 - i)));
    const float rz1 = __fsub_rn(oz, p.world_z1);
    const float rz2 = __fsub_rn(oz, p.world_z2);
    return __fadd_rn(__fmul_rn(rz1, __fsub_rn(1.0f, t)), __fmul_rn(rz2, t));
}

// Edge i after the optional stratified jitter (model_utils.py:302-307) or explicit edges.
__device__ __forceinline__ float sample_edge(const GnrProblem& p, float oz, long ray_g, int i) {
    const int np = p.n_samples;
    if (p.z_edges) return p.z_edges[ray_g * (np + 1) + i];
    const float z = sweep_edge(p, oz, i);
    if (!p.t_rand) return z;
    const float zl = i > 0 ? sweep_edge(p, oz, i - 1) : z;
    const float zu = i < np ? sweep_edge(p, oz, i + 1) : z;
    const float lower = i > 0 ? __fmul_rn(0.5f, __fadd_rn(z, zl)) : z;
    const float upper = i < np ? __fmul_rn(0.5f, __fadd_rn(zu, z)) : z;
    const float tr = p.t_rand[ray_g * (np + 1) + i];
    return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), tr));
}

// Positional encoding of one point for lane-half h: the 32 k-step values of enc_channel().
// Embedder.forward, utils/model_utils.py:272-280 (freq = 2^f exactly; accurate sin/cos: the
// arguments reach ~1.7e3 rad).
__device__ __forceinline__ void encode_point(float px, float py, float pz, int h, float (&e)[ENC_STEPS]) {
    e[0] = h ? py : px;
    e[1] = h ? 0.0f : pz;
    const float pa[3] = {px, py, pz};
#pragma unroll
    for (int fl = 0; fl < 5; ++fl) {
        const float scale = (float)(1 << fl) * (h ? 32.0f : 1.0f);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float s, c;
            sincosf(pa[a] * scale, &s, &c);
            e[2 + 6 * fl + a] = s;
            e[2 + 6 * fl + 3 + a] = c;
        }
    }
}

// sum over the 32 lanes of each wave half (lanes 0-31 / 32-63); result in every lane.
__device__ __forceinline__ float half_sum32(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}

}  // namespace gnr
