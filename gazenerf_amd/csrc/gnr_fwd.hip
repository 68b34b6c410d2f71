// gnr_fwd.hip -- fused march -> encode -> two-stream MLP -> chunk-local composite (gfx950).
//
// Replaces, per 32-sample chunk of a ray and entirely in registers:
//   GenSamplePoints  utils/model_utils.py:283-375     Embedder   utils/model_utils.py:240-280
//   MLPforNeRF       models/mlp_nerf.py:95-119        CalcRayColor utils/model_utils.py:493-534
//
// One wavefront = one chunk = the 32 columns (samples) of v_mfma_f32_32x32x2_f32 tiles.  The whole
// 11-layer chain runs transposed, Y^T = W X^T: weights are the streamed A operand, activations the
// B operand.  Because the C/D register layout of one layer IS the B layout of the next (with the
// k-order the packer bakes into the weights), activations never leave the register file:
//   192 regs (12 tiles x 16) current activations + 192 regs accumulators, 1 wave per SIMD.
// Only the pre-packed weights stream (L2-resident, one coalesced 1 KiB float4 row per 4 MFMAs) and
// the chunk's composited partial (288 floats + 3 scalars) is written.  No LDS traffic for the GEMMs,
// no barriers: the four waves of a workgroup are independent.
#include "gnr_chain.h"

namespace gnr {

template <bool SAVE>
__global__ __launch_bounds__(256, 1) void fwd_kernel(const FwdParams fp) {
    __shared__ float enc_lds[ENC_STEPS * 256];      // [step][thread]: each thread owns a column

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long chunk = (long)blockIdx.x * WAVES_PER_WG + wave;
    if (chunk >= fp.n_chunks) return;               // wave-uniform; no barriers below
    const GnrProblem& p = fp.prob;
    const int cpr = fp.chunks_per_ray;
    const long ray_g = chunk / cpr;
    const int c_in = (int)(chunk - ray_g * cpr);
    const int b = (int)(ray_g / p.n_rays);
    const int ray = (int)(ray_g - (long)b * p.n_rays);
    const int i = c_in * CHUNK + j;
    const bool valid = i < p.n_samples;
    const long row = chunk * CHUNK + j;             // padded global sample index

    // ---- A1: ray + sample ----
    const Ray r = make_ray(p, b, ray);
    const int ic = valid ? i : p.n_samples - 1;
    const float z0 = sample_edge(p, r.oz, ray_g, ic);
    const float z1 = sample_edge(p, r.oz, ray_g, ic + 1);
    const float delta = valid ? __fmul_rn(__fsub_rn(z1, z0), r.l) : 0.0f;
    const float px = __fadd_rn(r.ox, __fmul_rn(__fmul_rn(r.dx, r.l), z0));
    const float py = __fadd_rn(r.oy, __fmul_rn(__fmul_rn(r.dy, r.l), z0));
    const float pz = __fadd_rn(r.oz, __fmul_rn(__fmul_rn(r.dz, r.l), z0));

    // ---- A2: positional encoding -> LDS column (re-used by L0 and L5 of both streams) ----
    float* enc_col = enc_lds + tid;
    {
        float e[ENC_STEPS];
        encode_point(px, py, pz, h, e);
#pragma unroll
        for (int s = 0; s < ENC_STEPS; ++s) enc_col[s * 256] = e[s];
        if (SAVE) {
            // CCM [chunk][64][32] in our k-order: channel slot 2*step + h
#pragma unroll
            for (int s = 0; s < ENC_STEPS; ++s)
                fp.enc[chunk * (CHUNK * ENC_PAD) + (2 * s + h) * CHUNK + j] = e[s];
            if (h == 0) {
                fp.delta[row] = delta;
                fp.zval[row] = z0;
                *(f32x4*)(fp.pts + row * 4) = f32x4{px, py, pz, 0.0f};
            }
        } else if (fp.want_wl && h == 0) {
            fp.zval[row] = z0;
        }
    }

    f32x16 A[NT_H], Bv[NT_H];

#pragma unroll 1
    for (int s = 0; s < fp.n_streams; ++s) {
        const StreamWs& ws = fp.ws[s];
        const float* bias_b = ws.bias + (long)b * H;                 // + layer * B * H
        const long bstride = (long)p.batch * H;
        const f32x4* Pk = (const f32x4*)ws.packed;
        auto Pl = [&](int l) { return Pk + packed_offset(l) / 4; };
        float* acth = ws.act_h;

        auto dp = [&](float* dst, int C) -> float* { return SAVE ? dump_ptr(dst, C, chunk, j, h) : nullptr; };
        auto sb = [&](int layer) { return ws.relu_bits + relu_bits_offset(layer, fp.n_chunks, chunk); };
        // Every layer's output is dumped (training forward) by the NEXT layer's mm_h, spread over its
        // MFMA loop; only the sign bits are written at the layer boundary.

        // L0: enc -> A
        init_bias<NT_H>(A, bias_b + 0 * bstride, h);
        mm_enc<NT_H>(enc_col, A, Pl(0), lane);
        relu<NT_H>(A);
        if (SAVE) store_relu_bits<NT_H>(A, sb(0), lane);

        // L1..L4: A -> Bv -> A -> Bv -> A   (each mm_h dumps its input h_{l-1})
#pragma unroll 1
        for (int rep = 0; rep < 2; ++rep) {
            const int la = 1 + 2 * rep, lb = 2 + 2 * rep;
            init_bias<NT_H>(Bv, bias_b + la * bstride, h);
            mm_h<NT_H, NT_H, SAVE>(A, Bv, Pk + (packed_offset(1) + (size_t)(la - 1) * layer_packed_floats(1)) / 4, lane,
                                   dp(acth + (la - 1) * fp.M * H, H));
            relu<NT_H>(Bv);
            if (SAVE) store_relu_bits<NT_H>(Bv, sb(la), lane);
            init_bias<NT_H>(A, bias_b + lb * bstride, h);
            mm_h<NT_H, NT_H, SAVE>(Bv, A, Pk + (packed_offset(1) + (size_t)(lb - 1) * layer_packed_floats(1)) / 4, lane,
                                   dp(acth + (lb - 1) * fp.M * H, H));
            relu<NT_H>(A);
            if (SAVE) store_relu_bits<NT_H>(A, sb(lb), lane);
        }

        // L5: [enc | A] -> Bv   (skip connection, models/mlp_nerf.py:107); dumps h4
        init_bias<NT_H>(Bv, bias_b + 5 * bstride, h);
        mm_enc<NT_H>(enc_col, Bv, Pl(5), lane);
        mm_h<NT_H, NT_H, SAVE>(A, Bv, Pl(5) + (size_t)ENC_STEPS * NT_H * 64 / 4, lane, dp(acth + 4 * fp.M * H, H));
        relu<NT_H>(Bv);
        if (SAVE) store_relu_bits<NT_H>(Bv, sb(5), lane);

        // L6: Bv -> A (dumps h5), L7: A -> Bv (dumps h6)
        init_bias<NT_H>(A, bias_b + 6 * bstride, h);
        mm_h<NT_H, NT_H, SAVE>(Bv, A, Pl(6), lane, dp(acth + 5 * fp.M * H, H));
        relu<NT_H>(A);
        if (SAVE) store_relu_bits<NT_H>(A, sb(6), lane);
        init_bias<NT_H>(Bv, bias_b + 7 * bstride, h);
        mm_h<NT_H, NT_H, SAVE>(A, Bv, Pl(7), lane, dp(acth + 6 * fp.M * H, H));
        relu<NT_H>(Bv);
        if (SAVE) store_relu_bits<NT_H>(Bv, sb(7), lane);

        // density head on h7 (models/mlp_nerf.py:109): 384-long dot, split over the two lane halves
        float sig = 0.0f;
#pragma unroll
        for (int t = 0; t < NT_H; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 w4 = *(const f32x4*)(ws.wsig + 32 * t + 8 * rq + 4 * h);
                sig = fmaf(w4.x, Bv[t][4 * rq + 0], sig);
                sig = fmaf(w4.y, Bv[t][4 * rq + 1], sig);
                sig = fmaf(w4.z, Bv[t][4 * rq + 2], sig);
                sig = fmaf(w4.w, Bv[t][4 * rq + 3], sig);
            }
        sig += __shfl_xor(sig, 32);
        sig += ws.wsig[H];
        if (SAVE && h == 0) ws.sigma_raw[row] = sig;

        // RGB0: Bv -> A (no activation, mlp_nerf.py:110); dumps h7
        init_bias<NT_H>(A, bias_b + LR0 * bstride, h);
        mm_h<NT_H, NT_H, SAVE>(Bv, A, Pl(LR0), lane, dp(acth + 7 * fp.M * H, H));
        // RGB1: A -> Bv[0..6) (+ folded appearance code), ReLU; dumps y0
        init_bias<NT_H2>(Bv, bias_b + LR1 * bstride, h);
        mm_h<NT_H, NT_H2, SAVE>(A, Bv, Pl(LR1), lane, dp(ws.act_y0, H));
        relu<NT_H2>(Bv);
        if (SAVE) store_relu_bits<NT_H2>(Bv, sb(8), lane);
        // RGB2: Bv[0..6) -> A[0..9)  (258 channels padded to 288; no sigmoid, mlp_nerf.py:116); dumps y1
        init_bias<NT_F>(A, bias_b + LR2 * bstride, h);
        mm_h<NT_H2, NT_F, SAVE>(Bv, A, Pl(LR2), lane, dp(ws.act_y1, H2));
        if (SAVE) dump<NT_F>(A, ws.act_feat, FEAT_PAD, chunk, j, h);

        // ---- A5: chunk-local compositing (utils/model_utils.py:498-534) ----
        const float sigma = fmaxf(sig, 0.0f);
        const float alpha = 1.0f - expf(-sigma * delta);
        const float x = (1.0f - alpha) + 1e-10f;
        // inclusive prefix product over the 32 samples of the chunk
        float incl = x;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const float o = __shfl_up(incl, d, 32);
            if (j >= d) incl *= o;
        }
        float excl = __shfl_up(incl, 1, 32);
        if (j == 0) excl = 1.0f;
        const float wl = alpha * excl;
        const float ptot = __shfl(incl, 31, 32);
        const float accw = half_sum32(wl);
        const float dsum = half_sum32(wl * z0);
        if (lane == 0) *(f32x4*)(ws.part_sc + chunk * 4) = f32x4{ptot, accw, dsum, 0.0f};
        if ((SAVE || fp.want_wl) && h == 0) ws.wl[row] = wl;

        float* pf = ws.part_feat + chunk * FEAT_PAD + 4 * h;
#pragma unroll
        for (int t = 0; t < NT_F; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 v;
                v.x = half_sum32(wl * A[t][4 * rq + 0]);
                v.y = half_sum32(wl * A[t][4 * rq + 1]);
                v.z = half_sum32(wl * A[t][4 * rq + 2]);
                v.w = half_sum32(wl * A[t][4 * rq + 3]);
                if (j == 0) *(f32x4*)(pf + 32 * t + 8 * rq) = v;
            }
    }
}

void launch_fwd(const FwdParams& fp, hipStream_t stream) {
    const unsigned grid = (unsigned)((fp.n_chunks + WAVES_PER_WG - 1) / WAVES_PER_WG);
    if (fp.save)
        hipLaunchKernelGGL(fwd_kernel<true>, dim3(grid), dim3(256), 0, stream, fp);
    else
        hipLaunchKernelGGL(fwd_kernel<false>, dim3(grid), dim3(256), 0, stream, fp);
}

}  // namespace gnr
