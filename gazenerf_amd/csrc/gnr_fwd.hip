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

// per-wave LDS bias table: rows L0..L7, RGB0 (384 each), RGB1 (192), RGB2 (288)
constexpr int BIAS_ROW = H;
constexpr int BIAS_FLOATS = N_CHAIN * BIAS_ROW;                       // 4224 floats = 16.5 KiB per wave
constexpr size_t FWD_LDS_BYTES = (size_t)(ENC_STEPS * 256 + WAVES_PER_WG * BIAS_FLOATS) * sizeof(float);

template <bool SAVE>
__global__ __launch_bounds__(256, 1) void fwd_kernel(const FwdParams fp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* enc_lds = smem;                              // [step][thread]: each thread owns a column
    const ClkProbe clk0 = clk_begin();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    float* bias_lds = smem + ENC_STEPS * 256 + wave * BIAS_FLOATS;     // this wave's table
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

    // the weight stream starts now: its first rows land while the geometry / encoding is computed
    WStream w;
    wstream_init(w, fp.ws[0].packed, lane);

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
                dump_store(fp.enc + chunk * (CHUNK * ENC_PAD) + (2 * s + h) * CHUNK + j, e[s]);
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
        float* acth = ws.act_h;
        // this image's biases (latent codes folded in) -> the wave's LDS table; wave-private, so an
        // LDS wait is all the synchronisation needed
        {
            const long bstride = (long)p.batch * H;
            const float* bsrc = ws.bias + (long)b * H;
            for (int q = lane; q < N_CHAIN * (H / 4); q += 64) {
                const int l = q / (H / 4), c4 = q - l * (H / 4);
                *(f32x4*)(bias_lds + l * BIAS_ROW + 4 * c4) = *(const f32x4*)(bsrc + l * bstride + 4 * c4);
            }
            // per-ray bias of RGB_layer_1 (the caller's fold of the view-direction columns, include/gnr.h); LDS
            // operations of one wave execute in order
            if (ws.ray_bias) {
                const int nrb = p.hidden / 2;
                const float* rb = ws.ray_bias + ray_g * nrb;
                for (int c = lane; c < nrb; c += 64) bias_lds[LR1 * BIAS_ROW + c] += rb[c];
            }
        }
        auto bl = [&](int l) { return bias_lds + l * BIAS_ROW; };
        auto dp = [&](float* dst, int C) -> float* { return SAVE ? dump_ptr(dst, C, chunk, j, h) : nullptr; };
        auto sb = [&](int layer) { return ws.relu_bits + relu_bits_offset(layer, fp.n_chunks, chunk); };
        // Every layer's output is dumped (training forward) by the NEXT layer's mm_h, spread over its
        // MFMA loop; only the sign bits are written at the layer boundary.

        // Layer epilogue (bias + activation, sign bits in training) runs per output tile inside the
        // tail of the producing mm_h.  mkw collects the ReLU bit words of the layer.
        unsigned mkw[RELU_WORDS];
#define GNR_EPI(X, L, RELU)                                                                              \
    [&](int t) {                                                                                         \
        const unsigned bt = bias_act_tile<RELU>(X[t], bl(L) + 32 * t, h);                                \
        if (SAVE) mkw[t >> 1] = (t & 1) ? (mkw[t >> 1] | (bt << 16)) : bt;                               \
    }
        auto put_bits = [&](int layer, int words) {
            if (SAVE) {
                unsigned* dst = sb(layer);
#pragma unroll
                for (int q = 0; q < RELU_WORDS; ++q)
                    if (q < words) dump_store(dst + q * 64 + lane, mkw[q]);
            }
        };

        // L0: enc -> A
        mm_enc<NT_H, SAVE>(enc_col, A, w);
        {
            auto epi = GNR_EPI(A, 0, true);
#pragma unroll
            for (int t = 0; t < NT_H; ++t) epi(t);
        }
        put_bits(0, 6);

        // L1..L4: A -> Bv -> A -> Bv -> A   (each mm_h dumps its input h_{l-1})
#pragma unroll 1
        for (int rep = 0; rep < 2; ++rep) {
            const int la = 1 + 2 * rep, lb = 2 + 2 * rep;
            mm_h<NT_H, NT_H, true, SAVE>(A, Bv, w, dp(acth + (la - 1) * fp.M * H, H), GNR_EPI(Bv, la, true));
            put_bits(la, 6);
            mm_h<NT_H, NT_H, true, SAVE>(Bv, A, w, dp(acth + (lb - 1) * fp.M * H, H), GNR_EPI(A, lb, true));
            put_bits(lb, 6);
        }

        // L5: [enc | A] -> Bv   (skip connection, models/mlp_nerf.py:107); dumps h4
        mm_enc<NT_H, SAVE>(enc_col, Bv, w);
        mm_h<NT_H, NT_H, false, SAVE>(A, Bv, w, dp(acth + 4 * fp.M * H, H), GNR_EPI(Bv, 5, true));
        put_bits(5, 6);

        // L6: Bv -> A (dumps h5), L7: A -> Bv (dumps h6)
        mm_h<NT_H, NT_H, true, SAVE>(Bv, A, w, dp(acth + 5 * fp.M * H, H), GNR_EPI(A, 6, true));
        put_bits(6, 6);
        mm_h<NT_H, NT_H, true, SAVE>(A, Bv, w, dp(acth + 6 * fp.M * H, H), GNR_EPI(Bv, 7, true));
        put_bits(7, 6);

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
        mm_h<NT_H, NT_H, true, SAVE>(Bv, A, w, dp(acth + 7 * fp.M * H, H), GNR_EPI(A, LR0, false));
        // RGB1: A -> Bv[0..6) (+ folded appearance code), ReLU; dumps y0
        mm_h<NT_H, NT_H2, true, SAVE>(A, Bv, w, dp(ws.act_y0, H), GNR_EPI(Bv, LR1, true));
        put_bits(8, 3);
        // RGB2: Bv[0..6) -> A[0..9)  (258 channels padded to 288; no sigmoid, mlp_nerf.py:116); dumps y1
        mm_h<NT_H2, NT_F, true, SAVE>(Bv, A, w, dp(ws.act_y1, H2), GNR_EPI(A, LR2, false));
#undef GNR_EPI
        if (SAVE) dump<NT_F>(A, ws.act_feat, FEAT_PAD, chunk, j, h);

        // ---- A5: chunk-local compositing (utils/model_utils.py:498-534) ----
        composite_chunk(A, sig, delta, z0, ws, chunk, row, lane, SAVE || fp.want_wl);
    }
    clk_end(clk0, fp.clk);
}

void launch_fwd(const FwdParams& fp, hipStream_t stream) {
    const unsigned grid = (unsigned)((fp.n_chunks + WAVES_PER_WG - 1) / WAVES_PER_WG);
    // > 64 KiB of dynamic LDS needs an opt-in per device: set it on every launch (cheap, and correct for
    // several devices / threads per process -- a process-wide 'done' flag would not be)
    (void)hipFuncSetAttribute((const void*)(fp.save ? fwd_kernel<true> : fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FWD_LDS_BYTES);
    if (fp.save)
        hipLaunchKernelGGL(fwd_kernel<true>, dim3(grid), dim3(256), FWD_LDS_BYTES, stream, fp);
    else
        hipLaunchKernelGGL(fwd_kernel<false>, dim3(grid), dim3(256), FWD_LDS_BYTES, stream, fp);
}

}  // namespace gnr
