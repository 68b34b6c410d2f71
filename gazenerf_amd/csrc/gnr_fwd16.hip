// gnr_fwd16.hip -- fused march -> encode -> two-stream MLP -> sub-chunk composite, two waves per SIMD (gfx950).
//
// Replaces, per 16-sample sub-chunk of a ray and entirely in registers:
//   GenSamplePoints  utils/model_utils.py:283-375     Embedder   utils/model_utils.py:240-280
//   MLPforNeRF       models/mlp_nerf.py:95-119        CalcRayColor utils/model_utils.py:493-534
//
// One wavefront = 16 samples = the 16 columns of v_mfma_f32_16x16x4_f32 tiles; the 11-layer chain runs transposed
// (Y^T = W X^T: weights are the streamed A operand) with activations chained through the C/D register layout
// (gnr_chain16.h): 96 registers of current activations + 96 accumulators + 32 of weight rows in flight, 256 per
// wave, TWO waves per SIMD (two 256-thread workgroups per CU).  The four waves of a workgroup are independent: no
// barriers, per-wave buffer loads of the L2-resident packed weights, per-wave LDS bias table.
#include "gnr_chain16.h"

namespace gnr {

// Timing experiments only (results are incomplete with any bit set): -DGNR_ABL16=<bits>, training forward:
//   1 no sign-bit stores   2 no per-layer activation dumps (mm16_h runs its inference loop)   4 no feature dump
//   8 no encoding / geometry dump   16 no sign-bit collection
#ifndef GNR_ABL16
#define GNR_ABL16 0
#endif
constexpr int ABL16 = GNR_ABL16;

// per-wave LDS bias table: rows L0..L7, RGB0 (384 each), RGB1 (192), RGB2 (288)
constexpr int B16_R1 = 9 * H;
constexpr int B16_R2 = B16_R1 + H2;
constexpr int B16_FLOATS = B16_R2 + FEAT_PAD;                          // 3936 floats = 15.4 KiB per wave
__host__ __device__ constexpr int b16_off(int l) { return l <= LR0 ? l * H : (l == LR1 ? B16_R1 : B16_R2); }
constexpr size_t FWD16_LDS_BYTES = (size_t)(ENC16 * 256 + WAVES_PER_WG * B16_FLOATS) * sizeof(float);   // 77.5 KiB
static_assert(2 * FWD16_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

template <bool SAVE>
__global__ __launch_bounds__(256, 2) void fwd16_kernel(const FwdParams fp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* enc_lds = smem;                              // [idx][thread]: each thread owns a column
    const ClkProbe clk0 = clk_begin();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: sub-chunk, ray, image, every base pointer
    float* bias_lds = smem + ENC16 * 256 + wave * B16_FLOATS;          // this wave's table
    const int j = lane & 15, g = lane >> 4;
    const long n_sub = 2 * fp.n_chunks;
    const long sub = (long)blockIdx.x * WAVES_PER_WG + wave;
    if (sub >= n_sub) return;                       // wave-uniform; no barriers below
    const GnrProblem& p = fp.prob;
    const int cpr = fp.chunks_per_ray;
    const long chunk = sub >> 1;
    const int hh = (int)(sub & 1);
    const long ray_g = chunk / cpr;
    const int c_in = (int)(chunk - ray_g * cpr);
    const int b = (int)(ray_g / p.n_rays);
    const int ray = (int)(ray_g - (long)b * p.n_rays);
    const int i = c_in * CHUNK + hh * SUB + j;
    const bool valid = i < p.n_samples;
    const long row = chunk * CHUNK + hh * SUB + j;  // padded global sample index

    // One weight set per workgroup (blockIdx.y): the grid runs all sub-chunks through the first MLP, then through the
    // second, so the chip streams ONE 5.4 MB weight set at a time through its 4 MB L2s instead of two (the dgrad
    // kernel, one set per launch, fetched a third less per set and tolerated the dump stores; this kernel did not).
    // Price: geometry + encoding are computed once per weight set (~0.4 %).
    const int s_lo = blockIdx.y, s_hi = s_lo + 1;
    dephase_first_round(blockIdx.y * gridDim.x + blockIdx.x);
    // the weight stream starts now: its first rows land while the geometry / encoding is computed
    WStream16 w;
    wstream16_init(w, fp.ws[s_lo].packed, lane);

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
        float e[ENC16];
        encode_point16(px, py, pz, g, e);
#pragma unroll
        for (int s = 0; s < ENC16; ++s) enc_col[s * 256] = e[s];
        if (SAVE && !(ABL16 & 8) && s_lo == 0) {
            // CCM [chunk][64 slots][32] in round 1's slot order (the weight-gradient / backward kernels' format)
            float* eb = fp.enc + chunk * (CHUNK * ENC_PAD) + hh * SUB + j;
#pragma unroll
            for (int s = 0; s < ENC16; ++s) dump_store(eb + enc16_slot_rt(s, g) * CHUNK, e[s]);
            if (g == 0) {
                fp.delta[row] = delta;
                fp.zval[row] = z0;
                *(f32x4*)(fp.pts + row * 4) = f32x4{px, py, pz, 0.0f};
            }
        } else if (!SAVE && fp.want_wl && g == 0 && s_lo == 0) {
            fp.zval[row] = z0;
        }
    }

    f32x4 A[NT16_H], Bv[NT16_H];

#pragma unroll 1
    for (int s = s_lo; s < s_hi; ++s) {
        const StreamWs& ws = fp.ws[s];
        float* acth = ws.act_h;
        // this image's biases (latent codes folded in) -> the wave's LDS table; wave-private, so an
        // LDS wait is all the synchronisation needed
        {
            const long bstride = (long)p.batch * H;
            const float* bsrc = ws.bias + (long)b * H;
            // all 16 loads first, then the LDS writes: the rolled load -> write loop paid one global-memory latency
            // per iteration (16 x ~1 us per stream while the SIMD's other wave sat in the same phase)
            constexpr int NQ = (B16_FLOATS / 4 + 63) / 64;
            f32x4 tmp[NQ];
#pragma unroll
            for (int it = 0; it < NQ; ++it) {
                const int f = 4 * (lane + 64 * it);
                const int l = f < B16_R1 ? f / H : (f < B16_R2 ? LR1 : LR2);
                const int c = f - b16_off(l);
                tmp[it] = f < B16_FLOATS ? *(const f32x4*)(bsrc + l * bstride + c) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int it = 0; it < NQ; ++it) {
                const int f = 4 * (lane + 64 * it);
                if (f < B16_FLOATS) *(f32x4*)(bias_lds + f) = tmp[it];
            }
            // per-ray bias of RGB_layer_1 (the caller's fold of the view-direction columns, include/gnr.h); LDS
            // operations of one wave execute in order
            if (ws.ray_bias) {
                const int nrb = p.hidden / 2;
                const float* rb = ws.ray_bias + ray_g * nrb;
                for (int c = lane; c < nrb; c += 64) bias_lds[B16_R1 + c] += rb[c];
            }
        }
        // The dumps' lane offset, made opaque once per weight set: as a loop invariant of this stream loop hipcc hoists
        // "lane offset + immediate" for all 32 immediates into VGPRs that then live across the whole kernel (each store
        // with its own address register instead of the 12-bit offset field) -- the register pressure cost the training
        // forward 2.6 ms per launch.
        unsigned lane_off = dump_lane_off16(j, g);
        asm volatile("" : "+v"(lane_off));
        auto dp = [&](float* dst, int C) { return dump_dst16(SAVE ? dst : nullptr, C, SAVE ? sub : 0, lane_off); };
        auto sb = [&](int layer) {           // sign-bit words [3][64 lanes] of this sub-chunk
            Dump16 d;
            d.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(ws.relu_bits + relu16_offset(layer, n_sub, sub)), 0, 0x7ffffff0, 0x00020000);
            d.voff = (unsigned)lane * 4u;
            return d;
        };
        // Every layer's output is dumped (training forward) by the NEXT layer's mm16_h, spread over its MFMA loop;
        // only the sign bits are written at the layer boundary.  The bias enters through the C operand of the first
        // MFMA of every output tile (read from the LDS table), so the epilogue is the activation alone.
        unsigned mkw[RELU16_WORDS];
#define GNR_BIAS(L) [&](int nt) { return *(const f32x4*)(bias_lds + b16_off(L) + 16 * nt + 4 * g); }
#define GNR_RELU(X)                                                                                      \
    [&](int t) {                                                                                         \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
            /* ReLU as one v_max_i32 on the bit pattern (negative floats are negative integers).  Training:     \
               sign bit = min(relu bits, 1), pushed into the tile group's word -- v_min_u32 (asm: LLVM turns    \
               umin(select, 1) back into compare + select) + v_lshl_or_b32: 3 VALU per value in all */          \
            const float v = X[t][e];                                                                     \
            const int bi = __builtin_bit_cast(int, v);                                                   \
            const int r = bi > 0 ? bi : 0;                                                               \
            X[t][e] = __builtin_bit_cast(float, r);                                                      \
            if (SAVE && !(ABL16 & 16)) {                                                                 \
                unsigned b;                                                                              \
                if ((t & 7) == 0 && e == 0) mkw[t >> 3] = 0;                                             \
                asm("v_min_u32 %0, 1, %2\n\tv_lshl_or_b32 %1, %1, 1, %0"                                 \
                    : "=&v"(b), "+v"(mkw[t >> 3]) : "v"(r));                                             \
            }                                                                                            \
        }                                                                                                \
    }
        auto noneA = [](int) {};
        auto put_bits = [&](int layer, int words) {
            if (SAVE && !(ABL16 & 1)) {
                const Dump16 dst = sb(layer);
                // value 4 (t & 7) + e of a word sits at bit 31 - value: a half-filled word (RGB_layer_1: 12 tiles) is
                // left-aligned here
#pragma unroll
                for (int q = 0; q < RELU16_WORDS; ++q)
                    if (q < words) dump_store16_bytes(dst, (unsigned)q * 256u, (words == 2 && q == 1) ? (mkw[q] << 16) : mkw[q]);
            }
        };

        // L0: enc -> A
        mm16_enc<NT16_H, SAVE>(enc_col, A, w, GNR_BIAS(0));
        {
            auto epi = GNR_RELU(A);
#pragma unroll
            for (int t = 0; t < NT16_H; ++t) epi(t);
        }
        put_bits(0, 3);

        // L1..L4: A -> Bv -> A -> Bv -> A   (each mm16_h dumps its input h_{l-1})
#pragma unroll 1
        for (int rep = 0; rep < 2; ++rep) {
            const int la = 1 + 2 * rep, lb = 2 + 2 * rep;
            mm16_h<NT16_H, NT16_H, true, SAVE && !(ABL16 & 2)>(A, Bv, w, dp(acth + (la - 1) * fp.M * H, H),
                GNR_BIAS(la), GNR_RELU(Bv));
            put_bits(la, 3);
            mm16_h<NT16_H, NT16_H, true, SAVE && !(ABL16 & 2)>(Bv, A, w, dp(acth + (lb - 1) * fp.M * H, H),
                GNR_BIAS(lb), GNR_RELU(A));
            put_bits(lb, 3);
        }

        // L5: [enc | A] -> Bv   (skip connection, models/mlp_nerf.py:107); dumps h4
        mm16_enc<NT16_H, SAVE>(enc_col, Bv, w, GNR_BIAS(5));
        mm16_h<NT16_H, NT16_H, false, SAVE && !(ABL16 & 2)>(A, Bv, w, dp(acth + 4 * fp.M * H, H), ZeroInit16(), GNR_RELU(Bv));
        put_bits(5, 3);

        // L6: Bv -> A (dumps h5), L7: A -> Bv (dumps h6)
        mm16_h<NT16_H, NT16_H, true, SAVE && !(ABL16 & 2)>(Bv, A, w, dp(acth + 5 * fp.M * H, H), GNR_BIAS(6), GNR_RELU(A));
        put_bits(6, 3);
        mm16_h<NT16_H, NT16_H, true, SAVE && !(ABL16 & 2)>(A, Bv, w, dp(acth + 6 * fp.M * H, H), GNR_BIAS(7), GNR_RELU(Bv));
        put_bits(7, 3);

        // density head on h7 (models/mlp_nerf.py:109): 384-long dot, split over the four lane groups
        float sig = 0.0f;
#pragma unroll
        for (int t = 0; t < NT16_H; ++t) {
            const f32x4 w4 = *(const f32x4*)(ws.wsig + 16 * t + 4 * g);
            sig = fmaf(w4.x, Bv[t][0], sig);
            sig = fmaf(w4.y, Bv[t][1], sig);
            sig = fmaf(w4.z, Bv[t][2], sig);
            sig = fmaf(w4.w, Bv[t][3], sig);
        }
        sig += __shfl_xor(sig, 16);
        sig += __shfl_xor(sig, 32);
        sig += ws.wsig[H];
        if (SAVE && g == 0) ws.sigma_raw[row] = sig;

        // RGB0: Bv -> A (no activation, mlp_nerf.py:110); dumps h7
        mm16_h<NT16_H, NT16_H, true, SAVE && !(ABL16 & 2)>(Bv, A, w, dp(acth + 7 * fp.M * H, H), GNR_BIAS(LR0), noneA);
        // RGB1: A -> Bv[0..12) (+ folded appearance code), ReLU; dumps y0
        mm16_h<NT16_H, NT16_H2, true, SAVE && !(ABL16 & 2)>(A, Bv, w, dp(ws.act_y0, H), GNR_BIAS(LR1), GNR_RELU(Bv));
        put_bits(8, 2);
        // RGB2: Bv[0..12) -> A[0..18)  (258 channels padded to 288; no sigmoid, mlp_nerf.py:116); dumps y1
        mm16_h<NT16_H2, NT16_F, true, SAVE && !(ABL16 & 2)>(Bv, A, w, dp(ws.act_y1, H2), GNR_BIAS(LR2), noneA);
#undef GNR_BIAS
#undef GNR_RELU
        if (SAVE && !(ABL16 & 4)) dump16_ccm<NT16_F>(A, dump_dst16_ccm(ws.act_feat, FEAT_PAD, sub, j, g));

        // ---- A5: sub-chunk-local compositing (utils/model_utils.py:498-534) ----
        composite_sub(A, sig, delta, z0, ws, sub, row, lane, SAVE || fp.want_wl);
    }
    clk_end(clk0, fp.clk);
}

void launch_fwd16(const FwdParams& fp, hipStream_t stream) {
    const long n_sub = 2 * fp.n_chunks;
    const unsigned grid = (unsigned)((n_sub + WAVES_PER_WG - 1) / WAVES_PER_WG);
    // > 64 KiB of dynamic LDS needs an opt-in per device: set it on every launch (cheap, and correct for
    // several devices / threads per process -- a process-wide 'done' flag would not be)
    (void)hipFuncSetAttribute((const void*)(fp.save ? fwd16_kernel<true> : fwd16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FWD16_LDS_BYTES);
    const dim3 g3(grid, (unsigned)fp.n_streams);
    if (fp.save)
        hipLaunchKernelGGL(fwd16_kernel<true>, g3, dim3(256), FWD16_LDS_BYTES, stream, fp);
    else
        hipLaunchKernelGGL(fwd16_kernel<false>, g3, dim3(256), FWD16_LDS_BYTES, stream, fp);
}

}  // namespace gnr
