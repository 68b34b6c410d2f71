// gnr_fwd3.hip -- the fused forward (inference and training variants) with the dense layers on bf16 MFMA via a
// 3-term split ("bf16x3"); building blocks and the arithmetic in gnr_chain3.h.
//
// Accuracy (tests/test_parity_gpu.py): against the reference run in fp64 these kernels are as
// close to the exact result as the reference's own fp32 arithmetic; against its fp32 output the feature map stays
// within the 1e-4 contract (2e-7 .. 6e-6 on the fixtures), whereas plain bf16 is off by 1.5e-2.  SURVEY.md section 7
// lists this split as the sanctioned "later, measured optimisation"; the exact-fp32 kernels stay the default.
//
// Structure (differences from gnr_fwd.hip):
//   * a layer's output stays in registers as RAW fp32 accumulators that were started from the bias; the NEXT layer
//     converts one 32-channel input tile at a time (ReLU / sign bits / dump, hi/lo split, bf16 pack: 16 registers),
//     spread between its own MFMA groups, so only one converted tile and the one being built are live;
//   * the weight stream ([hi][lo] rows of 2 KiB, 2 KiB per 96 matrix-pipe cycles per wave) is fetched once per
//     workgroup into an LDS ring by LDS-DMA and read from there by all four waves (one barrier per 4 rows);
//   * the density head is still an fp32 VALU dot (folded into the conversion of h7); ray geometry, encoding (fp32
//     sincosf, then split), compositing and combine are shared with the fp32 path.
namespace gnr { constexpr bool kChain3DumpBranch = false; }      // see gnr_chain3.h
#include "gnr_chain3.h"

namespace gnr {
int fail(const char* fmt, ...);

// ---------------------------------------------------------------------------------------------
// packed weight stream: per row (K=16 step, n-tile): [64 lanes x 8 bf16 hi][64 lanes x 8 bf16 lo]
// = 2 KiB, rows in execution order; same total bytes as the fp32 stream (4 B per weight).
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int l3_enc_steps(int l) { return (l == L0 || l == L5) ? 4 : 0; }
__host__ __device__ constexpr int l3_h_steps(int l) { return l == L0 ? 0 : (l == LR2 ? 2 * NT_H2 : 2 * NT_H); }
__host__ __device__ constexpr size_t l3_rows(int l) { return (size_t)(l3_enc_steps(l) + l3_h_steps(l)) * layer_nt(l); }
__host__ __device__ constexpr size_t l3_row_offset(int l) {
    size_t o = 0;
    for (int i = 0; i < l; ++i) o += l3_rows(i);
    return o;
}
constexpr size_t ROWS3 = l3_row_offset(N_CHAIN);          // 2652 rows = 5.43 MB per MLP
static_assert(ROWS3 * 512 == PACKED_FLOATS, "bf16x3 stream must fit the fp32 stream's workspace slot");

struct Pack3Params {
    const float* w[N_CHAIN];
    int ld[N_CHAIN], n_out[N_CHAIN], hcol[N_CHAIN], kh[N_CHAIN];
    unsigned short* packed;
};

__global__ void pack3_kernel(const Pack3Params pp) {
    const size_t total = ROWS3 * 64 * 8;                     // (row, lane, q) triples
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / 512;
        const int lane = (int)((e % 512) / 8), q = (int)(e % 8);
        int l = 0;
        size_t off = 0;
        while (l + 1 < N_CHAIN && row >= off + l3_rows(l)) { off += l3_rows(l); ++l; }
        const int nt_n = layer_nt(l);
        const int s = (int)((row - off) / nt_n), nt = (int)((row - off) % nt_n);
        const int h = lane >> 5, n = 32 * nt + (lane & 31);
        int col = -1;
        if (s < l3_enc_steps(l)) {
            col = enc_channel(8 * s + q, h);
        } else {
            const int k = dlayout3_channel(s - l3_enc_steps(l), h, q);
            if (k < pp.kh[l]) col = pp.hcol[l] + k;
        }
        float v = 0.0f;
        if (n < pp.n_out[l] && col >= 0) v = pp.w[l][(size_t)n * pp.ld[l] + col];
        const unsigned hi = bf16_rne(v);
        const unsigned lo = bf16_rne(v - bf16_to_f32(hi));
        unsigned short* r = pp.packed + row * 1024;          // 2 KiB = 1024 shorts
        r[lane * 8 + q] = (unsigned short)hi;
        r[512 + lane * 8 + q] = (unsigned short)lo;
    }
}

// LDS: [ring 48 KiB][encoding 32 x 256 words][per-wave bias table (N_CHAIN layers + density row) x 4]
constexpr int BIAS_ROWS = N_CHAIN + 1;
constexpr size_t FWD3_LDS_BYTES = RING_BYTES + (size_t)(ENC_STEPS * 256 + WAVES_PER_WG * BIAS_ROWS * H) * sizeof(float);
static_assert(FWD3_LDS_BYTES <= 160 * 1024, "LDS budget");

// SAVE (training forward): the dumps of fwd_kernel<true> -- every layer's post-activation output, ReLU
// sign bits, sigma_raw, the encoding and the sample geometry -- with the activations and the encoding in
// the channel-quad layout (gnr_chain3.h) that gnr_bwd_bf16x3 expects; act_feat stays chunk-channel-major
// for comp_bwd_kernel.  A layer's output is dumped by the transform of the NEXT layer's mm3_h, i.e.
// spread over that layer's MFMA stream.
template <bool SAVE>
__global__ __launch_bounds__(256, 1) void fwd3_kernel(const FwdParams fp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* ring = (char*)smem;
    const ClkProbe clk0 = clk_begin();
    unsigned* enc_lds = (unsigned*)(smem + RING_BYTES / 4);          // [32 words][256 threads]
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* bias_lds = smem + RING_BYTES / 4 + ENC_STEPS * 256 + wave * (BIAS_ROWS * H);
    const int j = lane & 31, h = lane >> 5;
    const GnrProblem& p = fp.prob;

    WRing w;
    ring_init(w, fp.ws[0].packed, (unsigned)(fp.n_streams * ROWS3 * 2048), ring, lane, wave);

    // every wave stays in the barrier protocol: past the end it recomputes the last chunk and stores the
    // same values to the same places (no divergent control flow inside the layer code: each branch there
    // costs the register allocator ~200 spills)
    const long chunk_raw = (long)blockIdx.x * WAVES_PER_WG + wave;
    const long chunk = chunk_raw < fp.n_chunks ? chunk_raw : fp.n_chunks - 1;
    const int cpr = fp.chunks_per_ray;
    const long ray_g = chunk / cpr;
    const int c_in = (int)(chunk - ray_g * cpr);
    const int b = (int)(ray_g / p.n_rays);
    const int ray = (int)(ray_g - (long)b * p.n_rays);
    const int i = c_in * CHUNK + j;
    const bool valid = i < p.n_samples;
    const long row = chunk * CHUNK + j;

    const Ray r = make_ray(p, b, ray);
    const int ic = valid ? i : p.n_samples - 1;
    const float z0 = sample_edge(p, r.oz, ray_g, ic);
    const float z1 = sample_edge(p, r.oz, ray_g, ic + 1);
    const float delta = valid ? __fmul_rn(__fsub_rn(z1, z0), r.l) : 0.0f;
    const float px = __fadd_rn(r.ox, __fmul_rn(__fmul_rn(r.dx, r.l), z0));
    const float py = __fadd_rn(r.oy, __fmul_rn(__fmul_rn(r.dy, r.l), z0));
    const float pz = __fadd_rn(r.oz, __fmul_rn(__fmul_rn(r.dz, r.l), z0));

    // encoding: fp32 sincosf, then hi/lo split into two pre-converted B tiles kept in LDS
    unsigned* enc_col = enc_lds + tid;
    {
        float e[ENC_STEPS];
        encode_point(px, py, pz, h, e);
        // Lane (j, h) holds the slots 2 s + h; quad q = s>>1 of the dumps is {slot(2q, 0), slot(2q, 1), slot(2q+1, 0),
        // slot(2q+1, 1)}: half of it sits in the other lane half.  v_permlane32_swap_b32 (semantics pinned by
        // tools/ubench/permlane_probe.hip: lanes < 32 receive (x of lane, x of lane + 32), lanes >= 32 receive (y of
        // lane - 32, y of lane)) completes quad q in the lower half and quad q + 1 in the upper half: every dump is one
        // 16-byte store per lane (dword stores of the same data: 4x the instructions, +5 % on the training forward).
        auto swp = [](unsigned x, unsigned y, unsigned& a, unsigned& b) {
            auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
            a = r[0]; b = r[1];
        };
        const QDump qe = qdump(SAVE ? fp.enc3 : nullptr, ENC_PAD, chunk, j, h);
        float* ef = fp.enc + chunk * (CHUNK * ENC_PAD) + h * 128 + 4 * j;
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            unsigned hi[2], lo[2];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int T = (q + o) >> 3, u = ((q + o) >> 2) & 1, wd = (q + o) & 3;
                split_pair(e[2 * (q + o)], e[2 * (q + o) + 1], hi[o], lo[o]);
                enc_col[(T * 16 + u * 4 + wd) * 256] = hi[o];
                enc_col[(T * 16 + 8 + u * 4 + wd) * 256] = lo[o];
            }
            if (SAVE) {
                // the weight-gradient operand of the encoding columns, QHL layout (gnr_chain3.h) with the "channel"
                // order k' = 4 (s>>1) + 2 h + (s&1) of the slots 2 s + h (wgrad_reduce_kernel maps it back)
                unsigned a0, a1, b0, b1;
                swp(hi[0], hi[1], a0, a1);
                swp(lo[0], lo[1], b0, b1);
                dump_store((u32x4*)(qe.base + q * 512 + (((q >> 1) & 1) ? qe.s1 : qe.s0)),
                           ((q >> 2) & 1) ? u32x4{b0, b1, a0, a1} : u32x4{a0, a1, b0, b1});
                // fp32 copy for the Embedder backward: channel-quad layout, channel = encoding slot 2 s + h
                swp(__builtin_bit_cast(unsigned, e[2 * q]), __builtin_bit_cast(unsigned, e[2 * q + 2]), a0, a1);
                swp(__builtin_bit_cast(unsigned, e[2 * q + 1]), __builtin_bit_cast(unsigned, e[2 * q + 3]), b0, b1);
                dump_store((u32x4*)(ef + q * 128), u32x4{a0, a1, b0, b1});
            }
        }
        if (SAVE) {
            if (h == 0) {
                fp.delta[row] = delta;
                fp.zval[row] = z0;
                *(f32x4*)(fp.pts + row * 4) = f32x4{px, py, pz, 0.0f};
            }
        } else if (fp.want_wl && h == 0) {
            fp.zval[row] = z0;
        }
    }
    ring_start(w);

    f32x16 A[NT_H], Bv[NT_H];
#pragma unroll 1
    for (int s = 0; s < fp.n_streams; ++s) {
        const StreamWs& ws = fp.ws[s];
        {   // this wave's biases (latent codes folded in) + the density row -> LDS
            const long bstride = (long)p.batch * H;
            const float* bsrc = ws.bias + (long)b * H;
            for (int q = lane; q < N_CHAIN * (H / 4); q += 64) {
                const int l = q / (H / 4), c4 = q - l * (H / 4);
                *(f32x4*)(bias_lds + l * H + 4 * c4) = *(const f32x4*)(bsrc + l * bstride + 4 * c4);
            }
            for (int q = lane; q < H / 4; q += 64)
                *(f32x4*)(bias_lds + N_CHAIN * H + 4 * q) = *(const f32x4*)(ws.wsig + 4 * q);
            if (ws.ray_bias) {      // per-ray bias of RGB_layer_1 (view-direction columns folded by the caller, gnr.h)
                const int nrb = p.hidden / 2;
                const float* rb = ws.ray_bias + ray_g * nrb;
                for (int c = lane; c < nrb; c += 64) bias_lds[LR1 * H + c] += rb[c];
            }
        }
        auto bl = [&](int l) { return bias_lds + l * H; };
        // training forward: the transform of a layer input = activation + sign bits + dump of that input
        unsigned word = 0;
        auto qd = [&](float* dst, int C) { return SAVE ? qdump(dst, C, chunk, j, h) : QDump{nullptr, 0, 0}; };
        // ReLU in the transform (XfRelu: one v_max_i32 per value); the sign bits ride two MFMAs later (`late` of mm3_h):
        // bit = min(bits of relu(v), 1), word = (word << 1) | bit -- v_min_u32 + v_lshl_or_b32 per value
        auto xf_relu = [&](unsigned*) { return XfRelu(); };
        auto late_bits = [&](unsigned* bits) {
            return [=, &word](int t, int rr, const f32x4& v) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = v[e];
                    const unsigned b = __builtin_bit_cast(unsigned, x);
                    word = (word << 1) | (b < 1u ? b : 1u);
                }
                // 32 sign bits (tiles 2q, 2q+1) complete: first inserted = register 0 of the even tile
                if ((t & 1) && rr == 12) dump_store(bits + (t >> 1) * 64 + lane, __builtin_bitreverse32(word));
            };
        };
        auto lbits = [&](unsigned* bits) {
            if constexpr (SAVE) return late_bits(bits);
            else return XfLateNone();
        };
        auto sb = [&](int layer) { return SAVE ? ws.relu_bits + relu_bits_offset(layer, fp.n_chunks, chunk) : nullptr; };
        auto ah = [&](int l) { return SAVE ? ws.act_h + (long)l * fp.M * H : nullptr; };

        mm3_enc<NT_H>(enc_col, A, bl(0), h, w);                                // L0 -> A (pre-activation)
#pragma unroll 1
        for (int rep = 0; rep < 2; ++rep) {                                    // L1..L4
            const int la = 2 * rep + 1, lb = la + 1;
            mm3_h<NT_H, NT_H, INIT_BIAS, false, SAVE ? 1 : 0>(A, Bv, bl(la), h, w, xf_relu(sb(la - 1)), qd(ah(la - 1), H), lbits(sb(la - 1)));
            mm3_h<NT_H, NT_H, INIT_BIAS, false, SAVE ? 1 : 0>(Bv, A, bl(lb), h, w, xf_relu(sb(lb - 1)), qd(ah(lb - 1), H), lbits(sb(lb - 1)));
        }
        mm3_enc<NT_H>(enc_col, Bv, bl(5), h, w);                               // L5: encoding part, then h4 part
        mm3_h<NT_H, NT_H, INIT_NONE, false, SAVE ? 1 : 0>(A, Bv, bl(5), h, w, xf_relu(sb(4)), qd(ah(4), H), lbits(sb(4)));
        mm3_h<NT_H, NT_H, INIT_BIAS, false, SAVE ? 1 : 0>(Bv, A, bl(6), h, w, xf_relu(sb(5)), qd(ah(5), H), lbits(sb(5)));      // L6
        mm3_h<NT_H, NT_H, INIT_BIAS, false, SAVE ? 1 : 0>(A, Bv, bl(7), h, w, xf_relu(sb(6)), qd(ah(6), H), lbits(sb(6)));      // L7
        // RGB0 consumes h7 = relu(Bv); the density head rides on the conversion (fp32 VALU dot)
        float sig = 0.0f;
        {
            const float* wsg = bias_lds + N_CHAIN * H + 4 * h;
            auto base = xf_relu(sb(7));
            auto late7 = lbits(sb(7));
            mm3_h<NT_H, NT_H, INIT_BIAS, false, SAVE ? 1 : 0>(Bv, A, bl(LR0), h, w, [&, base](int t, int rr, f32x4& v) {
                base(t, rr, v);
                const f32x4 w4 = *(const f32x4*)(wsg + 32 * t + 8 * (rr >> 2));
                sig = fmaf(w4.x, v.x, sig);
                sig = fmaf(w4.y, v.y, sig);
                sig = fmaf(w4.z, v.z, sig);
                sig = fmaf(w4.w, v.w, sig);
            }, qd(ah(7), H), late7);
        }
        sig += __shfl_xor(sig, 32);
        sig += ws.wsig[H];
        if (SAVE && h == 0) ws.sigma_raw[row] = sig;
        mm3_h<NT_H, NT_H2, INIT_BIAS, false, SAVE ? 1 : 0>(A, Bv, bl(LR1), h, w, XfNone(), qd(ws.act_y0, H));
        mm3_h<NT_H2, NT_F, INIT_BIAS, false, SAVE ? 1 : 0, 3>(Bv, A, bl(LR2), h, w, xf_relu(sb(8)), qd(ws.act_y1, H2), lbits(sb(8)));   // 27 + 3 phases
        if (SAVE) dump<NT_F>(A, ws.act_feat, FEAT_PAD, chunk, j, h);
        composite_chunk(A, sig, delta, z0, ws, chunk, row, lane, SAVE || fp.want_wl != 0);
    }
    wait_vm<0>();       // no LDS-DMA may outlive the wave
    clk_end(clk0, fp.clk);
}

void launch_prep3(const GnrProblem& p, int n_streams, const GnrWeights* const* wts, StreamWs* ws, hipStream_t stream) {
    const int vp = ENC_CH + p.shape_dims + p.gaze_dims;
    const int Hh = p.hidden, Hh2 = Hh / 2;      // narrower networks: zero rows / columns beyond their width (launch_prep)
    for (int s = 0; s < n_streams; ++s) {
        Pack3Params pp;
        for (int l = 0; l < N_CHAIN; ++l) {
            if (l <= 7) {
                pp.w[l] = wts[s]->fea_w[l];
                pp.ld[l] = (l == 0) ? vp : (l == 5 ? vp + Hh : Hh);
                pp.n_out[l] = Hh; pp.hcol[l] = (l == 5) ? vp : 0; pp.kh[l] = (l == 0) ? 0 : Hh;
            } else if (l == LR0) {
                pp.w[l] = wts[s]->rgb_w[0]; pp.ld[l] = Hh; pp.n_out[l] = Hh; pp.hcol[l] = 0; pp.kh[l] = Hh;
            } else if (l == LR1) {
                pp.w[l] = wts[s]->rgb_w[1]; pp.ld[l] = Hh + p.vd_dims + p.appea_dims; pp.n_out[l] = Hh2; pp.hcol[l] = 0; pp.kh[l] = Hh;
            } else {
                pp.w[l] = wts[s]->rgb_w[2]; pp.ld[l] = Hh2; pp.n_out[l] = p.feat_nc; pp.hcol[l] = 0; pp.kh[l] = Hh2;
            }
        }
        pp.packed = (unsigned short*)ws[s].packed;
        hipLaunchKernelGGL(pack3_kernel, dim3(1024), dim3(256), 0, stream, pp);
    }
}

void launch_fwd3(const FwdParams& fp, hipStream_t stream) {
    const unsigned grid = (unsigned)((fp.n_chunks + WAVES_PER_WG - 1) / WAVES_PER_WG);
    // > 64 KiB of dynamic LDS needs an opt-in per device: set it on every launch (cheap, and correct for
    // several devices / threads per process -- a process-wide 'done' flag would not be)
    (void)hipFuncSetAttribute((const void*)(fp.save ? fwd3_kernel<true> : fwd3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FWD3_LDS_BYTES);
    if (fp.save)
        hipLaunchKernelGGL(fwd3_kernel<true>, dim3(grid), dim3(256), FWD3_LDS_BYTES, stream, fp);
    else
        hipLaunchKernelGGL(fwd3_kernel<false>, dim3(grid), dim3(256), FWD3_LDS_BYTES, stream, fp);
}

}  // namespace gnr
