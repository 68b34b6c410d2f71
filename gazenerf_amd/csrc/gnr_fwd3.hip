// gnr_fwd3.hip -- inference forward with the dense layers on bf16 MFMA via a 3-term split ("bf16x3").
//
// Every fp32 operand x is split x = hi + lo (+ ~2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi), and
//     a * b  ~=  a_hi b_hi + a_lo b_hi + a_hi b_lo          (fp32 accumulate in the MFMA)
// i.e. three v_mfma_f32_32x32x16_bf16 (32 matrix-pipe cycles each, 16 k per instruction) instead of
// eight v_mfma_f32_32x32x2_f32 (64 cycles each): 5.3x the fp32-MFMA rate at ~16 mantissa bits per
// operand.  Measured against the reference fixtures (emulation in tools, kernels in
// tests/test_parity_gpu.py): feature map 2e-7 .. 6e-6, bg_alpha <= 3.1e-5 even with the x50 opaque
// density head -- the same order as the fp32 path's own rounding noise and well inside the 1e-4
// contract, whereas plain bf16 is off by 1.5e-2.  SURVEY.md section 7 lists this split as the
// sanctioned "later, measured optimisation"; the fp32 path stays the default.
//
// Structure (differences from gnr_fwd.hip):
//   * a layer's output stays in registers as RAW fp32 accumulators; the NEXT layer converts one
//     32-channel input tile at a time (bias + ReLU + hi/lo split + bf16 pack: 16 registers), spread
//     between its own MFMA batches -- the conversion never stalls the matrix pipe as a block, and only
//     one converted tile (+ the one being built) is live, which frees ~170 registers;
//   * those registers hold a deeper weight prefetch: bf16x3 consumes 2 KiB of weights (hi + lo rows)
//     per 96 matrix-pipe cycles, 5x the fp32 kernel's rate;
//   * the density head is still an fp32 VALU dot (folded into the conversion of h7); ray geometry,
//     encoding (fp32 sincosf, then split), compositing and combine are shared with the fp32 path.
#include "gnr_chain.h"

namespace gnr {
int fail(const char* fmt, ...);

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
    // A: lane l holds A[i = l&31][k = 8(l>>5) + 0..7]; B: B[k = 8(l>>5) + 0..7][j = l&31]; C/D as f32 32x32
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// round-to-nearest-even fp32 -> bf16 (finite inputs)
__host__ __device__ __forceinline__ unsigned bf16_rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__host__ __device__ __forceinline__ float bf16_to_f32(unsigned b) { return __builtin_bit_cast(float, b << 16); }

// split a pair of fp32 values into packed {hi(a), hi(b)} and {lo(a), lo(b)}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    const unsigned ha = bf16_rne(a), hb = bf16_rne(b);
    hi = ha | (hb << 16);
    lo = bf16_rne(a - bf16_to_f32(ha)) | (bf16_rne(b - bf16_to_f32(hb)) << 16);
}

// k-order of a K=16 bf16 step s = 2t + u over an activation tile held in the C/D layout: lane-half h
// supplies, as element q (0..7), register r = 8u + q of tile t.
__host__ __device__ inline int dlayout3_channel(int step16, int h, int q) {
    const int t = step16 >> 1, r = 8 * (step16 & 1) + q;
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
}

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

// ---------------------------------------------------------------------------------------------
// weight stream: batches of WB3 rows (hi + lo = 2 buffer loads per row), NBUF batches in registers
// ---------------------------------------------------------------------------------------------
constexpr int WB3 = 2;              // rows per batch (4 KiB = 4 buffer loads per wave)
constexpr int NBUF = 3;             // batches in registers (48 VGPRs): 2 in flight = 384 matrix-pipe cycles

struct W3Stream {
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff;                       // lane*16 + byte offset of the batch most recently requested
    u32x4 g[NBUF][WB3][2];               // [buffer][row][hi/lo]
};

template <int Q>
__device__ __forceinline__ u32x4 w3load(const W3Stream& w) {
    // 1 KiB piece Q (0..3) of a 4 KiB batch, addressed through the 12-bit immediate
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.voff + (unsigned)Q * 1024u, 0, 0));
}
__device__ __forceinline__ void w3batch(const W3Stream& w, u32x4 (&g)[WB3][2]) {
    g[0][0] = w3load<0>(w); g[0][1] = w3load<1>(w);
    g[1][0] = w3load<2>(w); g[1][1] = w3load<3>(w);
}
__device__ __forceinline__ void w3_init(W3Stream& w, const float* packed, int lane) {
    w.rs = __builtin_amdgcn_make_buffer_rsrc((void*)packed, 0, 0x7ffffff0, 0x00020000);
    w.voff = (unsigned)lane * 16u;
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) {
        w3batch(w, w.g[b]);
        if (b + 1 < NBUF - 1) w.voff += WB3 * 2048u;
    }
}

// converted B operands of one 32-channel input tile: two K=16 steps, hi and lo
struct BTile {
    u32x4 h[2], l[2];
};

struct NoSide {
    __device__ __forceinline__ void operator()(int, int, float, float) const {}
};

// bias + optional ReLU + split of register pair (r, r+1) of raw accumulator tile `src` into `dst`
template <bool RELU, class Side>
__device__ __forceinline__ void convert_pair(const f32x16& src, const float* bias_tile, int h, int r, BTile& dst, int t,
                                             Side side) {
    const int ch = (r & 3) + 8 * (r >> 2) + 4 * h;                       // r even: channels ch, ch+1
    float a = src[r] + bias_tile[ch], b = src[r + 1] + bias_tile[ch + 1];
    if (RELU) { a = a > 0.0f ? a : 0.0f; b = b > 0.0f ? b : 0.0f; }
    side(t, r, a, b);
    unsigned hi, lo;
    split_pair(a, b, hi, lo);
    const int u = r >> 3, w = (r & 7) >> 1;
    dst.h[u][w] = hi;
    dst.l[u][w] = lo;
}

// ---- one dense layer from RAW previous-layer accumulators --------------------------------------
// prev[t] (+ prev_bias, ReLU?) is converted tile by tile while the MFMAs of the previous tile run.
template <int NT_IN, int NT_OUT, bool ZERO, bool RELU_IN, class Side = NoSide>
__device__ __forceinline__ void mm3_h(const f32x16 (&prev)[NT_H], const float* prev_bias, int h, f32x16 (&acc)[NT_H],
                                      W3Stream& w, Side side = Side()) {
    constexpr int ROWS_PER_TILE = 2 * NT_OUT;               // two K=16 steps
    constexpr int NROW = NT_IN * ROWS_PER_TILE;
    constexpr int NB = NROW / WB3, NB_TILE = ROWS_PER_TILE / WB3;
    static_assert(ROWS_PER_TILE % WB3 == 0 && NB % NBUF == 0, "rows must keep batch and buffer phase");
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    BTile cur, nxt;
#pragma unroll
    for (int r = 0; r < 16; r += 2) convert_pair<RELU_IN>(prev[0], prev_bias, h, r, cur, 0, side);
#pragma clang loop unroll(full)
    for (int t = 0; t < NT_IN; ++t) {
#pragma clang loop unroll(full)
        for (int bt = 0; bt < NB_TILE; ++bt) {
            const int kb = t * NB_TILE + bt;
            // request the batch NBUF-1 ahead (runs on into the next layer's rows)
            w.voff += WB3 * 2048u;
            w3batch(w, w.g[(kb + NBUF - 1) % NBUF]);
            // a slice of the NEXT input tile's conversion
            if (t + 1 < NT_IN) {
#pragma unroll
                for (int pr = (bt * 8) / NB_TILE; pr < ((bt + 1) * 8) / NB_TILE; ++pr)
                    convert_pair<RELU_IN>(prev[t + 1], prev_bias + 32 * (t + 1), h, 2 * pr, nxt, t + 1, side);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the rows of the batch interleaved term by term: consecutive MFMAs hit different accumulators
#pragma unroll
            for (int term = 0; term < 3; ++term) {
#pragma unroll
                for (int q = 0; q < WB3; ++q) {
                    const int i = bt * WB3 + q;             // row within the tile
                    const int u = i / NT_OUT, nt = i % NT_OUT;
                    const u32x4 a = w.g[kb % NBUF][q][term == 1 ? 1 : 0];      // hi, lo, hi
                    const u32x4 b = term == 2 ? cur.l[u] : cur.h[u];           // hi, hi, lo
                    acc[nt] = mfma_bf(a, b, (ZERO && term == 0 && t == 0 && u == 0) ? zero : acc[nt]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < NT_IN) cur = nxt;
    }
}

// ---- the 64-slot positional encoding (4 K=16 steps = two pre-converted BTiles in LDS) ------------
template <int NT_OUT>
__device__ __forceinline__ void mm3_enc(const unsigned* enc_col, f32x16 (&acc)[NT_H], W3Stream& w) {
    constexpr int NROW = 4 * NT_OUT;
    constexpr int NB = NROW / WB3;
    static_assert(NROW % WB3 == 0 && NB % NBUF == 0, "rows must keep batch and buffer phase");
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // LDS column layout: word index = tile*16 + (hi? 0 : 8) + u*4 + w, stride 256 threads
    u32x4 bh[4], bl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bh[s][c] = enc_col[((s >> 1) * 16 + (s & 1) * 4 + c) * 256];
            bl[s][c] = enc_col[((s >> 1) * 16 + 8 + (s & 1) * 4 + c) * 256];
        }
#pragma clang loop unroll(full)
    for (int kb = 0; kb < NB; ++kb) {
        w.voff += WB3 * 2048u;
        w3batch(w, w.g[(kb + NBUF - 1) % NBUF]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int term = 0; term < 3; ++term) {
#pragma unroll
            for (int q = 0; q < WB3; ++q) {
                const int i = kb * WB3 + q;
                const int s = i / NT_OUT, nt = i % NT_OUT;
                const u32x4 a = w.g[kb % NBUF][q][term == 1 ? 1 : 0];
                const u32x4 b = term == 2 ? bl[s] : bh[s];
                acc[nt] = mfma_bf(a, b, (term == 0 && s == 0) ? zero : acc[nt]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

constexpr size_t FWD3_LDS_BYTES = (size_t)(ENC_STEPS * 256 + WAVES_PER_WG * N_CHAIN * H) * sizeof(float);

__global__ __launch_bounds__(256, 1) void fwd3_kernel(const FwdParams fp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* enc_lds = (unsigned*)smem;                 // [32 words][256 threads]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* bias_lds = smem + ENC_STEPS * 256 + wave * (N_CHAIN * H);
    const int j = lane & 31, h = lane >> 5;
    const long chunk = (long)blockIdx.x * WAVES_PER_WG + wave;
    if (chunk >= fp.n_chunks) return;
    const GnrProblem& p = fp.prob;
    const int cpr = fp.chunks_per_ray;
    const long ray_g = chunk / cpr;
    const int c_in = (int)(chunk - ray_g * cpr);
    const int b = (int)(ray_g / p.n_rays);
    const int ray = (int)(ray_g - (long)b * p.n_rays);
    const int i = c_in * CHUNK + j;
    const bool valid = i < p.n_samples;
    const long row = chunk * CHUNK + j;

    W3Stream w;
    w3_init(w, fp.ws[0].packed, lane);

    const Ray r = make_ray(p, b, ray);
    const int ic = valid ? i : p.n_samples - 1;
    const float z0 = sample_edge(p, r.oz, ray_g, ic);
    const float z1 = sample_edge(p, r.oz, ray_g, ic + 1);
    const float delta = valid ? __fmul_rn(__fsub_rn(z1, z0), r.l) : 0.0f;
    const float px = __fadd_rn(r.ox, __fmul_rn(__fmul_rn(r.dx, r.l), z0));
    const float py = __fadd_rn(r.oy, __fmul_rn(__fmul_rn(r.dy, r.l), z0));
    const float pz = __fadd_rn(r.oz, __fmul_rn(__fmul_rn(r.dz, r.l), z0));
    if (fp.want_wl && h == 0) fp.zval[row] = z0;

    // encoding: fp32 sincosf, then hi/lo split into two pre-converted B tiles kept in LDS
    unsigned* enc_col = enc_lds + tid;
    {
        float e[ENC_STEPS];
        encode_point(px, py, pz, h, e);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) {
                    unsigned hi, lo;
                    split_pair(e[16 * T + 8 * u + 2 * wd], e[16 * T + 8 * u + 2 * wd + 1], hi, lo);
                    enc_col[(T * 16 + u * 4 + wd) * 256] = hi;
                    enc_col[(T * 16 + 8 + u * 4 + wd) * 256] = lo;
                }
    }

    f32x16 A[NT_H], Bv[NT_H];
#pragma unroll 1
    for (int s = 0; s < fp.n_streams; ++s) {
        const StreamWs& ws = fp.ws[s];
        {
            const long bstride = (long)p.batch * H;
            const float* bsrc = ws.bias + (long)b * H;
            for (int q = lane; q < N_CHAIN * (H / 4); q += 64) {
                const int l = q / (H / 4), c4 = q - l * (H / 4);
                *(f32x4*)(bias_lds + l * H + 4 * c4) = *(const f32x4*)(bsrc + l * bstride + 4 * c4);
            }
        }
        auto bl = [&](int l) { return bias_lds + l * H; };

        mm3_enc<NT_H>(enc_col, A, w);                                         // L0 (raw) -> A
#pragma unroll 1
        for (int rep = 0; rep < 2; ++rep) {                                   // L1..L4
            mm3_h<NT_H, NT_H, true, true>(A, bl(2 * rep), h, Bv, w);
            mm3_h<NT_H, NT_H, true, true>(Bv, bl(2 * rep + 1), h, A, w);
        }
        mm3_enc<NT_H>(enc_col, Bv, w);                                        // L5: enc part, then h4 part
        mm3_h<NT_H, NT_H, false, true>(A, bl(4), h, Bv, w);
        mm3_h<NT_H, NT_H, true, true>(Bv, bl(5), h, A, w);                    // L6
        mm3_h<NT_H, NT_H, true, true>(A, bl(6), h, Bv, w);                    // L7 (raw h7 in Bv)
        // RGB0 consumes h7 = relu(Bv + b7); the density head rides on the conversion (fp32 VALU dot)
        float sig = 0.0f;
        const float* wsg = ws.wsig + 4 * h;
        mm3_h<NT_H, NT_H, true, true>(Bv, bl(7), h, A, w, [&](int t, int rr, float a, float bb) {
            const int ch = 32 * t + (rr & 3) + 8 * (rr >> 2);
            sig = fmaf(wsg[ch], a, sig);
            sig = fmaf(wsg[ch + 1], bb, sig);
        });
        sig += __shfl_xor(sig, 32);
        sig += ws.wsig[H];
        mm3_h<NT_H, NT_H2, true, false>(A, bl(LR0), h, Bv, w);                // RGB1 from y0 (no activation)
        mm3_h<NT_H2, NT_F, true, true>(Bv, bl(LR1), h, A, w);                 // RGB2 from relu(y1)
        bias_act<NT_F, false>(A, bl(LR2), h);
        composite_chunk(A, sig, delta, z0, ws, chunk, row, lane, fp.want_wl != 0);
    }
}

void launch_prep3(const GnrProblem& p, int n_streams, const GnrWeights* const* wts, StreamWs* ws, hipStream_t stream) {
    const int vp = ENC_CH + p.shape_dims + p.gaze_dims;
    for (int s = 0; s < n_streams; ++s) {
        Pack3Params pp;
        for (int l = 0; l < N_CHAIN; ++l) {
            if (l <= 7) {
                pp.w[l] = wts[s]->fea_w[l];
                pp.ld[l] = (l == 0) ? vp : (l == 5 ? vp + H : H);
                pp.n_out[l] = H; pp.hcol[l] = (l == 5) ? vp : 0; pp.kh[l] = (l == 0) ? 0 : H;
            } else if (l == LR0) {
                pp.w[l] = wts[s]->rgb_w[0]; pp.ld[l] = H; pp.n_out[l] = H; pp.hcol[l] = 0; pp.kh[l] = H;
            } else if (l == LR1) {
                pp.w[l] = wts[s]->rgb_w[1]; pp.ld[l] = H + p.appea_dims; pp.n_out[l] = H2; pp.hcol[l] = 0; pp.kh[l] = H;
            } else {
                pp.w[l] = wts[s]->rgb_w[2]; pp.ld[l] = H2; pp.n_out[l] = p.feat_nc; pp.hcol[l] = 0; pp.kh[l] = H2;
            }
        }
        pp.packed = (unsigned short*)ws[s].packed;
        hipLaunchKernelGGL(pack3_kernel, dim3(1024), dim3(256), 0, stream, pp);
    }
}

void launch_fwd3(const FwdParams& fp, hipStream_t stream) {
    const unsigned grid = (unsigned)((fp.n_chunks + WAVES_PER_WG - 1) / WAVES_PER_WG);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)fwd3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FWD3_LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(fwd3_kernel, dim3(grid), dim3(256), FWD3_LDS_BYTES, stream, fp);
}

}  // namespace gnr
