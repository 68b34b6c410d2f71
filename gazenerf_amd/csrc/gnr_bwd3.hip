// gnr_bwd3.hip -- the dgrad chain (RGB2^T .. L0^T) on bf16 MFMA with the 3-term hi/lo split.
//
// Mirror of bwd_chain_kernel (gnr_bwd.hip) on the building blocks of gnr_chain3.h: the transposed
// weights stream through the workgroup's LDS ring as [hi][lo] bf16 rows, a layer's output stays in
// registers as raw fp32 accumulators and the NEXT layer's transform applies the ReLU mask (+ the
// density-head term), dumps the result -- that layer's dY, the A operand of the weight-gradient GEMMs,
// fp32 in the channel-quad layout of gnr_chain3.h (one 16-byte store per quad) -- and splits it into
// bf16 hi/lo.  Needs the saved workspace of gnr_fwd_bf16x3 (encoding and activations in that layout).
// tests/diagnostics/cpu_bf16x3_grad_probe.py: gradients of a bf16x3 step sit inside the reference's own fp32-vs-fp64
// noise on every tensor (worst rel-L2 9.96e-3 against 1.01e-2 for plain fp32; 3e-5 where fp32 has 3e-6).
#include "gnr_bwd_common.h"
namespace gnr { constexpr bool kChain3DumpBranch = true; }       // see gnr_chain3.h
#include "gnr_chain3.h"

namespace gnr {

__host__ __device__ constexpr size_t bl3_rows(int l) { return (size_t)bl_in_tiles(l) * 2 * bl_out_tiles(l); }
__host__ __device__ constexpr size_t bl3_row_offset(int l) {
    size_t o = 0;
    for (int i = 0; i < l; ++i) o += bl3_rows(i);
    return o;
}
constexpr size_t ROWS3T = bl3_row_offset(N_BL);
static_assert(ROWS3T * 512 == PACKEDT_FLOATS, "bf16x3 transposed stream must fit the fp32 stream's scratch slot");

// rows: (K=16 step s over the forward OUTPUT channels, output tile kt of this backward layer)
__global__ void pack3T_kernel(const PackTParams pp) {
    const size_t total = ROWS3T * 64 * 8;
    unsigned short* packed = (unsigned short*)pp.packed;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / 512;
        const int lane = (int)((e % 512) / 8), q = (int)(e % 8);
        int l = 0;
        size_t off = 0;
        while (l + 1 < N_BL && row >= off + bl3_rows(l)) { off += bl3_rows(l); ++l; }
        const int kt_n = bl_out_tiles(l);
        const int s = (int)((row - off) / kt_n), kt = (int)((row - off) % kt_n);
        const int h = lane >> 5;
        const int n = dlayout3_channel(s, h, q);             // contraction index: forward output channel
        const int krow = 32 * kt + (lane & 31);              // output row of this backward layer
        int col = -1;
        if (pp.enc[l]) {
            // same output-row convention as packT_kernel: an encoding register file in C/D layout
            const int tq = krow >> 5, iq = krow & 31;
            const int hq = (iq >> 2) & 1, r = (iq & 3) + 4 * (iq >> 3);
            col = enc_channel(16 * tq + r, hq);
        } else if (krow < pp.k_valid[l]) {
            col = pp.col0[l] + krow;
        }
        float v = 0.0f;
        if (n < pp.n_valid[l] && col >= 0) v = pp.w[l][(size_t)n * pp.ld[l] + col];
        const unsigned hi = bf16_rne(v);
        const unsigned lo = bf16_rne(v - bf16_to_f32(hi));
        unsigned short* rp = packed + row * 1024;
        rp[lane * 8 + q] = (unsigned short)hi;
        rp[512 + lane * 8 + q] = (unsigned short)lo;
    }
}

// LDS: [ring 48 KiB][density weight row]
constexpr size_t BWD3_LDS_BYTES = RING_BYTES + (size_t)H * sizeof(float);

__global__ __launch_bounds__(256, 1) void bwd3_chain_kernel(const BwdParams bp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* ring = (char*)smem;
    const ClkProbe clk0 = clk_begin();
    float* wsig_lds = smem + RING_BYTES / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    WRing w;
    ring_init(w, bp.packedT, (unsigned)(ROWS3T * 2048), ring, lane, wave);
    // past the end a wave recomputes the last chunk and stores identical values (see fwd3_kernel)
    const long chunk_raw = (long)blockIdx.x * WAVES_PER_WG + wave;
    const long chunk = chunk_raw < bp.n_chunks ? chunk_raw : bp.n_chunks - 1;
    const long ray_g = chunk / bp.chunks_per_ray;
    const long row = chunk * CHUNK + j;
    const long M = bp.M;
    for (int q = tid; q < H; q += 256) wsig_lds[q] = bp.wsig[q];
    const float* enc_row = bp.enc + chunk * (CHUNK * ENC_PAD) + 4 * j;     // channel-quad layout
    const float ds = bp.dsig[row];
    f32x16 A[NT_H], Bv[NT_H];
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;

    // d(feat_i) = w_i * g  (9 tiles) -> A
    {
        const float wv = bp.wglob[row];
        const float* gr = bp.gT + ray_g * FEAT_PAD + 4 * h;
#pragma unroll
        for (int t = 0; t < NT_F; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 g4 = *(const f32x4*)(gr + 32 * t + 8 * rq);
                A[t][4 * rq + 0] = wv * g4.x; A[t][4 * rq + 1] = wv * g4.y;
                A[t][4 * rq + 2] = wv * g4.z; A[t][4 * rq + 3] = wv * g4.w;
            }
    }
    __syncthreads();          // wsig row visible (also drains the ring prologue once; ring_start follows)
    ring_start(w);

    // sign bits of a forward layer: loaded one layer ahead of their use so that the load never waits in
    // the in-order vmcnt queue behind fresh LDS-DMA pieces
    unsigned mk[RELU_WORDS], mkn[RELU_WORDS];
    auto bits = [&](int layer) { return bp.relu_bits + relu_bits_offset(layer, bp.n_chunks, chunk); };
    auto dyh = [&](int l) { return qdump(bp.dY_h + l * M * H, H, chunk, j, h); };
    auto keep = [](unsigned word, int t, int rr, float v) {
        // bit 16 (t&1) + rr of the word -> all-ones / zero mask (v_bfe_i32 + v_and)
        const int m = __builtin_amdgcn_sbfe((int)word, 16 * (t & 1) + rr, 1);
        return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & m);
    };
    // transform of a layer input: [mask with the sign bits in mk]; the dump of the (masked) input is the QDump argument
    // of mm3_h (QHL layout, gnr_chain3.h)
#define GNR_XF(MASK, DP)                                                                          \
    [&](int t, int rr, f32x4& v) {                                                                \
        if (MASK) {                                                                               \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = keep(mk[t >> 1], t, rr + e, v[e]); \
        }                                                                                         \
    }, (DP)
    auto promote = [&]() {
#pragma unroll
        for (int q = 0; q < RELU_WORDS; ++q) mk[q] = mkn[q];
    };

    load_relu_bits<NT_H2>(mkn, bits(8), lane);
    // RGB2^T: A(9) -> Bv(6)                          (dumps dfeat)
    mm3_h<NT_F, NT_H2, INIT_ZERO, false, 1, 3>(A, Bv, nullptr, h, w, GNR_XF(false, qdump(bp.dfeat, FEAT_PAD, chunk, j, h)));    // 27 + 3 phases
    promote();
    load_relu_bits<NT_H>(mkn, bits(7), lane);
    // RGB1^T: Bv(6) -> A(12), input masked by y1 > 0  (dumps dY_r1)
    mm3_h<NT_H2, NT_H, INIT_ZERO, false, 1>(Bv, A, nullptr, h, w, GNR_XF(true, qdump(bp.dY_r1, H2, chunk, j, h)));
    // RGB0^T: A -> Bv, no activation on y0           (dumps dY_r0)
    mm3_h<NT_H, NT_H, INIT_ZERO, false, 1>(A, Bv, nullptr, h, w, GNR_XF(false, qdump(bp.dY_r0, H, chunk, j, h)));
    promote();
    load_relu_bits<NT_H>(mkn, bits(6), lane);
    // L7^T: Bv -> A; input = (d h7 + density head) masked by h7 > 0   (dumps dY_7)
    {
        const float* wsg = wsig_lds + 4 * h;
        mm3_h<NT_H, NT_H, INIT_ZERO, false, 1>(Bv, A, nullptr, h, w, [&](int t, int rr, f32x4& v) {
            const f32x4 w4 = *(const f32x4*)(wsg + 32 * t + 8 * (rr >> 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = keep(mk[t >> 1], t, rr + e, fmaf(w4[e], ds, v[e]));
        }, dyh(7));
    }
    promote();
    load_relu_bits<NT_H>(mkn, bits(5), lane);
    // L6^T: A -> Bv, input masked by h6 > 0           (dumps dY_6)
    mm3_h<NT_H, NT_H, INIT_ZERO, false, 1>(A, Bv, nullptr, h, w, GNR_XF(true, dyh(6)));
    promote();
    load_relu_bits<NT_H>(mkn, bits(4), lane);
    // L5: encoding columns first (2 tiles; masks h5 > 0 in place, dumps dY_5), then the hidden columns -> A
    mm3_h<NT_H, 2, INIT_ZERO, true, 1>(Bv, A, nullptr, h, w, GNR_XF(true, dyh(5)));
    enc_backward<true>(A, enc_row, h, gx, gy, gz);
    mm3_h<NT_H, NT_H, INIT_ZERO, false, 0>(Bv, A, nullptr, h, w, XfNone());
    // L4^T..L1^T (dump dY_4 .. dY_1)
#pragma unroll 1
    for (int rep = 0; rep < 2; ++rep) {
        const int la = 4 - 2 * rep, lb = 3 - 2 * rep;        // inputs dY_4, dY_3 then dY_2, dY_1
        promote();
        load_relu_bits<NT_H>(mkn, bits(la - 1), lane);
        mm3_h<NT_H, NT_H, INIT_ZERO, false, 1>(A, Bv, nullptr, h, w, GNR_XF(true, dyh(la)));
        promote();
        load_relu_bits<NT_H>(mkn, bits(lb - 1 >= 0 ? lb - 1 : 0), lane);
        mm3_h<NT_H, NT_H, INIT_ZERO, false, 1>(Bv, A, nullptr, h, w, GNR_XF(true, dyh(lb)));
    }
    promote();
    // L0: encoding columns from dY_0 (in A, masked by h0 > 0; dumps dY_0)
    mm3_h<NT_H, 2, INIT_ZERO, false, 1>(A, Bv, nullptr, h, w, GNR_XF(true, dyh(0)));
#undef GNR_XF
    enc_backward<true>(Bv, enc_row, h, gx, gy, gz);

    // chunk partials for the geometry gradient: sum dpts, sum z * dpts
    const float z = bp.zval[row];
    const float sx = half_sum32(gx), sy = half_sum32(gy), sz = half_sum32(gz);
    const float zx = half_sum32(gx * z), zy = half_sum32(gy * z), zz = half_sum32(gz * z);
    if (lane == 0 && chunk_raw < bp.n_chunks) {
        float* gc = bp.geo_chunk + chunk * 8;
        if (bp.accumulate_geo) {
            gc[0] += sx; gc[1] += sy; gc[2] += sz; gc[3] += zx; gc[4] += zy; gc[5] += zz;
        } else {
            gc[0] = sx; gc[1] = sy; gc[2] = sz; gc[3] = zx; gc[4] = zy; gc[5] = zz; gc[6] = 0.0f; gc[7] = 0.0f;
        }
    }
    wait_vm<0>();       // no LDS-DMA may outlive the wave
    clk_end(clk0, bp.clk);
}

void launch_packT3(const PackTParams& pt, hipStream_t stream) {
    hipLaunchKernelGGL(pack3T_kernel, dim3(1024), dim3(256), 0, stream, pt);
}

void launch_bwd3_chain(const BwdParams& bp, hipStream_t stream) {
    hipLaunchKernelGGL(bwd3_chain_kernel, dim3((unsigned)((bp.n_chunks + WAVES_PER_WG - 1) / WAVES_PER_WG)), dim3(256),
                       BWD3_LDS_BYTES, stream, bp);
}

}  // namespace gnr
