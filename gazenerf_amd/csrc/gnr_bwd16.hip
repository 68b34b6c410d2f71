// gnr_bwd16.hip -- the fp32 dgrad chain on v_mfma_f32_16x16x4_f32, two waves per SIMD (gfx950; see gnr_chain16.h).
//
//   packT16_kernel       W^T as A fragments of the 16x16x4 chain, layer after layer in execution order
//   bwd16_chain_kernel   register-chained dgrad through RGB2..L0 for one 16-sample sub-chunk per wave (the mirror image
//                        of fwd16_kernel: same FLOPs), ReLU masks from the saved sign bits, every layer's dY dumped in
//                        the chunk-channel-major layout the weight-gradient kernels read, d(encoding) -> d(pts) partials
// Backward of MLPforNeRF.forward (models/mlp_nerf.py:95-119) and Embedder.forward (utils/model_utils.py:272-280).
#include "gnr_bwd_common.h"
#include "gnr_chain16.h"

namespace gnr {

__host__ __device__ constexpr int bl16_in_tiles(int l) { return l == 0 ? NT16_F : (l == 1 ? NT16_H2 : NT16_H); }
__host__ __device__ constexpr int bl16_out_tiles(int l) { return l == 0 ? NT16_H2 : ((l == 5 || l == 11) ? NT16_E : NT16_H); }
static_assert((size_t)bl16_in_tiles(0) * bl16_out_tiles(0) * 256 == bl_floats(0) && (size_t)bl16_in_tiles(5) * bl16_out_tiles(5) * 256 == bl_floats(5) &&
              (size_t)bl16_in_tiles(2) * bl16_out_tiles(2) * 256 == bl_floats(2), "same stream size as the 32x32x2 packing");

__global__ void packT16_kernel(const PackTParams pp) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < PACKEDT_FLOATS;
         e += (size_t)gridDim.x * blockDim.x) {
        int l = 0;
        size_t off = 0;
        while (l + 1 < N_BL && e >= off + bl_floats(l)) { off += bl_floats(l); ++l; }
        const size_t loc = e - off;
        const int kt_n = bl16_out_tiles(l);
        const int rowi = (int)(loc / 256), rem = (int)(loc % 256);
        const int kg = rowi / kt_n, kt = rowi % kt_n;
        const int lane = rem / 4, ee = rem % 4;
        const int i = lane & 15, gk = lane >> 4;
        const int n = d16_channel(kg, ee, gk);               // contraction index: forward output channel
        const int krow = 16 * kt + i;                        // output row of this backward layer
        int col = -1;
        if (pp.enc[l]) {
            // output rows follow the C/D layout of an encoding register file: row 4 g' + r' of tile kt belongs to
            // lane group g', register r' = lane value idx 4 kt + r'
            col = enc16_channel(4 * kt + (i & 3), i >> 2);
        } else if (krow < pp.k_valid[l]) {
            col = pp.col0[l] + krow;
        }
        float v = 0.0f;
        if (n < pp.n_valid[l] && col >= 0) v = pp.w[l][(size_t)n * pp.ld[l] + col];
        pp.packed[e] = v;
    }
}

// d(encoding) held as 4 tiles in the C/D layout (lane group g owns the 16 values it encoded) -> d(pts).
// Embedder backward: d/dp sin(a p) = a cos(a p), d/dp cos(a p) = -a sin(a p).  enc_base: the saved encoding of this
// lane's sample, slot stride CHUNK (chunk-channel-major, round 1's slot order).
__device__ __forceinline__ void enc_backward16(const f32x4 (&E)[NT16_H], const float* __restrict__ enc_base, int g,
                                               float& gx, float& gy, float& gz) {
    float d[ENC16];
#pragma unroll
    for (int s = 0; s < ENC16; ++s) d[s] = E[s >> 2][s & 3];
    const bool lowg = g < 2;
    const int base = g < 2 ? 7 * g : 14 + 8 * (g - 2);
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    if (g == 0) { ax += d[0]; az += d[1]; }
    if (g == 1) ay += d[0];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float dsn = lowg ? (q < 7 ? d[(2 + 2 * q) & 15] : 0.0f) : d[2 * q];
        const float dcs = lowg ? (q < 7 ? d[(3 + 2 * q) & 15] : 0.0f) : d[2 * q + 1];
        const int p = min(base + q, 29);
        const int h = p >= 15 ? 1 : 0, pp = p - 15 * h, fl = pp / 3, a = pp - 3 * fl;
        const float scale = (float)(1 << fl) * (h ? 32.0f : 1.0f);
        const int slot = 2 * (2 + 6 * fl + a) + h;           // sin; the cosine sits 3 steps = 6 slots later
        const float sv = enc_base[slot * CHUNK], cv = enc_base[(slot + 6) * CHUNK];
        const float c = scale * (cv * dsn - sv * dcs);
        ax += a == 0 ? c : 0.0f;
        ay += a == 1 ? c : 0.0f;
        az += a == 2 ? c : 0.0f;
    }
    ax += __shfl_xor(ax, 16); ay += __shfl_xor(ay, 16); az += __shfl_xor(az, 16);
    ax += __shfl_xor(ax, 32); ay += __shfl_xor(ay, 32); az += __shfl_xor(az, 32);
    gx += ax; gy += ay; gz += az;
}

__device__ __forceinline__ void apply_relu16(f32x4& acc, unsigned word, int t) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // bit -> all-ones / zero mask (v_bfe_i32), then one v_and_b32 on the bit pattern: 2 VALU per value
        const int m = (int)(word << (4 * (t & 7) + e)) >> 31;
        const float a = acc[e];
        acc[e] = __builtin_bit_cast(float, __builtin_bit_cast(int, a) & m);
    }
}

__global__ __launch_bounds__(256, 2) void bwd16_chain_kernel(const BwdParams bp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const long n_sub = 2 * bp.n_chunks;
    const long sub = (long)blockIdx.x * WAVES_PER_WG + wave;
    if (sub >= n_sub) return;
    const ClkProbe clk0 = clk_begin();
    const long chunk = sub >> 1;
    const int hh = (int)(sub & 1);
    const long ray_g = chunk / bp.chunks_per_ray;
    const long row = chunk * CHUNK + hh * SUB + j;
    const long M = bp.M;
    dephase_first_round(blockIdx.x);
    WStream16 w;
    wstream16_init(w, bp.packedT, lane);
    const float* enc_base = bp.enc + chunk * (CHUNK * ENC_PAD) + hh * SUB + j;     // CCM: slot stride 32
    f32x4 A[NT16_H], Bv[NT16_H];
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;

    // d(feat_i) = w_i * g  (18 tiles) -> A
    {
        const float wg = bp.wglob[row];
        const float* gr = bp.gT + ray_g * FEAT_PAD + 4 * g;
#pragma unroll
        for (int t = 0; t < NT16_F; ++t) {
            const f32x4 g4 = *(const f32x4*)(gr + 16 * t);
            A[t][0] = wg * g4.x; A[t][1] = wg * g4.y; A[t][2] = wg * g4.z; A[t][3] = wg * g4.w;
        }
    }
    unsigned mk[RELU16_WORDS];
    auto bits = [&](int layer, int words) {
        const unsigned* src = bp.relu_bits + relu16_offset(layer, n_sub, sub);
#pragma unroll
        for (int q = 0; q < RELU16_WORDS; ++q)
            if (q < words) mk[q] = src[q * 64 + lane];
    };
    const unsigned lane_off = dump_lane_off16(j, g);
    auto dyh = [&](int l) { return dump_dst16(bp.dY_h + l * M * H, H, sub, lane_off); };
    auto none = [](int) {};
    // Each mm16_h dumps ITS INPUT (the dY of the layer above) while its MFMAs run; accumulators start from a zero
    // C operand; the ReLU mask of each output tile is applied in the loop tail.
#define GNR_MASK(X) [&](int t) { apply_relu16(X[t], mk[t >> 3], t); }
    // RGB2^T: A(18) -> Bv(12), mask y1 > 0            (dumps dfeat)
    bits(8, 2);
    mm16_h<NT16_F, NT16_H2, true, true>(A, Bv, w, dump_dst16(bp.dfeat, FEAT_PAD, sub, lane_off), ZeroInit16(), GNR_MASK(Bv));
    // RGB1^T: Bv(12) -> A(24), no activation on y0    (dumps dY_r1)
    mm16_h<NT16_H2, NT16_H, true, true>(Bv, A, w, dump_dst16(bp.dY_r1, H2, sub, lane_off), ZeroInit16(), none);
    // RGB0^T: A -> Bv, + density head, mask h7        (dumps dY_r0)
    bits(7, 3);
    {
        const float ds = bp.dsig[row];
        const float* wsg = bp.wsig + 4 * g;
        mm16_h<NT16_H, NT16_H, true, true>(A, Bv, w, dump_dst16(bp.dY_r0, H, sub, lane_off), ZeroInit16(), [&](int t) {
            const f32x4 w4 = *(const f32x4*)(wsg + 16 * t);
#pragma unroll
            for (int e = 0; e < 4; ++e) Bv[t][e] = fmaf(w4[e], ds, Bv[t][e]);
            apply_relu16(Bv[t], mk[t >> 3], t);
        });
    }
    // L7^T: Bv -> A mask h6 (dumps dY_7); L6^T: A -> Bv mask h5 (dumps dY_6)
    bits(6, 3);
    mm16_h<NT16_H, NT16_H, true, true>(Bv, A, w, dyh(7), ZeroInit16(), GNR_MASK(A));
    bits(5, 3);
    mm16_h<NT16_H, NT16_H, true, true>(A, Bv, w, dyh(6), ZeroInit16(), GNR_MASK(Bv));
    // L5: encoding columns first (4 tiles, A is dead here; dumps dY_5), then the hidden columns -> A mask h4
    mm16_h<NT16_H, NT16_E, true, true>(Bv, A, w, dyh(5), ZeroInit16(), none);
    enc_backward16(A, enc_base, g, gx, gy, gz);
    bits(4, 3);
    mm16_h<NT16_H, NT16_H, true, false>(Bv, A, w, Dump16{}, ZeroInit16(), GNR_MASK(A));
    // L4^T..L1^T (dump dY_4 .. dY_1)
#pragma unroll 1
    for (int rep = 0; rep < 2; ++rep) {
        const int la = 3 - 2 * rep, lb = 2 - 2 * rep;      // outputs dY_3, dY_2 then dY_1, dY_0
        bits(la, 3);
        mm16_h<NT16_H, NT16_H, true, true>(A, Bv, w, dyh(la + 1), ZeroInit16(), GNR_MASK(Bv));
        bits(lb, 3);
        mm16_h<NT16_H, NT16_H, true, true>(Bv, A, w, dyh(lb + 1), ZeroInit16(), GNR_MASK(A));
    }
#undef GNR_MASK
    // L0: encoding columns from dY_0 (in A; dumps dY_0)
    mm16_h<NT16_H, NT16_E, true, true>(A, Bv, w, dyh(0), ZeroInit16(), none);
    enc_backward16(Bv, enc_base, g, gx, gy, gz);

    // sub-chunk partials for the geometry gradient: sum dpts, sum z * dpts
    const float z = bp.zval[row];
    const float sx = row_sum16(gx), sy = row_sum16(gy), sz = row_sum16(gz);
    const float zx = row_sum16(gx * z), zy = row_sum16(gy * z), zz = row_sum16(gz * z);
    if (lane == 0) {
        float* gc = bp.geo_chunk + sub * 8;
        if (bp.accumulate_geo) {
            gc[0] += sx; gc[1] += sy; gc[2] += sz; gc[3] += zx; gc[4] += zy; gc[5] += zz;
        } else {
            gc[0] = sx; gc[1] = sy; gc[2] = sz; gc[3] = zx; gc[4] = zy; gc[5] = zz; gc[6] = 0.0f; gc[7] = 0.0f;
        }
    }
    clk_end(clk0, bp.clk);
}

void launch_packT16(const PackTParams& pt, hipStream_t stream) {
    hipLaunchKernelGGL(packT16_kernel, dim3(1024), dim3(256), 0, stream, pt);
}

void launch_bwd16_chain(const BwdParams& bp, hipStream_t stream) {
    const long n_sub = 2 * bp.n_chunks;
    hipLaunchKernelGGL(bwd16_chain_kernel, dim3((unsigned)((n_sub + WAVES_PER_WG - 1) / WAVES_PER_WG)), dim3(256), 0, stream, bp);
}

}  // namespace gnr
