// gnr_bwd_common.h -- declarations shared by the fp32 (gnr_bwd.hip) and bf16x3 (gnr_bwd3.hip) dgrad chains.
#pragma once
#include "gnr_chain.h"

namespace gnr {

// ---------------------------------------------------------------------------------------------
// transposed weight stream.  Backward "layer" ids, in execution order:
//   0: RGB2^T (in 9 tiles -> out 6)   1: RGB1^T (6 -> 12)   2: RGB0^T (12 -> 12)
//   3,4: L7^T, L6^T (12 -> 12)   5: L5e^T (12 -> 2)   6: L5h^T (12 -> 12)   7..10: L4^T..L1^T
//   11: L0e^T (12 -> 2).  Memory order == execution order: the kernel reads one linear row stream.
// ---------------------------------------------------------------------------------------------
constexpr int N_BL = 12;
__host__ __device__ constexpr int bl_in_tiles(int l) { return l == 0 ? NT_F : (l == 1 ? NT_H2 : NT_H); }
__host__ __device__ constexpr int bl_out_tiles(int l) { return l == 0 ? NT_H2 : ((l == 5 || l == 11) ? 2 : NT_H); }
__host__ __device__ constexpr size_t bl_floats(int l) { return (size_t)bl_in_tiles(l) * 16 * bl_out_tiles(l) * 64; }
__host__ __device__ constexpr size_t bl_offset(int l) {
    size_t o = 0;
    for (int i = 0; i < l; ++i) o += bl_floats(i);
    return o;
}
constexpr size_t PACKEDT_FLOATS = bl_offset(N_BL);

struct PackTParams {
    const float* w[N_BL];
    int ld[N_BL];
    int n_valid[N_BL];     // forward outputs (contraction length here)
    int col0[N_BL];        // first source column of the outputs of this backward layer
    int k_valid[N_BL];     // number of valid output channels (hidden) ; enc layers: 64 slots
    int enc[N_BL];
    float* packed;
};

struct BwdParams {
    GnrProblem prob;
    int chunks_per_ray;
    long n_chunks, M;
    const float* packedT;
    const float* wsig;        // [H] density weight
    const float* gT;          // [rays][288]
    const float* wglob;       // [M]
    const float* dsig;        // [M]
    const unsigned* relu_bits;  // [9][n_chunks][6][64]
    const float* enc;         // [M][64] CCM
    const float* zval;        // [M]
    float* dY_h;              // [8][M][H]
    float* dY_r0;             // [M][H]
    float* dY_r1;             // [M][H2]
    float* dfeat;             // [M][288]
    float* geo_chunk;         // [n_chunks][8]: sum dpts (3), sum z*dpts (3)
    int accumulate_geo;
    unsigned long long* clk;  // shader-clock probe or nullptr
};

// d(encoding) held as a 2-tile C/D register file (lane-half h owns the slots it encoded) -> d(pts).
// Embedder backward: d/dp sin(a p) = a cos(a p), d/dp cos(a p) = -a sin(a p).
// QUAD: the saved encoding is in the channel-quad layout (enc_row = chunk base + 4 j), else
// chunk-channel-major (enc_row = chunk base + j).
template <bool QUAD = false>
__device__ __forceinline__ void enc_backward(const f32x16 (&E)[NT_H], const float* __restrict__ enc_row, int h,
                                             float& gx, float& gy, float& gz) {
    auto at = [&](int n) { return QUAD ? enc_row[(n >> 2) * 128 + (n & 3)] : enc_row[n * CHUNK]; };
    float d[ENC_STEPS];
#pragma unroll
    for (int s = 0; s < ENC_STEPS; ++s) d[s] = E[s >> 4][s & 15];
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    if (h == 0) { ax += d[0]; az += d[1]; } else { ay += d[0]; }
#pragma unroll
    for (int fl = 0; fl < 5; ++fl) {
        const float scale = (float)(1 << fl) * (h ? 32.0f : 1.0f);
        float acc3[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int si = 2 + 6 * fl + a, ci = si + 3;
            const float sv = at(2 * si + h), cv = at(2 * ci + h);
            acc3[a] = scale * (cv * d[si] - sv * d[ci]);
        }
        ax += acc3[0]; ay += acc3[1]; az += acc3[2];
    }
    ax += __shfl_xor(ax, 32);
    ay += __shfl_xor(ay, 32);
    az += __shfl_xor(az, 32);
    gx += ax; gy += ay; gz += az;
}

void launch_packT3(const PackTParams& pt, hipStream_t stream);
void launch_bwd3_chain(const BwdParams& bp, hipStream_t stream);

}  // namespace gnr
