// gnr_internal.h -- shared device/host definitions of libgnr (gfx950 only).
//
// Data layout in HBM:
//   * a "chunk" is 32 consecutive samples of one ray = the 32 columns of one
//     v_mfma_f32_32x32x2_f32 tile; one wavefront owns one chunk through the whole MLP chain;
//   * activations of a chunk live in registers in the MFMA C/D layout
//     (lane l: sample = l&31; register r of tile t: channel = 32t + (r&3) + 8(r>>2) + 4(l>>5)),
//     which is also the B-operand layout of the next layer when the k-steps are ordered
//     (t, r): lane-half h supplies channel 32t + (r&3) + 8(r>>2) + 4h.  Layer outputs therefore
//     feed the next layer with no data movement; only the weights stream;
//   * weights are pre-packed per layer into that k-order as MFMA A-fragments
//     P[step/4][n_tile][lane][4] so that a wave's load is one contiguous 1 KiB float4 row.
#pragma once
// Timing-experiment switches (tools/ab_*.sh, tools/ablate_fwd3.sh): several of them give WRONG or incomplete results.  They
// only compile in a build that declares itself experimental -- gazenerf_amd/build.py adds -DGNR_EXPERIMENTAL_BUILD whenever
// extra flags are given, gnr_build_info() then reports them, and the Python binding refuses such a library unless asked.
#if !defined(GNR_EXPERIMENTAL_BUILD) &&                                                                                     \
    (defined(GNR_NODUMP_TIMING) || defined(GNR_TEMPORAL_DUMP_TIMING) || defined(GNR_ABL16) || defined(GNR_C16_ABL) ||       \
     defined(GNR_PIPE_ABL) || defined(GNR_TR_ABL) || defined(GNR_ABLATE) || defined(GNR_WG_RIDERS) || defined(GNR_CANARY) ||  \
     defined(GNR_SOFTSTART))
#error "GNR_* timing switches need -DGNR_EXPERIMENTAL_BUILD (python -m gazenerf_amd.build adds it when GNR_EXTRA_HIPCC_FLAGS is set)"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gnr.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace gnr {

constexpr int H = 384;             // hidden width (opt.mlp_hidden_nchannels)
constexpr int NT_H = H / 32;       // 12 output tiles of 32 channels
constexpr int H2 = H / 2;          // RGB_layer_1 width (192)
constexpr int NT_H2 = H2 / 32;     // 6
constexpr int FEAT_PAD = 288;      // feat_nc (258) padded to 9 tiles
constexpr int NT_F = FEAT_PAD / 32;
constexpr int ENC_CH = 63;         // 3 + 6*10
constexpr int ENC_PAD = 64;
constexpr int ENC_STEPS = 32;      // k-steps (2 channels each) of the encoding
constexpr int N_CHAIN = 11;        // L0..L7, RGB0, RGB1, RGB2
constexpr int CHUNK = 32;          // samples per wavefront tile
constexpr int WAVES_PER_WG = 4;

// S16 dump layout of the fp32 chain kernels (gnr_chain16.h): inside a 16-sample sub-chunk of 16 C floats channel
// n = 16 t + 4 g + e occupies the 64-byte row 16 t + 4 e + g
__host__ __device__ constexpr int s16_row(int n) { return (n & ~15) + 4 * (n & 3) + ((n >> 2) & 3); }

// chain layer ids
enum { L0 = 0, L5 = 5, L7 = 7, LR0 = 8, LR1 = 9, LR2 = 10 };

// packed-weight sizes (floats) per chain layer: (enc steps + h steps) * 2 channels * n_pad
__host__ __device__ constexpr int layer_nt(int l) { return l == LR1 ? NT_H2 : (l == LR2 ? NT_F : NT_H); }
__host__ __device__ constexpr int layer_enc_steps(int l) { return (l == L0 || l == L5) ? ENC_STEPS : 0; }
__host__ __device__ constexpr int layer_h_steps(int l) { return l == L0 ? 0 : (l == LR2 ? H2 / 2 : H / 2); }
__host__ __device__ constexpr int layer_steps(int l) { return layer_enc_steps(l) + layer_h_steps(l); }
__host__ __device__ constexpr size_t layer_packed_floats(int l) {
    return (size_t)layer_steps(l) * layer_nt(l) * 64;
}
__host__ __device__ constexpr size_t packed_offset(int l) {
    size_t o = 0;
    for (int i = 0; i < l; ++i) o += layer_packed_floats(i);
    return o;
}
constexpr size_t PACKED_FLOATS = packed_offset(N_CHAIN);   // 1 357 824 per stream

// k-order of the positional encoding (our choice; the packer matches it).  Returns the
// reference channel (utils/model_utils.py:272-280 order) or -1 for the zero pad.
//   step 0: h0 -> x, h1 -> y;  step 1: h0 -> z, h1 -> pad;
//   step 2 + 6 fl + 3 sc + a: lane-half h handles frequency f = 5 h + fl, sc = 0 sin / 1 cos, axis a.
__host__ __device__ inline int enc_channel(int step, int h) {
    if (step == 0) return h;
    if (step == 1) return h == 0 ? 2 : -1;
    const int idx = step - 2, fl = idx / 6, q = idx % 6;
    return 3 + 6 * (5 * h + fl) + q;      // q = 3*sc + a matches [sin(3) | cos(3)] per frequency
}

// Per-stream device pointers into the workspace, filled by the host (gnr_api.hip).
struct StreamWs {
    float* packed;        // [PACKED_FLOATS]
    float* bias;          // [N_CHAIN][B][H]  (folded per image where the layer sees latents)
    float* wsig;          // [H + 4]: density weight, then density bias at [H]
    float* part_feat;     // [2 n_chunks][FEAT_PAD]  (sub-)chunk-local composited features (fp32: per 16 samples)
    float* part_sc;       // [2 n_chunks][4]: transmittance, sum w, sum w*z, -
    float* wl;            // [M] chunk-local weights alpha_i * T_local_i            (optional)
    // saved for backward (training forward only)
    float* act_h;         // [8][M][H]  post-ReLU trunk activations
    float* act_y0;        // [M][H]     RGB_layer_0 output (no activation)
    float* act_y1;        // [M][H2]    RGB_layer_1 output (post-ReLU)
    float* act_feat;      // [M][FEAT_PAD]
    float* sigma_raw;     // [M]
    unsigned* relu_bits;  // [9][n_chunks][6][64]: sign bits of h0..h7, y1 in register order (lane-major);
                          // fp32 kernels: [9][2 n_chunks][3][64] (gnr_chain16.h), the same bytes
    const float* ray_bias;  // GnrProblem.ray_bias of this weight set ([B*N_r][hidden/2]) or nullptr
};

struct FwdParams {
    GnrProblem prob;
    int n_streams;
    int chunks_per_ray;
    long n_chunks;        // B * N_r * chunks_per_ray
    long M;               // n_chunks * 32
    StreamWs ws[2];
    // shared per-sample geometry saved for backward / weights output
    float* enc;           // [M][ENC_PAD]  (our k-order)   (training only)
    float* enc3;          // [M][ENC_PAD]  the same as bf16 (hi, lo) quads in the QHL layout: the weight-gradient operand
                          //               of the bf16x3 path (gnr_chain3.h); the fp32 copy feeds the Embedder backward
    float* delta;         // [M]
    float* zval;          // [M]
    float* pts;           // [M][4]        (training only)
    float* vd_embed;      // [rays][28]  view-direction embedding when the library folds it itself (gnr_vd.hip) or nullptr
    int save;
    int want_wl;
    unsigned long long* clk;   // shader-clock probe or nullptr
};

// Shader-clock probe (gnr_set_clock_probe): two scalar counter reads at kernel entry, two at exit, two atomics by
// one thread of every 64th workgroup (workgroups of the chain kernels live ~0.5 ms: sampling only workgroup 0 would
// report the clock at the start of the launch, not the one sustained over it).
struct ClkProbe {
    unsigned long long c, r;
};
__device__ __forceinline__ ClkProbe clk_begin() { return ClkProbe{__builtin_readcyclecounter(), __builtin_amdgcn_s_memrealtime()}; }
__device__ __forceinline__ void clk_end(const ClkProbe& s, unsigned long long* clk) {
    if (clk && (blockIdx.x & 63) == 0 && threadIdx.x == 0) {
        atomicAdd(clk, __builtin_readcyclecounter() - s.c);
        atomicAdd(clk + 1, __builtin_amdgcn_s_memrealtime() - s.r);
    }
}
unsigned long long* clock_probe_slot(int stage);     // gnr_api.hip; nullptr when the probe is off

struct CombineParams {
    GnrProblem prob;
    int n_streams;
    int chunks_per_ray;       // partials per ray: chunks of 32 samples (bf16x3 kernels) or sub-chunks of 16 (fp32 kernels)
    int chunk_len;            // samples per partial (32 or 16)
    const float* part_feat[2];
    const float* part_sc[2];
    const float* wl[2];
    GnrOutputs out;
};

// host-side launchers (defined in the .hip translation units)
void launch_prep(const GnrProblem& p, int n_streams, const GnrWeights* const* w, StreamWs* ws,
                 hipStream_t stream, bool pack_fp32 = true);
void launch_fwd16(const FwdParams& fp, hipStream_t stream);
// gnr_vd.hip: the view-direction option computed by the library (vd_dims > 0, no caller-supplied ray_bias)
bool vd_on_device(const GnrProblem* p);
int vd_check(const GnrProblem* p);
size_t vd_fwd_floats(const GnrProblem* p, int n_streams);
void vd_carve_fwd(const GnrProblem* p, int n_streams, float* base, float** embed, float** rb);
void launch_vd_fwd(const GnrProblem& p, int n_streams, const GnrWeights* const* w, float* embed, float* const* rb, hipStream_t st);
void launch_prep3(const GnrProblem& p, int n_streams, const GnrWeights* const* w, StreamWs* ws,
                  hipStream_t stream);
void launch_fwd3(const FwdParams& fp, hipStream_t stream);
void launch_combine(const CombineParams& cp, hipStream_t stream);

}  // namespace gnr
