// gnr_conv16.h -- the 1x1-convolution GEMM of the upsampler, round 3: interface between
// gnr_conv16.hip (kernels) and gnr_upsample.hip (the NeuralRenderer forward / backward that launches them).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace gnr {

// Tile variant of one GEMM: a wave owns MT row tiles of 16 channels x NT pixel tiles of 16 pixels.
struct Conv16Plan {
    int MT, NT;
    int slices;          // row slices of 16 MT channels (the last one zero padded)
    int nkb;             // 16-wide blocks of the contraction index
    size_t pack_floats;  // size of the packed A operand: slices * nkb * MT * 256
};

// Chooses the variant for C[M][pixels] = A[M][K] B[K][pixels] over `pixels_total` = batch * pixels per image.
// `blur_w` != 0: the B operand is read through the 3x3 blur stencil of an image `blur_w` pixels wide (forward
// feat_layers); only some variants have that instance -- plan.MT == 0 on return means "none fits, run the stencil as
// its own kernel".
Conv16Plan conv16_plan(int M, int K, long pixels_total, int blur_w);
// The variant for du = Wf^T g with the un-shuffle fused into its epilogue (NT == 8: a lane owns the 2 x 2 blocks of two
// adjacent low-resolution pixels).  MT == 0 on return: not available (side % 32, or a tile pinned by gnr_set_conv16_tile) --
// run the plain GEMM and the un-shuffle kernel.  M % 4 == 0: pack with perm4 and pass dres; otherwise pack plainly, pass
// dres = NULL and collect dres from dpre2 afterwards (unshuffle_dres_kernel).
Conv16Plan conv16_plan_unshuffle(int M, int K, int side);

constexpr int CONV16_MAX_JOBS = 12;
struct Conv16PackJobs {
    int n;
    struct Job {
        const float* W; long rs, cs;      // A(m,k) = W[m*rs + k*cs]
        int M, K, MT, nkb, slices;
        int perm4;                        // packed row r holds A row (r >> 2) + (r & 3) * (M / 4)   (conv16_plan_unshuffle)
        int kchain;                       // element s of lane group g is k = 16 kb + 4 g + s (register-chained operand: the
                                          // previous GEMM's C/D layout) instead of 16 kb + 4 s + g (operand loaded from memory)
        long dst_off, floats;             // into `dst`
    } j[CONV16_MAX_JOBS];
    float* dst;
};
// Adds a job; returns the offset (floats) of its packed operand inside the pack buffer.
long conv16_add_job(Conv16PackJobs& jobs, const float* W, long rs, long cs, int M, int K, const Conv16Plan& plan, int perm4 = 0);
void launch_conv16_pack(const Conv16PackJobs& jobs, hipStream_t st);      // ONE launch for every GEMM of a call

struct Conv16Params {
    Conv16Plan plan;
    const float* At;                               // packed A operand of this GEMM
    const float* B; long b_batch;                  // B(b,k,n) = B[b*b_batch + k*P + n]
    float* C; long c_batch;                        // C(b,m,n) = C[b*c_batch + m*P + n]   (plain store)
    int M, K, P, batch;                            // P = pixels per image, P % 256 == 0
    const float* bias;                             // [M] or NULL
    int leaky;                                     // LeakyReLU(0.2) on (acc + bias)
    const float* mask_ref; long mask_batch;        // result *= (mask_ref(b,m,n) > 0 ? 1 : 0.2)
    int accumulate;                                // C += result
    // round 5: the x.repeat adjoint collected in this GEMM's epilogue instead of by unshuffle_dres_kernel (layer_1's data gradient,
    // channel counts that are no multiple of 4):  C = result + dres,  dres(b,m,n) = sum_q G(b, m + q M, n),
    //   G(b,k,n) = dres_from(b,k,n) * (bit k&3 of dres_sign(b,k>>2,n) ? 1 : 1/0.2)      (dres_from = dpre2 [batch][4 M][P])
    const float* dres_from; long dres_from_batch;
    const unsigned char* dres_sign; long dres_sign_batch;
    int shuffle, W;                                // PixelShuffleUpsample tail: n = y*W + x
    const float* res; long res_batch;              // residual res(b, m % (M/4), n)
    unsigned char* sign_out; long sign_batch;      // shuffle: bit e of byte (b, m/4, n) = (acc + bias > 0) of channel 4(m/4)+e
    int blur, H;                                   // B operand = blur(B) with reflect padding, image H x W (W above)
    // plan.NT == 8 (conv16_plan_unshuffle): the adjoint of the PixelShuffleUpsample tail in the epilogue.  M = C out-channels
    // of the shuffled map, B = g [K][2W x 2W] (P = 4 W W), rows packed with perm4;  C = dpre2 [4 C][W x W]:
    //   dpre2(4c + 2i + j, y, x) = du(c, 2y + i, 2x + j) * (bit 2i + j of sign_in(c, y, x) ? 1 : 0.2),
    //   dres(c', y, x) = sum_q du((c' >> 2) + q C / 4, 2y + ((c' >> 1) & 1), 2x + (c' & 1))          (x.repeat adjoint)
    const unsigned char* sign_in;                  // [batch][C][W*W] (sign_batch above)
    float* dres; long dres_batch;                  // NULL: rows in natural order, dpre2 only
    // plain epilogue with plan.slices == 1 (a wave holds every output channel of its pixels): the 3-channel RGB branch
    // rides on it (round 4; rgb_conv_kernel re-read the whole block output for it) --
    //   rgb(b,o,n) = [rgb(b,o,n) +] rgb_bias[o] + sum_m rgb_w[o][m] C(b,m,n);  rgb_img = sigmoid(rgb) if given;
    //   rgb_out = rgb_img if given else rgb (the caller's image, last block)
    const float* rgb_w; const float* rgb_bias;     // [3][M], [3]
    float* rgb; int rgb_accumulate;                // [batch][3][P]
    float* rgb_img; float* rgb_out;
};
// ---------------------------------------------------------------------------------------------------------------
// PixelShuffleUpsample's layer_1 -> layer_2 in one kernel (round 4): a1 = lrelu(W1 x + b1), u = shuffle(lrelu(W2 a1 + b2) +
// repeat(x)).  A wave owns 16 NP pixels and EVERY channel: phase 1 reads its operand from memory like conv16_kernel and keeps
// all M1 = 2C output rows in registers; their C/D layout IS the B-operand layout of phase 2 (gnr_chain16.h), which runs over
// the M2 = 4C output rows in slabs of 8 / NP row tiles.  a1 is still written (the backward reads it) but never read back.
struct UpChainPlan {
    int T1, NP;                       // row tiles of the middle activation, pixel tiles per wave; T1 == 0: no instance
    int nkb1, slabs2;                 // k-blocks of phase 1, slabs of 8 / NP row tiles of phase 2
    size_t pack1_floats, pack2_floats;
};
UpChainPlan upchain_plan(int K1, int M1, int M2, long pixels_per_image);
// the two pack jobs (phase 1: rows M1 x K1, phase 2: rows M2 x M1); offsets returned through o1 / o2
void upchain_add_jobs(Conv16PackJobs& jobs, const UpChainPlan& plan, const float* W1, long rs1, long cs1, int M1, int K1,
                      const float* W2, long rs2, long cs2, int M2, long* o1, long* o2);
struct UpChainParams {
    UpChainPlan plan;
    const float* A1; const float* A2;              // packed operands
    const float* B; long b_batch;                  // phase-1 operand [batch][K1][P]
    int K1, M1, M2, P, batch;
    const float* bias1;                            // [M1]
    float* out1; long out1_batch;                  // a1 [batch][M1][P]
    const float* bias2;                            // [M2]
    const float* res; long res_batch;              // residual x [batch][M2/4][P]
    unsigned char* sign_out; long sign_batch;      // [batch][M2/4][P]
    float* out2; long out2_batch;                  // u [batch][M2/4][4P] (pixel-shuffled)
    int W;                                         // image width (n = y W + x)
};
int launch_upchain(const UpChainParams& cp, hipStream_t st);

int launch_conv16(const Conv16Params& cp, hipStream_t st);       // non-zero (+ gnr_last_error) when the plan names no instance
int conv16_set_tile(int mt, int nt);                             // gnr_set_conv16_tile

}  // namespace gnr
