// gnr_wgrad.hip -- weight gradients of the per-sample FC layers (gfx950).
//
//   dW[n][k] = sum_s dY[s][n] * X[s][k]      (s over all M samples of the call)
//
// dY and X are the row-major [M][C] dumps of the dgrad chain / training forward.  This is a plain
// "TN" GEMM whose contraction index is the sample: both MFMA operands want the sample on the
// k-slot (lane>>5) and 32 consecutive channels across lanes, which is exactly a row segment of the
// dumps -> coalesced global loads, conflict-free ds_read_b32.
//
// Tiling: 128(n) x 128(k) per 256-thread workgroup, 2x2 waves of 64x64 (2x2 v_mfma_f32_32x32x2_f32
// tiles, 64 accumulator registers), sample slabs of 32 double-buffered through LDS, the sample
// range split over gridDim.z with per-split partial tiles summed in a fixed order by
// wgrad_reduce_kernel (deterministic: the reference trains with cudnn.deterministic, train.py:57).
#include "gnr_device.h"

namespace gnr {

constexpr int WG_TN = 128, WG_TK = 128, WG_SLAB = 32, WG_LD = 132;   // LDS row stride (floats)

struct WgradParams {
    const float* A;      // dY [M][lda]
    const float* B;      // X  [M][ldb]
    int lda, ldb;
    int n_valid, k_valid;     // columns of A / B that exist (others read as zero)
    long M;
    long rows_per_split;      // multiple of WG_SLAB
    float* partial;           // [splits][tiles_n][tiles_k][128][128]
};

__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams wp) {
    __shared__ float lds[2][2][WG_SLAB][WG_LD];      // [buffer][A/B][sample][channel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;          // wave tile position (64x64 each)
    const int tn = blockIdx.x, tk = blockIdx.y, sp = blockIdx.z;
    const long s_begin = (long)sp * wp.rows_per_split;
    long s_end = s_begin + wp.rows_per_split;
    if (s_end > wp.M) s_end = wp.M;
    const int n0 = tn * WG_TN, k0 = tk * WG_TK;

    // staging: thread loads 4 float4 of A and 4 of B per slab: row = q*8 + tid/32, col4 = tid%32
    const int lr = tid >> 5, lc = (tid & 31) * 4;
    f32x4 ra[4], rb[4];
    auto gload = [&](long s0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long s = s0 + q * 8 + lr;
            f32x4 va = {0, 0, 0, 0}, vb = {0, 0, 0, 0};
            if (s < s_end) {
                const int ca = n0 + lc, cb = k0 + lc;
                if (ca + 3 < wp.n_valid) va = *(const f32x4*)(wp.A + s * wp.lda + ca);
                else {
                    if (ca + 0 < wp.n_valid) va.x = wp.A[s * wp.lda + ca + 0];
                    if (ca + 1 < wp.n_valid) va.y = wp.A[s * wp.lda + ca + 1];
                    if (ca + 2 < wp.n_valid) va.z = wp.A[s * wp.lda + ca + 2];
                }
                if (cb + 3 < wp.k_valid) vb = *(const f32x4*)(wp.B + s * wp.ldb + cb);
                else {
                    if (cb + 0 < wp.k_valid) vb.x = wp.B[s * wp.ldb + cb + 0];
                    if (cb + 1 < wp.k_valid) vb.y = wp.B[s * wp.ldb + cb + 1];
                    if (cb + 2 < wp.k_valid) vb.z = wp.B[s * wp.ldb + cb + 2];
                }
            }
            ra[q] = va;
            rb[q] = vb;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *(f32x4*)&lds[buf][0][q * 8 + lr][lc] = ra[q];
            *(f32x4*)&lds[buf][1][q * 8 + lr][lc] = rb[q];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const long n_slabs = (s_end - s_begin + WG_SLAB - 1) / WG_SLAB;
    if (n_slabs > 0) {
        gload(s_begin);
        lstore(0);
    }
    __syncthreads();
    const int li = lane & 31, lh = lane >> 5;
    for (long sl = 0; sl < n_slabs; ++sl) {
        const int buf = (int)(sl & 1);
        if (sl + 1 < n_slabs) gload(s_begin + (sl + 1) * WG_SLAB);
#pragma unroll
        for (int st = 0; st < WG_SLAB / 2; ++st) {
            const int s = 2 * st + lh;
            const float a0 = lds[buf][0][s][wn * 64 + li];
            const float a1 = lds[buf][0][s][wn * 64 + 32 + li];
            const float b0 = lds[buf][1][s][wk * 64 + li];
            const float b1 = lds[buf][1][s][wk * 64 + 32 + li];
            acc[0][0] = mfma32(a0, b0, acc[0][0]);
            acc[0][1] = mfma32(a0, b1, acc[0][1]);
            acc[1][0] = mfma32(a1, b0, acc[1][0]);
            acc[1][1] = mfma32(a1, b1, acc[1][1]);
        }
        if (sl + 1 < n_slabs) lstore(buf ^ 1);
        __syncthreads();
    }
    // partial tile: row i (n) = (r&3) + 8(r>>2) + 4 lh, col j (k) = li
    float* pt = wp.partial + (((long)sp * gridDim.x + tn) * gridDim.y + tk) * (WG_TN * WG_TK);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int j = wk * 64 + b * 32 + li;
                pt[i * WG_TK + j] = acc[a][b][r];
            }
}

struct WgradReduceParams {
    const float* partial;
    int splits, tiles_n, tiles_k;
    int n_valid, k_valid;
    float* dW;          // destination matrix
    int ldw, col_off;
    int enc_map;        // 1: column k is an encoding slot (2*step+h) -> reference channel
};

__global__ void wgrad_reduce_kernel(const WgradReduceParams rp) {
    const long total = (long)rp.tiles_n * WG_TN * rp.tiles_k * WG_TK;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e / (rp.tiles_k * WG_TK)), k = (int)(e % (rp.tiles_k * WG_TK));
        if (n >= rp.n_valid || k >= rp.k_valid) continue;
        const int tn = n / WG_TN, i = n % WG_TN, tk = k / WG_TK, j = k % WG_TK;
        float acc = 0.0f;
        for (int sp = 0; sp < rp.splits; ++sp)
            acc += rp.partial[(((long)sp * rp.tiles_n + tn) * rp.tiles_k + tk) * (WG_TN * WG_TK) + i * WG_TK + j];
        int col = k;
        if (rp.enc_map) {
            col = enc_channel(k >> 1, k & 1);
            if (col < 0) continue;
        }
        rp.dW[(long)n * rp.ldw + rp.col_off + col] = acc;
    }
}

// dW = A^T B over all M samples.  scratch_partial must hold splits*tiles_n*tiles_k*16384 floats.
size_t wgrad_partial_floats(long M, int n_valid, int k_valid, int* splits_out) {
    const int tiles_n = (n_valid + WG_TN - 1) / WG_TN, tiles_k = (k_valid + WG_TK - 1) / WG_TK;
    const long slabs = (M + WG_SLAB - 1) / WG_SLAB;
    long splits = 1024 / (tiles_n * tiles_k);          // ~4 workgroups per CU in flight
    if (splits < 1) splits = 1;
    if (splits > slabs) splits = slabs;
    if (splits_out) *splits_out = (int)splits;
    return (size_t)splits * tiles_n * tiles_k * WG_TN * WG_TK;
}

void launch_wgrad(const float* A, int lda, int n_valid, const float* B, int ldb, int k_valid, long M,
                  float* dW, int ldw, int col_off, int enc_map, float* partial, hipStream_t stream) {
    int splits = 1;
    wgrad_partial_floats(M, n_valid, k_valid, &splits);
    const int tiles_n = (n_valid + WG_TN - 1) / WG_TN, tiles_k = (k_valid + WG_TK - 1) / WG_TK;
    const long slabs = (M + WG_SLAB - 1) / WG_SLAB;
    WgradParams wp;
    wp.A = A; wp.B = B; wp.lda = lda; wp.ldb = ldb; wp.n_valid = n_valid; wp.k_valid = k_valid; wp.M = M;
    wp.rows_per_split = ((slabs + splits - 1) / splits) * WG_SLAB;
    wp.partial = partial;
    hipLaunchKernelGGL(wgrad_kernel, dim3(tiles_n, tiles_k, splits), dim3(256), 0, stream, wp);
    WgradReduceParams rp;
    rp.partial = partial; rp.splits = splits; rp.tiles_n = tiles_n; rp.tiles_k = tiles_k;
    rp.n_valid = n_valid; rp.k_valid = k_valid; rp.dW = dW; rp.ldw = ldw; rp.col_off = col_off; rp.enc_map = enc_map;
    const long total = (long)tiles_n * WG_TN * tiles_k * WG_TK;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, rp);
}

// ---------------------------------------------------------------------------------------------
// column sums of a row-major [rows][C] matrix per image: out[b][c] = sum_{s in image b} Y[s][c]
// (bias gradients; for the layers that see latent codes the per-image sums also give the latent
// and latent-column weight gradients).  Two deterministic stages.
// ---------------------------------------------------------------------------------------------
constexpr int CS_SPLITS = 512;

__global__ void colsum_stage1(const float* __restrict__ Y, int ld, int C, long rows_per_image, int splits,
                              float* __restrict__ part) {
    const int b = blockIdx.y, sp = blockIdx.x, c = threadIdx.x;
    if (c >= C) return;
    const long per = (rows_per_image + splits - 1) / splits;
    const long r0 = (long)b * rows_per_image + sp * per;
    long r1 = r0 + per;
    const long rmax = (long)(b + 1) * rows_per_image;
    if (r1 > rmax) r1 = rmax;
    float acc = 0.0f;
    for (long r = r0; r < r1; ++r) acc += Y[r * ld + c];
    part[((long)b * splits + sp) * C + c] = acc;
}

__global__ void colsum_stage2(const float* __restrict__ part, int C, int splits, float* __restrict__ out,
                              int out_ld) {
    const int b = blockIdx.x, c = threadIdx.x;
    if (c >= C) return;
    float acc = 0.0f;
    for (int sp = 0; sp < splits; ++sp) acc += part[((long)b * splits + sp) * C + c];
    out[(long)b * out_ld + c] = acc;
}

// out[b][0..C) (row stride out_ld) = per-image column sums.  part: B*CS_SPLITS*C floats.
void launch_colsum(const float* Y, int ld, int C, int batch, long rows_per_image, float* out, int out_ld,
                   float* part, hipStream_t stream) {
    int splits = CS_SPLITS;
    if (rows_per_image < splits) splits = (int)rows_per_image;
    const int threads = ((C + 63) / 64) * 64;
    hipLaunchKernelGGL(colsum_stage1, dim3(splits, batch), dim3(threads), 0, stream, Y, ld, C, rows_per_image,
                       splits, part);
    hipLaunchKernelGGL(colsum_stage2, dim3(batch), dim3(threads), 0, stream, part, C, splits, out, out_ld);
}

}  // namespace gnr
