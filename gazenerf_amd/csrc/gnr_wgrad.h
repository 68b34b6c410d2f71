// gnr_wgrad.h -- the weight-gradient GEMMs' split-K reductions, batched (round 4).
//
// Every weight-gradient GEMM (gnr_wgrad.hip, gnr_wgrad16.hip) writes one partial tile per (split, tile) and a small kernel
// adds the splits in a fixed order.  Launched right behind its GEMM that reduction is a 10-35 us launch of mostly latency:
// 13 of them per weight set in gnr_bwd (0.24 ms, a constant per call -- 2.3 % of the stage at 8192 rays), 9 per
// gnr_upsample_bwd (0.12 ms).  A caller can hand the launchers a WgradDefer instead: each GEMM then takes its partial
// tiles from a bump allocator over the caller's scratch and QUEUES its reduction; wgrad_defer_flush launches ONE kernel
// that runs every queued reduction -- same per-job block count, same per-output order of additions, hence the same bits
// as the launch-per-GEMM order.  When the scratch is exhausted the queue is flushed and the scratch reused (stream order
// makes that safe), so any scratch of at least wgrad_scratch_floats() works.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace gnr {

struct WgradReduceParams {
    const float* partial;
    int splits, tiles_n, tiles_k;
    int tn_rows, tk_cols;     // workgroup tile of the GEMM kernel that wrote the partials
    int cs_q, vs_q;           // rider shares per split (wgrad_pipe_kernel: 2 tiles_k / 2 tiles_n; else 1)
    int n_valid, k_valid;
    float* dW;          // destination matrix (NULL: skip)
    int ldw, col_off;
    int enc_map;        // 1: column k is an encoding slot (2*step+h) -> reference channel; 2: the bf16x3 dump's order
    // optional extras
    const float* colsum_part; float* colsum_out; int colsum_ld; int batch, spi;   // out[b][n]
    const float* vec_part; float* vec_out;                                         // out[k]
};

struct Wgrad16ReduceParams {
    const float* partial; int splits; long n_pad, k_pad;
    int M, K; float* dW; int ldw; float* bias;      // dW NULL: only the bias column is kept; bias NULL: no bias column
};

constexpr int WG_DEFER_MAX = 16;           // reductions per batched launch (kernel arguments: 16 x (120 + 64) bytes)

struct WgradReduceBatch {                  // kernel argument of wgrad_reduce_batch_kernel
    int n;
    int kind[WG_DEFER_MAX];                // 0: WgradReduceParams (gnr_wgrad.hip), 1: Wgrad16ReduceParams (gnr_wgrad16.hip)
    unsigned first[WG_DEFER_MAX + 1];      // job j owns blocks first[j] .. first[j + 1] - 1
    WgradReduceParams r[WG_DEFER_MAX];
    Wgrad16ReduceParams r16[WG_DEFER_MAX];
};

struct WgradDefer {
    float* arena = nullptr;
    size_t arena_floats = 0, cursor = 0;
    bool failed = false;                   // a GEMM needed more scratch than the whole arena and was NOT launched: the entry point
                                           // that owns the queue returns an error (arenas sized by wgrad_arena_floats never get here)
    WgradReduceBatch batch{};
};

inline void wgrad_defer_init(WgradDefer* d, float* arena, size_t floats) {
    d->arena = arena; d->arena_floats = floats; d->cursor = 0; d->failed = false; d->batch.n = 0; d->batch.first[0] = 0;
}
// Launches the queued reductions (one kernel) and makes the whole scratch available again.
void wgrad_defer_flush(WgradDefer* d, hipStream_t st);
// `floats` of scratch for one GEMM's partial tiles; flushes first when the queue or the scratch is full.  NULL: the request
// exceeds the whole scratch.
float* wgrad_defer_take(WgradDefer* d, size_t floats, hipStream_t st);
void wgrad_defer_push(WgradDefer* d, const WgradReduceParams& rp, unsigned blocks);
void wgrad_defer_push16(WgradDefer* d, const Wgrad16ReduceParams& rp, unsigned blocks);

// scratch of one GEMM launched without a WgradDefer (the layout the launchers assume in that case)
size_t wgrad_scratch_floats();
// Arena for the queued reductions of one weight set / one upsampler backward whose largest product is max_m x max_k over `batch`
// images: four single-GEMM scratches, or what that product's partial tiles + rider shares need when there are more (image, tile)
// pairs than one round of workgroups -- the partial tiles of a GEMM grow with the batch then.
size_t wgrad_arena_floats(int batch, int max_m, int max_k);
// One weight-gradient GEMM as its launcher will see it: what launch_wgrad / launch_wgrad_img plan from.
struct WgradShape {
    int lda, n_valid, ldb, k_valid;
    long chunks_per_image, pixels_per_image;      // pixels_per_image == 0: chunk-channel-major dumps (the MLP); > 0: channels-first images
    int with_vec, bf16x3, small_tiles;
};
// Scratch floats (partial tiles + rider shares) the GEMM's PLAN needs over `batch` images -- the same host-only plan the launch runs.
size_t wgrad_need_floats(const WgradShape& g, int batch);
// The arena a caller must provide for its list of GEMMs: wgrad_arena_floats()'s bound, raised to the largest single need of the list
// (so a GEMM can always take its scratch from an empty arena: sized from the plan, not from a hand bound).
size_t wgrad_arena_floats_for(const WgradShape* gemms, int n, int batch, int max_m, int max_k);

#ifdef __HIPCC__
// wgrad16_reduce_kernel's body: dW[n][k] = sum_s partial[s][n][k] (s ascending: fixed order), bias[n] = the last padded
// column.  One thread per output, k fastest.  bid = block index inside the job.
__device__ __forceinline__ void wgrad16_reduce_body(const Wgrad16ReduceParams& rp, unsigned bid) {
    const long idx = (long)bid * 256 + threadIdx.x;
    const int kw = rp.K + (rp.bias ? 1 : 0);
    if (idx >= (long)rp.M * kw) return;
    const int n = (int)(idx / kw), k = (int)(idx - (long)n * kw);
    const float* p = rp.partial + (long)n * rp.k_pad + (k < rp.K ? k : rp.k_pad - 1);
    const long ss = rp.n_pad * rp.k_pad;
    float acc = 0.0f;
    int s = 0;
    for (; s + 8 <= rp.splits; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long)(s + u) * ss];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < rp.splits; ++s) acc += p[(long)s * ss];
    if (k < rp.K) { if (rp.dW) rp.dW[(long)n * rp.ldw + k] = acc; }      // NULL == not wanted (include/gnr.h)
    else rp.bias[n] = acc;
}
#endif

}  // namespace gnr
