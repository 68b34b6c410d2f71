/*
 * gnr.h -- C ABI of libgnr.so, the MI355X (gfx950) volumetric renderer for GazeNeRF.
 *
 * This is the drop-in boundary for ONE hot path of the reference:
 * everything between `self.sample_func(...)` (models/gaze_nerf.py:231) and the two
 * `self.calc_color_func(...)` calls (models/gaze_nerf.py:157-162):
 *
 *     GenSamplePoints.forward      utils/model_utils.py:364-375   ray dirs + plane-sweep samples
 *     Embedder.forward             utils/model_utils.py:272-280   positional encoding
 *     latent concat                models/gaze_nerf.py:136-143, 248-262
 *     MLPforNeRF.forward (x2)      models/mlp_nerf.py:95-119      face + eyes streams
 *     CalcRayColor.forward (x2)    utils/model_utils.py:516-534   alpha compositing
 *     FineSample.forward           utils/model_utils.py:413-490   importance resampling
 *
 * The reference has no FFI layer of its own (it is pure Python/PyTorch); the entry points below
 * are what a ctypes / pybind stub inside GazeNeRFNet.calc_color_with_code would bind
 * (INTEGRATION.md shows that stub).  Plain pointers and sizes only: no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated; tensors are contiguous, row-major,
 *     in the reference's own layouts (channels-first outputs [B, C, N_r]);
 *   - the caller owns every buffer, including the workspace; the library allocates nothing and
 *     keeps no state between calls (safe for several modules / devices per process).  Which kernels run depends on the
 *     arguments alone: the library reads NO environment variable (until round 3 GNR_CHAIN32 / GNR_CONV16_FORCE selected
 *     kernels behind the caller's back; the first is gone with the kernels it selected, the second is the explicit
 *     gnr_set_conv16_tile below).  The only process-wide state are the hooks declared next to gnr_set_kernel_timing;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no host synchronisation;
 *   - every function returns 0 on success, non-zero on error; gnr_last_error() then describes it
 *     (thread-local).  Bindings turn that into an exception so the reference's
 *     `try: ... except: continue` around a batch (trainer/gazenerf_trainer.py:576-582) keeps working.
 *   - size handshake (ABI 3): the three descriptor structs that have grown over time -- GnrProblem,
 *     GnrMergeProblem, GnrUpsampleProblem -- start with `uint32_t struct_size`, which the caller sets to ITS
 *     sizeof() of the struct (GNR_INIT_* below do it in C).  Every entry point that takes one compares it with the
 *     library's own sizeof and fails with a message naming both sizes: a binder compiled against an older or newer
 *     header gets an error, never fields read from the wrong offsets.  The pointer-table structs (GnrWeights,
 *     GnrOutputs, ...) are fixed by GNR_N_TRUNK / GNR_N_RGB / GNR_UPSAMPLE_MAX_BLOCKS; a binding checks them once
 *     at load time against gnr_sizeof().
 */
#ifndef GNR_H_
#define GNR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNR_ABI_VERSION 4
#define GNR_N_TRUNK 8          /* FeaExt_module_0..7   (models/mlp_nerf.py:29-58)  */
#define GNR_N_RGB 3            /* RGB_layer_0..2       (models/mlp_nerf.py:68-93)  */

/* One call of the hot path: B images x N_r rays x N_p samples.
 * Replaces the argument list of GenSamplePoints.forward (utils/model_utils.py:364) plus the
 * latent codes GazeNeRFNet._forward receives (models/gaze_nerf.py:211-224). */
typedef struct GnrProblem {
    uint32_t struct_size;   /* = sizeof(GnrProblem) as the CALLER was compiled (ABI 3); mismatch -> error */
    int32_t batch;          /* B                                                              */
    int32_t n_rays;         /* N_r per image                                                  */
    int32_t n_samples;      /* N_p per ray  (opt.num_sample_coarse, or 192 for the fine pass) */
    int32_t hidden;         /* opt.mlp_hidden_nchannels: 384 in the reference; any even width <= 384 is
                               accepted and runs zero-padded in the 384-wide kernels (same results,
                               384-wide cost; weight and gradient tensors keep their own shapes)       */
    int32_t feat_nc;        /* opt.featmap_nc; this build supports <= 288 (reference: 258)    */
    int32_t shape_dims;     /* 179 = iden + expr (configs/gazenerf_options.py:17)             */
    int32_t gaze_dims;      /* 2                                                              */
    int32_t appea_dims;     /* 127                                                            */
    float   world_z1;       /* 2.5   (configs/gazenerf_options.py:24)                         */
    float   world_z2;       /* -3.5                                                           */
    const float* xy;          /* [B,2,N_r]  pixel grid (utils/render_utils.py:24-32)          */
    const float* R;           /* [B,3,3]    cam-to-world rotation                             */
    const float* T;           /* [B,3,1]    camera centre                                     */
    const float* Kinv;        /* [B,3,3]    inverse intrinsics at feature-map scale           */
    const float* shape_code;  /* [B,shape_dims]                                               */
    const float* gaze;        /* [B,gaze_dims]                                                */
    const float* appea_code;  /* [B,appea_dims]                                               */
    const float* t_rand;      /* [B,N_r,N_p+1] stratified jitter in [0,1) (model_utils.py:306),
                                 or NULL == reference `disturb=False`                         */
    const float* z_edges;     /* [B,N_r,N_p+1] explicit sample edges (the sorted z of
                                 FineSample, model_utils.py:476-481) or NULL == plane sweep    */
    int32_t edges_follow_T;   /* only read by gnr_bwd when z_edges != NULL.  0: the edges are constants
                                 (d z / d T = 0).  1: they shift 1:1 with T_z, as FineSample's merged
                                 edges do -- it detaches only the weights, the coarse z it interpolates
                                 between are (T_z - world_z) terms (model_utils.py:418, 455-476, 339-357)
                                 -- so dT is that of the plane sweep.  ABI 2.                   */
    int32_t weights_packed;   /* only read by gnr_fwd / gnr_fwd_bf16x3 with save_for_backward == 0.  Non-zero:
                                 `workspace` still holds the re-laid-out weights a previous call of the SAME entry
                                 point wrote for exactly these weight values, the same number of weight sets, the same
                                 problem dimensions and the same workspace address; the call then skips its two
                                 re-layout kernels (2 x 5.4 MB written per call: 1-3 % of a 64 x 64-ray inference).
                                 The caller owns the invalidation (gazenerf_amd.render.PackedWeightCache keys it on
                                 tensor identity + torch's version counter).  ABI 2.             */
    int32_t vd_dims;          /* columns of RGB_layer_1.weight between the hidden and the appearance columns: 0, or
                                 27 with the reference's view-direction encoder (`include_vd`, models/gaze_nerf.py:
                                 70-80, 140: the layer's input is cat([h, vd_embedding(27), appea_code])).  The chain kernels
                                 skip those columns; their contribution is a per-ray bias of that layer (the direction is
                                 constant along a ray).  With ray_bias NULL for every weight set the LIBRARY computes it
                                 (ABI 3, gnr_vd.hip: Embedder of the normalised ray direction, utils/model_utils.py:253-280,
                                 366-369; vd_dims = 3 + 6 n_freqs, n_freqs <= 8), and gnr_bwd returns the gradient of the
                                 vd columns in rgb_w[1] and the direction's share of dR -- include_vd needs nothing else
                                 from the caller.  ABI 2: the caller had to supply ray_bias.                      */
    const float* ray_bias[2]; /* per weight set: NULL, or [B,N_r,hidden/2] added to the bias of RGB_layer_1 for every
                                 sample of the ray (a caller-computed fold of the vd columns, or any other per-ray term;
                                 gnr_bwd returns its gradient in GnrInputGrads.ray_bias).  With vd_dims > 0: set for every
                                 weight set or for none (see vd_dims); a mix is rejected.                         */
} GnrProblem;

/* Parameters of one MLPforNeRF (models/mlp_nerf.py:13-93).  weight = Conv2d [out,in,1,1] memory
 * == row-major [out,in]; bias [out].  Shapes for hidden H, vp = 63+shape_dims+gaze_dims:
 *   fea_w[0] [H,vp]  fea_w[1..4,6,7] [H,H]  fea_w[5] [H,vp+H]  density_w [1,H]
 *   rgb_w[0] [H,H]   rgb_w[1] [H/2,H+appea_dims]   rgb_w[2] [feat_nc,H/2]                     */
typedef struct GnrWeights {
    const float* fea_w[GNR_N_TRUNK];
    const float* fea_b[GNR_N_TRUNK];
    const float* density_w;
    const float* density_b;
    const float* rgb_w[GNR_N_RGB];
    const float* rgb_b[GNR_N_RGB];
} GnrWeights;

/* Same shapes, gradients (written, not accumulated).  Any pointer may be NULL == not wanted. */
typedef struct GnrWeightGrads {
    float* fea_w[GNR_N_TRUNK];
    float* fea_b[GNR_N_TRUNK];
    float* density_w;
    float* density_b;
    float* rgb_w[GNR_N_RGB];
    float* rgb_b[GNR_N_RGB];
} GnrWeightGrads;

/* Results of CalcRayColor.forward (utils/model_utils.py:516-534) per stream
 * (index 0 = first weight set / "face", 1 = second / "eyes").  depth / weights may be NULL. */
typedef struct GnrOutputs {
    float* feat[2];        /* [B,feat_nc,N_r]                                       */
    float* bg_alpha[2];    /* [B,1,N_r]                                             */
    float* depth[2];       /* [B,1,N_r]      or NULL                                */
    float* weights[2];     /* [B,1,N_r,N_p]  or NULL (face weights feed FineSample) */
} GnrOutputs;

/* Upstream gradients of the four outputs the caller consumes (NULL == zero). */
typedef struct GnrOutputGrads {
    const float* feat[2];      /* [B,feat_nc,N_r] */
    const float* bg_alpha[2];  /* [B,1,N_r]       */
} GnrOutputGrads;

/* Gradients w.r.t. the problem's differentiable inputs (written; NULL == not wanted).
 * No gradient is produced for xy, Kinv, t_rand or z_edges. */
typedef struct GnrInputGrads {
    float* R;            /* [B,3,3] */
    float* T;            /* [B,3,1] */
    float* shape_code;   /* [B,shape_dims] */
    float* gaze;         /* [B,gaze_dims]  */
    float* appea_code;   /* [B,appea_dims] */
    float* ray_bias[2];  /* [B,N_r,hidden/2] per weight set: d loss / d GnrProblem.ray_bias (NULL == not wanted) */
} GnrInputGrads;

enum {
    GNR_WS_FWD = 0,        /* scratch for an inference forward                                   */
    GNR_WS_FWD_SAVE = 1,   /* scratch + saved activations for a later gnr_bwd (training forward) */
    GNR_WS_BWD = 2         /* scratch for gnr_bwd (in addition to the saved forward workspace)   */
};

int gnr_abi_version(void);

/* C callers: `GnrProblem p = GNR_INIT_PROBLEM;` zero-initialises and stamps the size. */
#define GNR_INIT_PROBLEM          { (uint32_t)sizeof(GnrProblem) }
#define GNR_INIT_MERGE_PROBLEM    { (uint32_t)sizeof(GnrMergeProblem) }
#define GNR_INIT_UPSAMPLE_PROBLEM { (uint32_t)sizeof(GnrUpsampleProblem) }

/* sizeof() of the ABI structs as this library was compiled, so a binding in another language can verify its
 * own declarations at load time (the ctypes binding does).  Unknown id -> 0. */
enum { GNR_SIZEOF_PROBLEM = 0, GNR_SIZEOF_WEIGHTS = 1, GNR_SIZEOF_OUTPUTS = 2, GNR_SIZEOF_OUTPUT_GRADS = 3,
       GNR_SIZEOF_INPUT_GRADS = 4, GNR_SIZEOF_MERGE_PROBLEM = 5, GNR_SIZEOF_UPSAMPLE_PROBLEM = 6,
       GNR_SIZEOF_UPSAMPLE_WEIGHTS = 7 };
size_t gnr_sizeof(int which);

/* Bytes of workspace `kind` needs for problem `p` with `n_streams` (1 or 2) weight sets. */
size_t gnr_workspace_bytes(const GnrProblem* p, int n_streams, int kind);

/* Forward: samples -> encoding -> MLP(s) -> compositing.  `eyes` may be NULL (single stream: the
 * hierarchical fine pass with the third MLP, models/gaze_nerf.py:102-108).
 * save_for_backward != 0 keeps per-layer activations in `workspace` (size GNR_WS_FWD_SAVE);
 * that buffer must then reach gnr_bwd untouched. */
int gnr_fwd(const GnrProblem* p, const GnrWeights* face, const GnrWeights* eyes,
            const GnrOutputs* out, int save_for_backward,
            void* workspace, size_t ws_bytes, void* stream);

/* gnr_fwd / gnr_bwd with the dense layers on bf16 MFMA through a 3-term split (x = hi + lo;
 * a*b ~ a_hi*b_hi + a_lo*b_hi + a_hi*b_lo, fp32 accumulate): ~16 mantissa bits per operand at 5.3x the
 * fp32-MFMA rate.  Same arguments, workspace sizes and outputs as gnr_fwd / gnr_bwd, but the saved
 * activations use a different internal layout: gnr_bwd_bf16x3 must follow gnr_fwd_bf16x3 (and gnr_bwd
 * must follow gnr_fwd) on the same saved workspace.  Accuracy: measured against the reference run in fp64,
 * these kernels are as close to the exact result as the reference's own fp32 arithmetic in every case tried
 * (DESIGN.md section 4); against the reference's fp32 output the feature map stays within the 1e-4 contract
 * (<= 6e-6 on the test-mode fixtures, <= 6e-5 under the x50 opaque-head + train-jitter stress), bg_alpha
 * within 3e-5 on the fixtures and within ~1.2e-4 under that stress -- there the reference's fp32 run is itself
 * 1.5e-4 from its fp64 run.  Gradients stay inside the reference's own fp32-vs-fp64 noise on every tensor.  The weight-gradient GEMMs (dW = dY^T X) use the same split; bias, latent-code,
 * density-head and geometry gradients, the compositing backward and all reductions stay fp32. */
int gnr_fwd_bf16x3(const GnrProblem* p, const GnrWeights* face, const GnrWeights* eyes,
                   const GnrOutputs* out, int save_for_backward, void* workspace, size_t ws_bytes,
                   void* stream);
int gnr_bwd_bf16x3(const GnrProblem* p, const GnrWeights* face, const GnrWeights* eyes,
                   const GnrOutputGrads* dout, const GnrInputGrads* din, const GnrWeightGrads* dface,
                   const GnrWeightGrads* deyes, void* saved_workspace, size_t saved_bytes, void* scratch,
                   size_t scratch_bytes, void* stream);

/* Backward of gnr_fwd for the same problem and weights. */
int gnr_bwd(const GnrProblem* p, const GnrWeights* face, const GnrWeights* eyes,
            const GnrOutputGrads* dout,
            const GnrInputGrads* din, const GnrWeightGrads* dface, const GnrWeightGrads* deyes,
            void* saved_workspace, size_t saved_bytes,
            void* scratch, size_t scratch_bytes, void* stream);

/* FineSample.forward (utils/model_utils.py:413-490): inverse-CDF resampling of the coarse
 * weights.  weights [n_rays_total, n_coarse], coarse_z [n_rays_total, n_coarse] (left edges),
 * u [n_rays_total, n_fine+1] uniform draws or NULL == linspace(0,1,n_fine+1) (`disturb=False`).
 * Writes the sorted merged edges z_out [n_rays_total, n_coarse + n_fine + 1]; feed them back
 * through GnrProblem.z_edges with n_samples = n_coarse + n_fine.
 * PRECONDITION: every row of coarse_z is ASCENDING (what gnr_sample_zvals produces for world_z1 > world_z2, the
 * reference's 2.5 / -3.5, with or without jitter).  The kernel MERGES the two sorted lists instead of calling a
 * general sort (the reference's torch.sort accepts any order); descending rows give undefined z_out.  The
 * precondition is not checked on the device; gazenerf_amd.importance_resample(..., validate=True) checks it. */
int gnr_resample(const float* weights, const float* coarse_z, const float* u,
                 int64_t n_rays_total, int32_t n_coarse, int32_t n_fine,
                 float* z_out, void* stream);

/* Left sample edges of the plane sweep, [B,N_r,N_p] -- the `zvals` FineSample needs
 * (utils/model_utils.py:312-313).  Honours p->t_rand / p->z_edges. */
int gnr_sample_zvals(const GnrProblem* p, float* zvals_out, void* stream);

/* Measurement hook (process-wide: a framework may run gnr_bwd on another thread than the one that
 * armed the hook, e.g. PyTorch's autograd engine): subsequent gnr_fwd / gnr_bwd calls record the two
 * hipEvent_t (passed as void*) on their stream immediately before / after the dominant kernel of
 * the call (the fused MLP kernel in gnr_fwd; the dgrad-chain kernel of the first weight set in
 * gnr_bwd), so a harness can time that kernel alone.  Pass NULLs to switch it off (the default).
 * The hooks of this family (and gnr_set_conv16_tile) are the only process-wide state in the library; the timing hooks never
 * affect results. */
int gnr_set_kernel_timing(void* ev_start, void* ev_stop);

/* The same hook per stage (ABI 2): records the event pair around
 *   GNR_STAGE_FWD_MLP  the fused march-encode-MLP-composite kernel of gnr_fwd (both weight sets),
 *   GNR_STAGE_DGRAD    the dgrad-chain kernel of gnr_bwd, first weight set,
 *   GNR_STAGE_COMP_BWD the compositing backward of gnr_bwd, first weight set,
 *   GNR_STAGE_WGRAD    all weight-gradient GEMMs (+ their reductions) of gnr_bwd, first weight set.
 * gnr_set_kernel_timing == stages FWD_MLP and DGRAD with one pair; gnr_set_aux_timing == COMP_BWD. */
enum { GNR_STAGE_FWD_MLP = 0, GNR_STAGE_DGRAD = 1, GNR_STAGE_COMP_BWD = 2, GNR_STAGE_WGRAD = 3, GNR_N_STAGES = 4 };
int gnr_set_stage_timing(int stage, void* ev_start, void* ev_stop);

/* Shader-clock probe (measurement hook like the ones above; NULL = off, the default).  dev_counters = device memory,
 * uint64_t [GNR_N_STAGES][2], zeroed by the caller: every 64th workgroup of every MFMA-bound kernel of a stage (the fused
 * forward kernel; the dgrad chain; each weight-gradient GEMM) adds {shader cycles (s_memtime), 100 MHz reference
 * ticks (s_memrealtime)} it lived through.  cycles / ticks x 100 MHz = the clock the power management sustained under
 * THAT kernel's load -- the MFMA peaks in DESIGN.md are quoted at 2.4 GHz.  COMP_BWD is not probed (HBM-bound). */
int gnr_set_clock_probe(void* dev_counters);

/* Same, for the HBM-bound compositing pass of gnr_bwd (CalcRayColor backward over the saved per-sample
 * features, first weight set): lets a harness report achieved GB/s for that pass. */
int gnr_set_aux_timing(void* ev_start, void* ev_stop);

/* ---- first "next" row: feature-map merge, the step right after the hot path ----
 * Replaces models/gaze_nerf.py:175-203 and rotate()/rotation_matrix_2d (utils/model_utils.py:11-46):
 *   merge_face = feat_face + bg_alpha_face * bg_featmap;  merge_eyes likewise;
 *   eyes_planes = merge_eyes with every channel triplet rotated by Rot(gaze) = M2(yaw) M1(pitch);
 *   merge = max(merge_face, eyes_planes).
 * Maps are channels-first [B, feat_nc, n_pix] (n_pix = featmap_size^2), exactly what gnr_fwd writes. */
typedef struct GnrMergeProblem {
    uint32_t struct_size;                    /* = sizeof(GnrMergeProblem) of the caller (ABI 3) */
    int32_t batch, n_pix, feat_nc;           /* feat_nc % 3 == 0 (258 = 3 * 86)                 */
    const float* feat_face;     const float* bg_alpha_face;    /* [B,feat_nc,n_pix], [B,1,n_pix] */
    const float* feat_eyes;     const float* bg_alpha_eyes;
    const float* bg_featmap;    /* [1,feat_nc,n_pix]: NeuralRenderer.bg_featmap (neural_renderer.py:37-57) */
    const float* gaze;          /* [B,2] (pitch, yaw)                                              */
} GnrMergeProblem;

size_t gnr_merge_scratch_bytes(const GnrMergeProblem* p);
/* Outputs [B,feat_nc,n_pix]; any may be NULL. */
int gnr_merge_fwd(const GnrMergeProblem* p, float* merge_face, float* eyes_planes, float* merge, void* stream);
/* Upstream gradients (NULL == zero) -> gradients of the six inputs (NULL == not wanted; written). */
int gnr_merge_bwd(const GnrMergeProblem* p, const float* g_merge_face, const float* g_eyes_planes,
                  const float* g_merge, float* d_feat_face, float* d_bg_alpha_face, float* d_feat_eyes,
                  float* d_bg_alpha_eyes, float* d_bg_featmap, float* d_gaze,
                  void* scratch, size_t scratch_bytes, void* stream);

/* ---- N1: the 2-D upsampler after the hot path ------------------------------------------
 * NeuralRenderer.forward (models/neural_renderer.py:100-113) with PixelShuffleUpsample
 * (models/pixel_shuffle_upsample.py:33-42) and Blur (:7-16; kornia filter2d, reflect border, normalised):
 * [B, feat_nc, S, S] feature map -> [B, 3, S*2^n, S*2^n] image, n = log2(img_size / featmap_size).
 * Channel schedule ch[i] = max(feat_nc >> i, min_feat) (neural_renderer.py:57-98).  Weights are the
 * reference's Conv2d tensors ([out,in,1,1] == row-major [out,in]) under their state-dict names:
 *   up1_* = feat_upsample_list.i.layer_1 [2c,c], up2_* = ...layer_2 [4c,2c], feat_* = feat_layers.i [c',c],
 *   rgb_* = feat_2_rgb_list.i [3,c_i], i = 0..n.  bg_featmap is not an input of forward().
 * Limits (rejected with an error, gnr_last_error()): feat_nc <= 1024; 16 * ch[i] * side[i]^2 < 2^31 for every block (the
 * reference's 258 channels at 64 x 64 -> 512 x 512 use 1.7 % of that). */
#define GNR_UPSAMPLE_MAX_BLOCKS 4
enum { GNR_UP_WS_FWD = 0, GNR_UP_WS_BWD = 1 };

typedef struct GnrUpsampleProblem {
    uint32_t struct_size;           /* = sizeof(GnrUpsampleProblem) of the caller (ABI 3) */
    int32_t batch, feat_nc, featmap_size, n_blocks, min_feat;
    int32_t final_sigmoid;          /* NeuralRenderer(final_actvn=True) in the reference (gaze_nerf.py:115) */
    const float* x;                 /* [B, feat_nc, S, S]; S a power of two >= 16 */
} GnrUpsampleProblem;

typedef struct GnrUpsampleWeights {
    const float* up1_w[GNR_UPSAMPLE_MAX_BLOCKS];  const float* up1_b[GNR_UPSAMPLE_MAX_BLOCKS];
    const float* up2_w[GNR_UPSAMPLE_MAX_BLOCKS];  const float* up2_b[GNR_UPSAMPLE_MAX_BLOCKS];
    const float* feat_w[GNR_UPSAMPLE_MAX_BLOCKS]; const float* feat_b[GNR_UPSAMPLE_MAX_BLOCKS];
    const float* rgb_w[GNR_UPSAMPLE_MAX_BLOCKS + 1]; const float* rgb_b[GNR_UPSAMPLE_MAX_BLOCKS + 1];
} GnrUpsampleWeights;

typedef struct GnrUpsampleWeightGrads {     /* same shapes; NULL == not wanted; written, not accumulated */
    float* up1_w[GNR_UPSAMPLE_MAX_BLOCKS];  float* up1_b[GNR_UPSAMPLE_MAX_BLOCKS];
    float* up2_w[GNR_UPSAMPLE_MAX_BLOCKS];  float* up2_b[GNR_UPSAMPLE_MAX_BLOCKS];
    float* feat_w[GNR_UPSAMPLE_MAX_BLOCKS]; float* feat_b[GNR_UPSAMPLE_MAX_BLOCKS];
    float* rgb_w[GNR_UPSAMPLE_MAX_BLOCKS + 1]; float* rgb_b[GNR_UPSAMPLE_MAX_BLOCKS + 1];
} GnrUpsampleWeightGrads;

/* GNR_UP_WS_FWD: workspace of gnr_upsample_fwd (it keeps what the backward needs and must stay untouched
 * until gnr_upsample_bwd); GNR_UP_WS_BWD: scratch of gnr_upsample_bwd.  0 + gnr_last_error() on bad input. */
size_t gnr_upsample_workspace_bytes(const GnrUpsampleProblem* p, int kind);
int gnr_upsample_fwd(const GnrUpsampleProblem* p, const GnrUpsampleWeights* w, float* img, void* workspace,
                     size_t ws_bytes, void* stream);
/* d_img [B,3,S*2^n,S*2^n] -> d_x [B,feat_nc,S,S] (NULL == not wanted) and the weight gradients. */
int gnr_upsample_bwd(const GnrUpsampleProblem* p, const GnrUpsampleWeights* w, const float* d_img, float* d_x,
                     const GnrUpsampleWeightGrads* dw, void* saved_workspace, size_t saved_bytes, void* scratch,
                     size_t scratch_bytes, void* stream);

const char* gnr_last_error(void);

/* ABI 4.  What this binary was built from: "src=<16 hex digits: sha256 over gazenerf_amd/csrc/{*.hip,*.h,*.cpp} and this
 * header, in name order>;flags=<extra -D switches of a timing-experiment build, with the files they applied to>;
 * experimental=<0|1>".  The binaries are not under version control (they travel to the GPU box prebuilt): a binding
 * recomputes the hash from the tree it runs in and refuses a library built from other sources, or an experimental build
 * (gazenerf_amd/_lib.py does both).  Timing switches that change results only compile with -DGNR_EXPERIMENTAL_BUILD
 * (gazenerf_amd/csrc/gnr_internal.h), which `python -m gazenerf_amd.build` adds -- and reports here -- whenever extra flags
 * are given. */
const char* gnr_build_info(void);

/* ABI 4.  Tuning / test hook (process-wide like the timing hooks; results are the same to rounding for every choice): pin the
 * (row tiles, pixel tiles) instance of the upsampler's 1x1-convolution GEMMs -- 2x4, 4x4, 8x4, 9x2, 11x2, 13x2 -- instead of
 * the cost model's choice per GEMM; (0, 0) restores the cost model.  A pair without an instance is an error.  The blur-fused
 * feat_layers GEMM exists for 2x4, 4x4, 9x2; with another pair pinned the stencil runs as its own kernel.  The backward's
 * du GEMM with the fused un-shuffle epilogue (channel counts that are multiples of 4, sides that are multiples of 32) has its
 * own instances 2x8, 3x8, 4x8: pinning one of those leaves every other GEMM to the cost model; with a plain pair pinned the
 * un-shuffle runs as its own kernel behind the plain GEMM.  Which kernels run therefore depends on the arguments AND on this
 * pin; it must not be changed while a gnr_upsample_* call is in flight on another thread, nor between a forward and the
 * workspace-size query it was sized by (a call whose plan no longer fits its workspace returns an error). */
int gnr_set_conv16_tile(int row_tiles, int pixel_tiles);

#ifdef __cplusplus
}
#endif
#endif /* GNR_H_ */
