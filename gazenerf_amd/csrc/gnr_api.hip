// gnr_api.hip -- the C ABI of libgnr.so (include/gnr.h): validation, workspace carving, launches.
// No torch types, no allocation, no global state besides the thread-local error string and the
// (result-neutral) measurement hooks.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "gnr_canary.h"
#include "gnr_internal.h"

namespace gnr {
void launch_zvals(const GnrProblem& p, float* out, hipStream_t stream);
void launch_resample(const float* w, const float* cz, const float* u, long n_rays, int nc, int nf,
                     float* zout, hipStream_t stream);
int run_bwd(const GnrProblem* p, int n_streams, const GnrWeights* const* w, const GnrOutputGrads* dout,
            const GnrInputGrads* din, const GnrWeightGrads* const* dw, void* saved, size_t saved_bytes,
            void* scratch, size_t scratch_bytes, hipStream_t stream, bool bf16x3);
size_t bwd_scratch_bytes(const GnrProblem* p, int n_streams);
int conv16_set_tile(int mt, int nt);        // gnr_conv16.hip

static thread_local std::string g_err;
// measurement hooks: process-wide on purpose (see include/gnr.h)
std::atomic<hipEvent_t> g_stage_ev[GNR_N_STAGES][2];
void stage_mark(int stage, int which, hipStream_t st) {
    if (hipEvent_t e = g_stage_ev[stage][which].load()) (void)hipEventRecord(e, st);
}

std::atomic<unsigned long long*> g_clk_probe{nullptr};
unsigned long long* clock_probe_slot(int stage) {
    unsigned long long* p = g_clk_probe.load();
    return p ? p + 2 * stage : nullptr;
}

int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

#ifdef GNR_CANARY
// ---- carve-internal canaries (gnr_canary.h): experimental builds only -------------------------------------------------
namespace {
constexpr unsigned CANARY_WORD = 0xC5C5C5C5u;
constexpr int CANARY_MAX = 2048;
struct CanaryGap { void* at; const char* what; int index; bool fill; };
thread_local CanaryGap t_gaps[CANARY_MAX];
thread_local int t_n_gaps = 0;
thread_local bool t_fill = true, t_overflow = false;
__global__ void canary_fill_kernel(void* const* gaps, int n) {
    if ((int)blockIdx.x < n) {
        unsigned* g = (unsigned*)gaps[blockIdx.x];
        for (int i = threadIdx.x; i < (int)(CANARY_BYTES / 4); i += blockDim.x) g[i] = CANARY_WORD;
    }
}
__global__ void canary_fill_one_kernel(unsigned* g) {
    for (int i = threadIdx.x; i < (int)(CANARY_BYTES / 4); i += blockDim.x) g[i] = CANARY_WORD;
}
__global__ void canary_check_kernel(void* const* gaps, int n, unsigned* result) {      // result: {bad words, first bad gap + 1}
    if ((int)blockIdx.x >= n) return;
    const unsigned* g = (const unsigned*)gaps[blockIdx.x];
    unsigned bad = 0;
    for (int i = threadIdx.x; i < (int)(CANARY_BYTES / 4); i += blockDim.x) bad += g[i] != CANARY_WORD;
    if (bad) {
        atomicAdd(result, bad);
        atomicMin(result + 1, (unsigned)blockIdx.x);
    }
}
// device staging for the gap table and the result (an experimental build may allocate: include/gnr.h's rule is the product's)
struct CanaryDev { void** table; unsigned* result; };
CanaryDev canary_dev() {
    static thread_local CanaryDev d{nullptr, nullptr};
    if (!d.table) {
        (void)hipMalloc((void**)&d.table, CANARY_MAX * sizeof(void*));
        (void)hipMalloc((void**)&d.result, 2 * sizeof(unsigned));
    }
    return d;
}
}  // namespace
void canary_begin(bool fill) { t_n_gaps = 0; t_fill = fill; t_overflow = false; }
void canary_fill_mode(bool fill) { t_fill = fill; }
void canary_note(void* gap, const char* what, int index) {
    if (t_n_gaps >= CANARY_MAX) { t_overflow = true; return; }
    t_gaps[t_n_gaps++] = CanaryGap{gap, what, index, t_fill};
}
void canary_note_now(void* gap, const char* what, int index, hipStream_t st) {
    if (t_n_gaps >= CANARY_MAX) { t_overflow = true; return; }
    t_gaps[t_n_gaps++] = CanaryGap{gap, what, index, false};       // filled here, not by canary_arm
    hipLaunchKernelGGL(canary_fill_one_kernel, dim3(1), dim3(256), 0, st, (unsigned*)gap);
}
void canary_arm(hipStream_t st) {
    void* host[CANARY_MAX];
    int n = 0;
    for (int i = 0; i < t_n_gaps; ++i)
        if (t_gaps[i].fill) host[n++] = t_gaps[i].at;
    if (!n) return;
    const CanaryDev d = canary_dev();
    (void)hipStreamSynchronize(st);                       // the table is re-used from call to call
    (void)hipMemcpy(d.table, host, n * sizeof(void*), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(canary_fill_kernel, dim3((unsigned)n), dim3(256), 0, st, (void* const*)d.table, n);
    (void)hipStreamSynchronize(st);                       // (the table is rewritten by canary_check)
}
int canary_check(hipStream_t st, const char* entry) {
    if (t_overflow) return fail("%s: canary table overflow (more than %d regions)", entry, CANARY_MAX);
    if (!t_n_gaps) return 0;
    void* host[CANARY_MAX];
    for (int i = 0; i < t_n_gaps; ++i) host[i] = t_gaps[i].at;
    const CanaryDev d = canary_dev();
    (void)hipStreamSynchronize(st);
    const unsigned init[2] = {0u, 0xFFFFFFFFu};
    (void)hipMemcpy(d.table, host, t_n_gaps * sizeof(void*), hipMemcpyHostToDevice);
    (void)hipMemcpy(d.result, init, sizeof init, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(canary_check_kernel, dim3((unsigned)t_n_gaps), dim3(256), 0, st, (void* const*)d.table, t_n_gaps, d.result);
    unsigned res[2] = {0, 0};
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(res, d.result, sizeof res, hipMemcpyDeviceToHost);
    if (res[0] == 0) return 0;
    const CanaryGap& g = t_gaps[res[1] < (unsigned)t_n_gaps ? res[1] : 0];
    fprintf(stderr, "gnr canary: %s: %u word(s) overwritten; first damaged gap: behind region #%d of %s (%d gaps checked)\n", entry, res[0],
            g.index, g.what, t_n_gaps);
    return fail("%s: carve-internal canary overwritten: %u word(s), first behind region #%d of %s", entry, res[0], g.index, g.what);
}
#endif

int check_problem(const GnrProblem* p, int n_streams) {
    if (!p) return fail("gnr: problem is NULL");
    // size handshake (since ABI 3): a caller built against another header must not have its fields read at our offsets
    if (p->struct_size != sizeof(GnrProblem))
        return fail("gnr: GnrProblem.struct_size is %u but this libgnr.so (ABI %d) has sizeof(GnrProblem) = %zu: the caller "
                    "was built against a different include/gnr.h (or did not set struct_size)", p->struct_size,
                    GNR_ABI_VERSION, sizeof(GnrProblem));
    if (n_streams < 1 || n_streams > 2) return fail("gnr: n_streams must be 1 or 2 (got %d)", n_streams);
    // narrower networks run zero-padded in the 384-wide kernels: same results, 384-wide cost
    if (p->hidden < 2 || p->hidden > H || (p->hidden & 1))
        return fail("gnr: this build supports even hidden widths up to hidden=%d (got %d)", H, p->hidden);
    if (p->feat_nc < 1 || p->feat_nc > FEAT_PAD) return fail("gnr: feat_nc must be in [1,%d] (got %d)", FEAT_PAD, p->feat_nc);
    if (p->batch < 1 || p->n_rays < 1) return fail("gnr: empty problem (batch=%d n_rays=%d)", p->batch, p->n_rays);
    if (p->n_samples < 2) return fail("gnr: n_samples must be >= 2 (got %d)", p->n_samples);
    if ((p->n_samples + CHUNK - 1) / CHUNK > 16) return fail("gnr: n_samples must be <= 512 (got %d)", p->n_samples);
    if (p->shape_dims < 0 || p->gaze_dims < 0 || p->appea_dims < 0 || p->vd_dims < 0) return fail("gnr: negative latent dims");
    if (!p->xy || !p->R || !p->T || !p->Kinv) return fail("gnr: xy/R/T/Kinv must be non-NULL");
    if ((p->shape_dims && !p->shape_code) || (p->gaze_dims && !p->gaze) || (p->appea_dims && !p->appea_code))
        return fail("gnr: latent code pointer is NULL");
    // The view-direction columns are skipped by the chain kernels and come back as a per-ray bias (include/gnr.h): either
    // the caller supplies it for EVERY weight set, or for none -- then the library computes the embedding and the fold
    // itself (gnr_vd.hip).  One set with and one without would silently drop the columns of the latter.
    for (int s = n_streams; s < 2; ++s)      // a stale pointer there would switch the device-side fold off for set 0 as well
        if (p->ray_bias[s]) return fail("gnr: ray_bias[%d] is set but the call has %d weight set(s)", s, n_streams);
    if (p->vd_dims > 0) {
        int given = 0;
        for (int s = 0; s < n_streams; ++s) given += p->ray_bias[s] ? 1 : 0;
        if (given != 0 && given != n_streams)
            return fail("gnr: vd_dims = %d but ray_bias is set for %d of %d weight sets: supply it for all of them, or for "
                        "none (the library then folds the view-direction columns itself)", p->vd_dims, given, n_streams);
        if (given == 0 && vd_check(p)) return 1;
    }
    return 0;
}

static int check_weights(const GnrWeights* w, const char* tag) {
    if (!w) return fail("gnr: %s weights are NULL", tag);
    for (int i = 0; i < GNR_N_TRUNK; ++i)
        if (!w->fea_w[i] || !w->fea_b[i]) return fail("gnr: %s FeaExt_module_%d is NULL", tag, i);
    if (!w->density_w || !w->density_b) return fail("gnr: %s density_module is NULL", tag);
    for (int i = 0; i < GNR_N_RGB; ++i)
        if (!w->rgb_w[i] || !w->rgb_b[i]) return fail("gnr: %s RGB_layer_%d is NULL", tag, i);
    return 0;
}

// Carve the forward workspace.  base == nullptr only sizes it.
size_t carve_fwd(const GnrProblem* p, int n_streams, bool save, char* base, FwdParams* fp) {
    const int cpr = (p->n_samples + CHUNK - 1) / CHUNK;
    const size_t n_chunks = (size_t)p->batch * p->n_rays * cpr;
    const size_t M = n_chunks * CHUNK;
    size_t off = 0;
    int region = 0;
    auto take = [&](size_t floats) {
        float* ptr = base ? (float*)(base + off) : nullptr;
        off += align_up(floats * sizeof(float));
        if (CANARY_BYTES) {                                   // experimental builds (gnr_canary.h): a gap behind every region
            if (base) canary_note(base + off, "carve_fwd", region);
            off += CANARY_BYTES;
        }
        ++region;
        return ptr;
    };
    if (fp) {
        fp->prob = *p;
        fp->n_streams = n_streams;
        fp->chunks_per_ray = cpr;
        fp->n_chunks = (long)n_chunks;
        fp->M = (long)M;
        fp->save = save ? 1 : 0;
    }
    // the packed weight streams of all MLPs are contiguous (+ one ring of padding): the kernel's
    // prefetch ring runs straight from the last layer of one stream into the first of the next
    float* packed_all = take((size_t)n_streams * PACKED_FLOATS + 16 * 256);
    for (int s = 0; s < n_streams; ++s) {
        StreamWs w{};
        w.packed = packed_all ? packed_all + (size_t)s * PACKED_FLOATS : nullptr;
        w.bias = take((size_t)N_CHAIN * p->batch * H);
        w.wsig = take(H + 4);
        w.part_feat = take(2 * n_chunks * FEAT_PAD);      // fp32 kernels: one partial per 16-sample sub-chunk
        w.part_sc = take(2 * n_chunks * 4);
        w.wl = take(M);
        if (save) {
            w.act_h = take((size_t)8 * M * H);
            w.act_y0 = take(M * H);
            w.act_y1 = take(M * H2);
            w.act_feat = take(M * FEAT_PAD);
            w.sigma_raw = take(M);
            w.relu_bits = (unsigned*)take((size_t)9 * n_chunks * 6 * 64);
        }
        if (fp) fp->ws[s] = w;
    }
    float* zval = take(M);
    float* enc = nullptr; float* enc3 = nullptr; float* delta = nullptr; float* pts = nullptr;
    if (save) { enc = take(M * ENC_PAD); enc3 = take(M * ENC_PAD); delta = take(M); pts = take(M * 4); }
    if (fp) { fp->zval = zval; fp->enc = enc; fp->enc3 = enc3; fp->delta = delta; fp->pts = pts; fp->vd_embed = nullptr; }
    if (vd_on_device(p)) {          // embedding + one per-ray bias per weight set, kept for the backward
        float* vbase = take(vd_fwd_floats(p, n_streams));
        if (fp && vbase) {
            float* rb[2] = {nullptr, nullptr};
            vd_carve_fwd(p, n_streams, vbase, &fp->vd_embed, rb);
            for (int s = 0; s < n_streams; ++s) fp->ws[s].ray_bias = rb[s];
        }
    }
    return off;
}

}  // namespace gnr

using namespace gnr;

extern "C" {

int gnr_abi_version(void) { return GNR_ABI_VERSION; }

#ifndef GNR_BUILD_INFO          /* gazenerf_amd/build.py passes it; a hand-made build says so */
#define GNR_BUILD_INFO "src=unknown;flags=;experimental=0"
#endif
const char* gnr_build_info(void) { return GNR_BUILD_INFO; }

int gnr_set_conv16_tile(int row_tiles, int pixel_tiles) { return conv16_set_tile(row_tiles, pixel_tiles); }

const char* gnr_last_error(void) { return g_err.c_str(); }

size_t gnr_sizeof(int which) {
    switch (which) {
        case GNR_SIZEOF_PROBLEM: return sizeof(GnrProblem);
        case GNR_SIZEOF_WEIGHTS: return sizeof(GnrWeights);
        case GNR_SIZEOF_OUTPUTS: return sizeof(GnrOutputs);
        case GNR_SIZEOF_OUTPUT_GRADS: return sizeof(GnrOutputGrads);
        case GNR_SIZEOF_INPUT_GRADS: return sizeof(GnrInputGrads);
        case GNR_SIZEOF_MERGE_PROBLEM: return sizeof(GnrMergeProblem);
        case GNR_SIZEOF_UPSAMPLE_PROBLEM: return sizeof(GnrUpsampleProblem);
        case GNR_SIZEOF_UPSAMPLE_WEIGHTS: return sizeof(GnrUpsampleWeights);
        default: return 0;
    }
}

int gnr_set_stage_timing(int stage, void* ev_start, void* ev_stop) {
    if (stage < 0 || stage >= GNR_N_STAGES) return fail("gnr_set_stage_timing: unknown stage %d", stage);
    g_stage_ev[stage][0] = (hipEvent_t)ev_start;
    g_stage_ev[stage][1] = (hipEvent_t)ev_stop;
    return 0;
}

int gnr_set_clock_probe(void* dev_counters) {
    g_clk_probe = (unsigned long long*)dev_counters;
    return 0;
}

int gnr_set_aux_timing(void* ev_start, void* ev_stop) { return gnr_set_stage_timing(GNR_STAGE_COMP_BWD, ev_start, ev_stop); }

int gnr_set_kernel_timing(void* ev_start, void* ev_stop) {
    gnr_set_stage_timing(GNR_STAGE_FWD_MLP, ev_start, ev_stop);
    return gnr_set_stage_timing(GNR_STAGE_DGRAD, ev_start, ev_stop);
}

size_t gnr_workspace_bytes(const GnrProblem* p, int n_streams, int kind) {
    if (check_problem(p, n_streams)) return 0;
    switch (kind) {
        case GNR_WS_FWD: return carve_fwd(p, n_streams, false, nullptr, nullptr);
        case GNR_WS_FWD_SAVE: return carve_fwd(p, n_streams, true, nullptr, nullptr);
        case GNR_WS_BWD: return bwd_scratch_bytes(p, n_streams);
        default: fail("gnr: unknown workspace kind %d", kind); return 0;
    }
}

static int fwd_impl(const GnrProblem* p, const GnrWeights* face, const GnrWeights* eyes, const GnrOutputs* out,
                    int save_for_backward, void* workspace, size_t ws_bytes, void* stream, bool bf16x3) {
    const int n_streams = eyes ? 2 : 1;
    if (check_problem(p, n_streams)) return 1;
    if (check_weights(face, "first-stream")) return 1;
    if (eyes && check_weights(eyes, "second-stream")) return 1;
    if (!out) return fail("gnr_fwd: outputs are NULL");
    for (int s = 0; s < n_streams; ++s)
        if (!out->feat[s] || !out->bg_alpha[s]) return fail("gnr_fwd: feat/bg_alpha output %d is NULL", s);
    const bool save = save_for_backward != 0;
    const size_t need = carve_fwd(p, n_streams, save, nullptr, nullptr);
    if (!workspace || ws_bytes < need) return fail("gnr_fwd: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    if (((uintptr_t)workspace & 255) != 0) return fail("gnr_fwd: workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;

    FwdParams fp{};
    canary_begin(true);
    carve_fwd(p, n_streams, save, (char*)workspace, &fp);
    canary_arm(st);
    fp.want_wl = (out->weights[0] || (n_streams > 1 && out->weights[1])) ? 1 : 0;
    fp.clk = clock_probe_slot(GNR_STAGE_FWD_MLP);
    const GnrWeights* ws_in[2] = {face, eyes};
    if (vd_on_device(p)) {
        float* rb[2] = {(float*)fp.ws[0].ray_bias, (float*)fp.ws[1].ray_bias};
        launch_vd_fwd(*p, n_streams, ws_in, fp.vd_embed, rb, st);
    } else {
        for (int s = 0; s < n_streams; ++s) fp.ws[s].ray_bias = p->ray_bias[s];
    }
    // weights_packed: the caller vouches that the packed streams in this workspace are current (inference only)
    const bool reuse = p->weights_packed != 0 && !save;
    launch_prep(*p, n_streams, ws_in, fp.ws, st, !bf16x3 && !reuse);  // (the code-folded biases are always rebuilt)
    if (bf16x3 && !reuse) launch_prep3(*p, n_streams, ws_in, fp.ws, st);
    stage_mark(GNR_STAGE_FWD_MLP, 0, st);
    if (bf16x3) launch_fwd3(fp, st);
    else launch_fwd16(fp, st);
    stage_mark(GNR_STAGE_FWD_MLP, 1, st);

    CombineParams cp{};
    cp.prob = *p;
    cp.n_streams = n_streams;
    cp.chunks_per_ray = bf16x3 ? fp.chunks_per_ray : 2 * fp.chunks_per_ray;      // fp32: one partial per 16-sample sub-chunk
    cp.chunk_len = bf16x3 ? CHUNK : 16;
    for (int s = 0; s < n_streams; ++s) {
        cp.part_feat[s] = fp.ws[s].part_feat;
        cp.part_sc[s] = fp.ws[s].part_sc;
        cp.wl[s] = fp.ws[s].wl;
    }
    cp.out = *out;
    launch_combine(cp, st);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("gnr_fwd: launch failed: %s", hipGetErrorString(e));
    if (canary_check(st, "gnr_fwd")) return 1;
    return 0;
}

int gnr_fwd(const GnrProblem* p, const GnrWeights* face, const GnrWeights* eyes, const GnrOutputs* out,
            int save_for_backward, void* workspace, size_t ws_bytes, void* stream) {
    return fwd_impl(p, face, eyes, out, save_for_backward, workspace, ws_bytes, stream, false);
}

int gnr_fwd_bf16x3(const GnrProblem* p, const GnrWeights* face, const GnrWeights* eyes, const GnrOutputs* out,
                   int save_for_backward, void* workspace, size_t ws_bytes, void* stream) {
    return fwd_impl(p, face, eyes, out, save_for_backward, workspace, ws_bytes, stream, true);
}

static int bwd_impl(const GnrProblem* p, const GnrWeights* face, const GnrWeights* eyes, const GnrOutputGrads* dout,
                    const GnrInputGrads* din, const GnrWeightGrads* dface, const GnrWeightGrads* deyes,
                    void* saved_workspace, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream,
                    bool bf16x3) {
    const int n_streams = eyes ? 2 : 1;
    if (check_problem(p, n_streams)) return 1;
    if (check_weights(face, "first-stream")) return 1;
    if (eyes && check_weights(eyes, "second-stream")) return 1;
    if (!dout) return fail("gnr_bwd: output gradients are NULL");
    const GnrWeights* w[2] = {face, eyes};
    const GnrWeightGrads* dw[2] = {dface, deyes};
    return run_bwd(p, n_streams, w, dout, din, dw, saved_workspace, saved_bytes, scratch, scratch_bytes,
                   (hipStream_t)stream, bf16x3);
}

int gnr_bwd(const GnrProblem* p, const GnrWeights* face, const GnrWeights* eyes, const GnrOutputGrads* dout,
            const GnrInputGrads* din, const GnrWeightGrads* dface, const GnrWeightGrads* deyes,
            void* saved_workspace, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream) {
    return bwd_impl(p, face, eyes, dout, din, dface, deyes, saved_workspace, saved_bytes, scratch, scratch_bytes,
                    stream, false);
}

int gnr_bwd_bf16x3(const GnrProblem* p, const GnrWeights* face, const GnrWeights* eyes, const GnrOutputGrads* dout,
                   const GnrInputGrads* din, const GnrWeightGrads* dface, const GnrWeightGrads* deyes,
                   void* saved_workspace, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream) {
    return bwd_impl(p, face, eyes, dout, din, dface, deyes, saved_workspace, saved_bytes, scratch, scratch_bytes,
                    stream, true);
}

int gnr_resample(const float* weights, const float* coarse_z, const float* u, int64_t n_rays_total,
                 int32_t n_coarse, int32_t n_fine, float* z_out, void* stream) {
    if (!weights || !coarse_z || !z_out) return fail("gnr_resample: NULL pointer");
    if (n_rays_total < 1) return fail("gnr_resample: no rays");
    if (n_coarse < 3 || n_fine < 1) return fail("gnr_resample: need n_coarse >= 3 and n_fine >= 1");
    if (n_coarse + n_fine + 1 > 512) return fail("gnr_resample: n_coarse + n_fine + 1 must be <= 512");
    launch_resample(weights, coarse_z, u, (long)n_rays_total, n_coarse, n_fine, z_out, (hipStream_t)stream);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("gnr_resample: launch failed: %s", hipGetErrorString(e));
    return 0;
}

int gnr_sample_zvals(const GnrProblem* p, float* zvals_out, void* stream) {
    if (check_problem(p, 1)) return 1;
    if (!zvals_out) return fail("gnr_sample_zvals: output is NULL");
    launch_zvals(*p, zvals_out, (hipStream_t)stream);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("gnr_sample_zvals: launch failed: %s", hipGetErrorString(e));
    return 0;
}

}  // extern "C"
