// gnr_torch.cpp -- the PyTorch-ROCm C++ binding of libgnr.so ("PyTorch-ROCm C++/HIP extension").
//
// Host code only: it validates torch tensors (TORCH_CHECK -> Python exceptions, so the reference's
// `try: ... except: continue` around a batch, trainer/gazenerf_trainer.py:576-582, keeps working), sets the device
// guard, takes the CURRENT HIP stream of the tensors' device, allocates outputs and workspaces as torch tensors
// (the caching allocator owns every byte) and calls the C ABI of include/gnr.h.  No kernels live here, and nothing
// here computes: a missing / failing library call is an error, never a fallback.
//
// Python surface (gazenerf_amd/render.py picks this binding when the module is built, else the ctypes one):
//   render_fwd(...) -> [feat_0, bg_alpha_0, (depth_0), (weights_0), feat_1, ..., workspace]
//   render_bwd(...) -> [dR, dT, dshape, dgaze, dappea, 24 gradients per weight set ...]
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <string>
#include <vector>

#include "../../include/gnr.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<Tensor>;

void check_f32(const Tensor& t, const char* name, const c10::Device& dev) {
    TORCH_CHECK(t.defined(), name, " must be a tensor");
    TORCH_CHECK(t.device() == dev, name, " must live on ", dev, " (the render op has no CPU path), got ", t.device());
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32, got ", t.scalar_type());
}

struct Problem {
    GnrProblem c = GNR_INIT_PROBLEM;   // zeroed + struct_size stamped (size handshake)
    std::vector<Tensor> keep;          // contiguous versions the pointers refer to
    int64_t B = 0, n_r = 0, n_p = 0;
    c10::Device dev{c10::kCPU};
};

Problem make_problem(const Tensor& xy, const Tensor& R, const Tensor& T, const Tensor& Kinv, const Tensor& shape_code,
                     const Tensor& gaze, const Tensor& appea_code, const OptTensor& t_rand, const OptTensor& z_edges,
                     int64_t n_samples, double world_z1, double world_z2, int64_t hidden, int64_t feat_nc,
                     bool edges_follow_T, int64_t vd_dims = 0, const OptTensor& ray_bias0 = c10::nullopt,
                     const OptTensor& ray_bias1 = c10::nullopt) {
    Problem p;
    TORCH_CHECK(xy.defined() && xy.is_cuda(), "batch_xy must live on a CUDA/ROCm device (the render op has no CPU path)");
    p.dev = xy.device();
    check_f32(xy, "batch_xy", p.dev);
    TORCH_CHECK(xy.dim() == 3 && xy.size(1) == 2, "batch_xy must be [B,2,N_r], got ", xy.sizes());
    p.B = xy.size(0); p.n_r = xy.size(2); p.n_p = n_samples;
    auto want = [&](const Tensor& t, const char* name, std::vector<int64_t> shape) {
        check_f32(t, name, p.dev);
        TORCH_CHECK(t.sizes() == at::IntArrayRef(shape), name, " must have shape ", at::IntArrayRef(shape), ", got ", t.sizes());
    };
    want(R, "R", {p.B, 3, 3});
    want(T, "T", {p.B, 3, 1});
    want(Kinv, "Kinv", {p.B, 3, 3});
    for (auto pr : {std::make_pair(&shape_code, "shape_code"), std::make_pair(&gaze, "gaze"), std::make_pair(&appea_code, "appea_code")}) {
        check_f32(*pr.first, pr.second, p.dev);
        TORCH_CHECK(pr.first->dim() == 2 && pr.first->size(0) == p.B, pr.second, " must be [", p.B, ", dims], got ", pr.first->sizes());
    }
    if (t_rand) want(*t_rand, "t_rand", {p.B, p.n_r, n_samples + 1});
    if (z_edges) want(*z_edges, "z_edges", {p.B, p.n_r, n_samples + 1});
    auto keep = [&](const Tensor& t) { p.keep.push_back(t.contiguous()); return p.keep.back().data_ptr<float>(); };
    GnrProblem& c = p.c;
    c.batch = (int32_t)p.B; c.n_rays = (int32_t)p.n_r; c.n_samples = (int32_t)n_samples;
    c.hidden = (int32_t)hidden; c.feat_nc = (int32_t)feat_nc;
    c.shape_dims = (int32_t)shape_code.size(1); c.gaze_dims = (int32_t)gaze.size(1); c.appea_dims = (int32_t)appea_code.size(1);
    c.world_z1 = (float)world_z1; c.world_z2 = (float)world_z2;
    c.xy = keep(xy); c.R = keep(R); c.T = keep(T); c.Kinv = keep(Kinv);
    c.shape_code = keep(shape_code); c.gaze = keep(gaze); c.appea_code = keep(appea_code);
    c.t_rand = t_rand ? keep(*t_rand) : nullptr;
    c.z_edges = z_edges ? keep(*z_edges) : nullptr;
    c.edges_follow_T = (edges_follow_T && z_edges) ? 1 : 0;
    // view-direction option (include/gnr.h): skipped weight columns + per-ray bias of RGB_layer_1 per weight set
    TORCH_CHECK(vd_dims >= 0, "vd_dims must be >= 0");
    c.vd_dims = (int32_t)vd_dims;
    const OptTensor* rb[2] = {&ray_bias0, &ray_bias1};
    for (int s = 0; s < 2; ++s)
        if (*rb[s]) {
            want(**rb[s], "ray_bias", {p.B, p.n_r, hidden / 2});
            c.ray_bias[s] = keep(**rb[s]);
        }
    return p;
}

// 24 tensors of one MLPforNeRF in PARAM_ORDER (FeaExt_module_0..7 weight/bias, density, RGB_layer_0..2); Conv2d
// [out,in,1,1] memory == row-major [out,in].  Returns contiguous 2-D/1-D views kept alive in `keep`.
void fill_weights(const std::vector<Tensor>& params, const Problem& p, int64_t hidden, int64_t feat_nc, const char* tag,
                  std::vector<Tensor>& keep, GnrWeights* w) {
    TORCH_CHECK(params.size() == 24, tag, ": expected 24 parameter tensors, got ", params.size());
    const int64_t vp = 63 + p.c.shape_dims + p.c.gaze_dims, ap = p.c.vd_dims + p.c.appea_dims;
    auto mat = [&](int i, int64_t rows, int64_t cols) {
        Tensor t = params[i];
        check_f32(t, tag, p.dev);
        if (t.dim() == 4) {
            TORCH_CHECK(t.size(2) == 1 && t.size(3) == 1, tag, "[", i, "]: only 1x1 kernels");
            t = t.reshape({t.size(0), t.size(1)});
        }
        TORCH_CHECK(t.dim() == 2 && t.size(0) == rows && t.size(1) == cols, tag, "[", i, "] must have shape [", rows, ", ", cols,
                    "], got ", params[i].sizes());
        keep.push_back(t.contiguous());
        return (const float*)keep.back().data_ptr<float>();
    };
    auto vec = [&](int i, int64_t n) {
        const Tensor& t = params[i];
        check_f32(t, tag, p.dev);
        TORCH_CHECK(t.dim() == 1 && t.size(0) == n, tag, "[", i, "] must have shape [", n, "], got ", t.sizes());
        keep.push_back(t.contiguous());
        return (const float*)keep.back().data_ptr<float>();
    };
    for (int l = 0; l < 8; ++l) {
        const int64_t cin = l == 0 ? vp : (l == 5 ? hidden + vp : hidden);
        w->fea_w[l] = mat(2 * l, hidden, cin);
        w->fea_b[l] = vec(2 * l + 1, hidden);
    }
    w->density_w = mat(16, 1, hidden);
    w->density_b = vec(17, 1);
    w->rgb_w[0] = mat(18, hidden, hidden);          w->rgb_b[0] = vec(19, hidden);
    w->rgb_w[1] = mat(20, hidden / 2, hidden + ap); w->rgb_b[1] = vec(21, hidden / 2);
    w->rgb_w[2] = mat(22, feat_nc, hidden / 2);     w->rgb_b[2] = vec(23, feat_nc);
}

void check_rc(int rc) { TORCH_CHECK(rc == 0, gnr_last_error()); }

Tensor alloc_ws(size_t bytes, const c10::Device& dev) {
    return at::empty({(int64_t)(bytes < 256 ? 256 : bytes)}, at::TensorOptions().dtype(at::kByte).device(dev));
}

std::vector<Tensor> render_fwd(const Tensor& xy, const Tensor& R, const Tensor& T, const Tensor& Kinv, const Tensor& shape_code,
                               const Tensor& gaze, const Tensor& appea_code, const OptTensor& t_rand, const OptTensor& z_edges,
                               const std::vector<Tensor>& face, const std::vector<Tensor>& eyes, int64_t n_samples, double world_z1,
                               double world_z2, int64_t hidden, int64_t feat_nc, bool save, bool want_depth, bool want_weights,
                               bool bf16x3, bool edges_follow_T, const OptTensor& ws_in, bool weights_packed, int64_t vd_dims,
                               const OptTensor& ray_bias0, const OptTensor& ray_bias1) {
    Problem p = make_problem(xy, R, T, Kinv, shape_code, gaze, appea_code, t_rand, z_edges, n_samples, world_z1, world_z2, hidden,
                             feat_nc, edges_follow_T, vd_dims, ray_bias0, ray_bias1);
    const c10::DeviceGuard guard(p.dev);
    const int n_streams = eyes.empty() ? 1 : 2;
    const size_t nbytes = gnr_workspace_bytes(&p.c, n_streams, save ? GNR_WS_FWD_SAVE : GNR_WS_FWD);
    // a caller-owned workspace (render.PackedWeightCache) whose packed weights may still be current
    TORCH_CHECK(!weights_packed || (ws_in && !save), "weights_packed needs the caller's workspace and an inference call");
    if (ws_in) {
        TORCH_CHECK(ws_in->is_cuda() && ws_in->device() == p.dev && ws_in->scalar_type() == at::kByte && ws_in->is_contiguous() &&
                    (size_t)ws_in->numel() >= nbytes, "workspace must be a contiguous uint8 tensor of >= ", nbytes, " bytes on ", p.dev);
    }
    p.c.weights_packed = weights_packed ? 1 : 0;
    TORCH_CHECK(nbytes != 0, gnr_last_error());
    std::vector<Tensor> keep;
    GnrWeights w[2]{};
    fill_weights(face, p, hidden, feat_nc, "stream0", keep, &w[0]);
    if (n_streams > 1) fill_weights(eyes, p, hidden, feat_nc, "stream1", keep, &w[1]);
    Tensor ws = ws_in ? *ws_in : alloc_ws(nbytes, p.dev);
    const auto opt = at::TensorOptions().dtype(at::kFloat).device(p.dev);
    GnrOutputs out{};
    std::vector<Tensor> res;
    for (int s = 0; s < n_streams; ++s) {
        Tensor feat = at::empty({p.B, feat_nc, p.n_r}, opt), bga = at::empty({p.B, 1, p.n_r}, opt);
        out.feat[s] = feat.data_ptr<float>();
        out.bg_alpha[s] = bga.data_ptr<float>();
        res.push_back(feat);
        res.push_back(bga);
        if (want_depth) {
            Tensor d = at::empty({p.B, 1, p.n_r}, opt);
            out.depth[s] = d.data_ptr<float>();
            res.push_back(d);
        }
        if (want_weights) {
            Tensor wt = at::empty({p.B, 1, p.n_r, p.n_p}, opt);
            out.weights[s] = wt.data_ptr<float>();
            res.push_back(wt);
        }
    }
    void* stream = (void*)c10::hip::getCurrentHIPStream(p.dev.index()).stream();
    auto fn = bf16x3 ? gnr_fwd_bf16x3 : gnr_fwd;
    check_rc(fn(&p.c, &w[0], n_streams > 1 ? &w[1] : nullptr, &out, save ? 1 : 0, ws.data_ptr(), (size_t)ws.numel(), stream));
    res.push_back(ws);
    return res;
}

std::vector<Tensor> render_bwd(const Tensor& xy, const Tensor& R, const Tensor& T, const Tensor& Kinv, const Tensor& shape_code,
                               const Tensor& gaze, const Tensor& appea_code, const OptTensor& t_rand, const OptTensor& z_edges,
                               const std::vector<Tensor>& face, const std::vector<Tensor>& eyes,
                               const std::vector<OptTensor>& d_feat, const std::vector<OptTensor>& d_bg_alpha, const Tensor& saved_ws,
                               int64_t n_samples, double world_z1, double world_z2, int64_t hidden, int64_t feat_nc, bool bf16x3,
                               bool edges_follow_T, int64_t vd_dims, const OptTensor& ray_bias0, const OptTensor& ray_bias1) {
    Problem p = make_problem(xy, R, T, Kinv, shape_code, gaze, appea_code, t_rand, z_edges, n_samples, world_z1, world_z2, hidden,
                             feat_nc, edges_follow_T, vd_dims, ray_bias0, ray_bias1);
    const c10::DeviceGuard guard(p.dev);
    const int n_streams = eyes.empty() ? 1 : 2;
    TORCH_CHECK((int)d_feat.size() == n_streams && (int)d_bg_alpha.size() == n_streams, "one (d_feat, d_bg_alpha) pair per weight set");
    TORCH_CHECK(saved_ws.defined() && saved_ws.device() == p.dev && saved_ws.scalar_type() == at::kByte,
                "saved workspace must be the uint8 tensor render_fwd(save=True) returned");
    std::vector<Tensor> keep;
    GnrWeights w[2]{};
    fill_weights(face, p, hidden, feat_nc, "stream0", keep, &w[0]);
    if (n_streams > 1) fill_weights(eyes, p, hidden, feat_nc, "stream1", keep, &w[1]);
    GnrOutputGrads dout{};
    for (int s = 0; s < n_streams; ++s) {
        if (d_feat[s]) {
            check_f32(*d_feat[s], "d_feat", p.dev);
            TORCH_CHECK(d_feat[s]->sizes() == at::IntArrayRef({p.B, feat_nc, p.n_r}), "d_feat must be [B,feat_nc,N_r]");
            keep.push_back(d_feat[s]->contiguous());
            dout.feat[s] = keep.back().data_ptr<float>();
        }
        if (d_bg_alpha[s]) {
            check_f32(*d_bg_alpha[s], "d_bg_alpha", p.dev);
            TORCH_CHECK(d_bg_alpha[s]->sizes() == at::IntArrayRef({p.B, 1, p.n_r}), "d_bg_alpha must be [B,1,N_r]");
            keep.push_back(d_bg_alpha[s]->contiguous());
            dout.bg_alpha[s] = keep.back().data_ptr<float>();
        }
    }
    const auto opt = at::TensorOptions().dtype(at::kFloat).device(p.dev);
    std::vector<Tensor> res = {at::empty({p.B, 3, 3}, opt), at::empty({p.B, 3, 1}, opt), at::empty({p.B, p.c.shape_dims}, opt),
                               at::empty({p.B, p.c.gaze_dims}, opt), at::empty({p.B, p.c.appea_dims}, opt)};
    GnrInputGrads din{};
    din.R = res[0].data_ptr<float>(); din.T = res[1].data_ptr<float>(); din.shape_code = res[2].data_ptr<float>();
    din.gaze = res[3].data_ptr<float>(); din.appea_code = res[4].data_ptr<float>();
    GnrWeightGrads dw[2]{};
    for (int s = 0; s < n_streams; ++s) {
        const std::vector<Tensor>& src = s == 0 ? face : eyes;
        float* ptr[24];
        for (int i = 0; i < 24; ++i) {
            Tensor g = at::empty(src[i].sizes(), opt);           // same shape as the parameter ([out,in,1,1] stays 4-D)
            ptr[i] = g.data_ptr<float>();
            res.push_back(g);
        }
        for (int l = 0; l < 8; ++l) { dw[s].fea_w[l] = ptr[2 * l]; dw[s].fea_b[l] = ptr[2 * l + 1]; }
        dw[s].density_w = ptr[16]; dw[s].density_b = ptr[17];
        for (int l = 0; l < 3; ++l) { dw[s].rgb_w[l] = ptr[18 + 2 * l]; dw[s].rgb_b[l] = ptr[19 + 2 * l]; }
    }
    // d ray_bias per weight set, appended after the parameter gradients (an undefined tensor == None where absent)
    Tensor grb[2];
    for (int s = 0; s < n_streams; ++s)
        if (p.c.ray_bias[s]) {
            grb[s] = at::empty({p.B, p.n_r, hidden / 2}, opt);
            din.ray_bias[s] = grb[s].data_ptr<float>();
        }
    const size_t sbytes = gnr_workspace_bytes(&p.c, n_streams, GNR_WS_BWD);
    TORCH_CHECK(sbytes != 0, gnr_last_error());
    Tensor scratch = alloc_ws(sbytes, p.dev);
    void* stream = (void*)c10::hip::getCurrentHIPStream(p.dev.index()).stream();
    auto fn = bf16x3 ? gnr_bwd_bf16x3 : gnr_bwd;
    check_rc(fn(&p.c, &w[0], n_streams > 1 ? &w[1] : nullptr, &dout, &din, &dw[0], n_streams > 1 ? &dw[1] : nullptr,
                saved_ws.data_ptr(), (size_t)saved_ws.numel(), scratch.data_ptr(), (size_t)scratch.numel(), stream));
    res.push_back(grb[0]);
    res.push_back(grb[1]);
    return res;
}

int64_t saved_workspace_bytes(int64_t batch, int64_t n_rays, int64_t n_samples, int64_t hidden, int64_t feat_nc, int64_t n_streams) {
    GnrProblem c = GNR_INIT_PROBLEM;
    c.batch = (int32_t)batch; c.n_rays = (int32_t)n_rays; c.n_samples = (int32_t)n_samples; c.hidden = (int32_t)hidden;
    c.feat_nc = (int32_t)feat_nc;
    static const float dummy = 0.0f;                 // sizes only: pointers are checked for NULL, never dereferenced
    c.xy = c.R = c.T = c.Kinv = &dummy;
    const size_t n = gnr_workspace_bytes(&c, (int)n_streams, GNR_WS_FWD_SAVE);
    TORCH_CHECK(n != 0, gnr_last_error());
    return (int64_t)n;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "PyTorch-ROCm C++ binding of libgnr.so (GazeNeRF volumetric hot path on MI355X)";
    m.def("abi_version", []() { return gnr_abi_version(); });
#ifndef GNR_SOURCE_HASH
#define GNR_SOURCE_HASH "unknown"
#endif
    m.def("source_hash", []() { return std::string(GNR_SOURCE_HASH); }, "hash of csrc/ + include/gnr.h this binding was built from");
    m.def("build_info", []() { return std::string(gnr_build_info()); }, "gnr_build_info() of the libgnr.so it is linked to");
    m.def("render_fwd", &render_fwd, "gnr_fwd / gnr_fwd_bf16x3 on the current HIP stream");
    m.def("render_bwd", &render_bwd, "gnr_bwd / gnr_bwd_bf16x3 on the current HIP stream");
    m.def("saved_workspace_bytes", &saved_workspace_bytes, "bytes of saved activations for a training forward of this size");
}
