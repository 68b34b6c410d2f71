"""CPU: is a bf16x3 (3-term hi/lo split, fp32 accumulate) TRAINING step inside the reference's own fp32 noise?
Compares gradients of the oracle run in fp32 and of an emulated bf16x3 run (every dense layer's forward, dgrad and
wgrad products through the split) against the oracle in fp64.  Same problem as tools/gpu_grad_probe.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from gazenerf_amd import synth
from oracle import oracle as O

torch.set_num_threads(8)
n_rays, n_p, B = 40, 64, 2
sub = torch.arange(n_rays) * 53 % 4096
p = synth.synth_problem(64, batch=B, camera="9", seed=31, ray_subset=sub)
DS = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
face = synth.hash_mlp_params("face", seed=4, density_scale=DS)
eyes = synth.hash_mlp_params("eyes", seed=4, density_scale=DS)
t_rand = synth.synth_jitter(B, n_rays, n_p, seed=6)


def split(x):
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo


def mm3(a, b):
    ah, al = split(a); bh, bl = split(b)
    return ah @ bh + al @ bh + ah @ bl


class Lin3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):          # x [N,K], w [M,K]
        ctx.save_for_backward(x, w)
        return mm3(x, w.t()) + b

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        return mm3(gy, w), mm3(gy.t(), x), gy.sum(0)


def fc_emul(params, name, x):
    w = params[name + ".weight"]
    if name == "density_module":        # fp32 VALU dot in the kernel
        return F.conv2d(x, w, params[name + ".bias"])
    Bn, C, a, b_ = x.shape
    y = Lin3.apply(x.permute(0, 2, 3, 1).reshape(-1, C), w.reshape(w.shape[0], -1), params[name + ".bias"])
    return y.reshape(Bn, a, b_, -1).permute(0, 3, 1, 2)


def run(pp, f, e, tr):
    leaves = {k: pp[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fp = {k: v.clone().requires_grad_(True) for k, v in f.items()}
    ep = {k: v.clone().requires_grad_(True) for k, v in e.items()}
    out = O.render_two_stream(pp["xy"], leaves["R"], leaves["T"], pp["Kinv"], leaves["shape_code"], leaves["gaze"],
                              leaves["appea_code"], fp, ep, n_p, t_rand=tr)
    O.synthetic_loss(out).backward()
    g = {"d" + k: v.grad for k, v in leaves.items()}
    for tag, d in (("face", fp), ("eyes", ep)):
        for k, v in d.items():
            g[tag + "." + k] = v.grad
    return g, {k: out[k].detach() for k in ("feat_face", "bg_alpha_face", "feat_eyes", "bg_alpha_eyes")}


d64 = lambda d: {k: v.double() for k, v in d.items()}
ref, oref = run(d64(p), d64(face), d64(eyes), t_rand.double())
g32, o32 = run(p, face, eyes, t_rand)
orig = O._fc
O._fc = fc_emul
g3, o3 = run(p, face, eyes, t_rand)
O._fc = orig
for k in oref:
    print("fwd %-16s fp32 %.2e   bf16x3 %.2e" % (k, (o32[k].double() - oref[k]).abs().max(), (o3[k].double() - oref[k]).abs().max()))
w32 = w3 = 0.0
print("%-36s %10s %10s | %10s %10s" % ("tensor", "fp32 relL2", "x3 relL2", "fp32 max/s", "x3 max/s"))
for k in ref:
    r = ref[k]; s = max(float(r.abs().max()), 1e-30); n = max(float(r.norm()), 1e-30)
    a = [float((g[k].double() - r).norm()) / n for g in (g32, g3)]
    m = [float((g[k].double() - r).abs().max()) / s for g in (g32, g3)]
    w32 = max(w32, a[0]); w3 = max(w3, a[1])
    print("%-36s %10.2e %10.2e | %10.2e %10.2e" % (k, a[0], a[1], m[0], m[1]))
print("worst relL2: fp32 %.2e  bf16x3 %.2e" % (w32, w3))
