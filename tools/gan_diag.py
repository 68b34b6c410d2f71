"""Diagnostic: gradients of the PatchGAN discriminator loss against an fp64 CPU run -- the shipped module (F.conv2d: oneDNN on
the CPU, MIOpen on the MI355X) and an im2col + GEMM restatement of its convolutions, CPU and GPU, plus the smallest
|pre-activation| of the four LeakyReLUs.  A gradient 0.7 % off with logits exact to 2e-6 is ONE LeakyReLU element on the other
slope, not an inexact convolution.  tools/gan_diag.sh runs it under MIOpen's solver switches.   python tools/gan_diag.py [seed]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gazenerf_amd import gan as G       # noqa: E402


def _conv_gemm(x, conv):
    k, s_, p = conv.kernel_size, conv.stride, conv.padding
    B, _, H, W = x.shape
    ho, wo = (H + 2 * p[0] - k[0]) // s_[0] + 1, (W + 2 * p[1] - k[1]) // s_[1] + 1
    out = torch.matmul(conv.weight.reshape(conv.out_channels, -1), F.unfold(x, kernel_size=k, padding=p, stride=s_))
    if conv.bias is not None:
        out = out + conv.bias.view(1, -1, 1)
    return out.view(B, conv.out_channels, ho, wo)


class GemmPatchGAN(G.PatchGAN):
    def forward(self, x):
        x = self.act(_conv_gemm(x, self.conv1))
        x = self.act(self.norm1(_conv_gemm(x, self.conv2)))
        x = self.act(self.norm2(_conv_gemm(x, self.conv3)))
        x = self.act(self.norm3(_conv_gemm(x, self.conv4)))
        return _conv_gemm(x, self.conv5)


SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 21


def run(cls, dev, dtype=torch.float32):
    d = cls(3, 8)
    d.load_state_dict(G.hash_patchgan_state(seed=3, ndf=8))
    d.train()
    d = d.to(dev).to(dtype)
    case = {k: v.to(dev).to(dtype) for k, v in G.synth_gan_case(seed=SEED).items()}
    real, fake = d(case["real_img"]), d(case["fake_img"])
    G.discriminator_loss(real, fake).backward()
    return {k: q.grad.double().cpu() for k, q in d.named_parameters()}, real.detach().double().cpu()


ref64, l64 = run(G.PatchGAN, "cpu", torch.float64)
rel = lambda a, b: float((a - b).norm() / b.norm())
cases = {"cpu fp32, F.conv2d (shipped)": (G.PatchGAN, "cpu"), "cpu fp32, im2col+GEMM": (GemmPatchGAN, "cpu")}
if torch.cuda.is_available():
    cases.update({"gpu fp32, F.conv2d (shipped)": (G.PatchGAN, "cuda:0"), "gpu fp32, im2col+GEMM": (GemmPatchGAN, "cuda:0")})
print("input seed %d: smallest |LeakyReLU input| in fp64 %.2e" % (SEED, G.min_abs_preactivation(G.PatchGAN(3, 8), G.hash_patchgan_state(seed=3, ndf=8), G.synth_gan_case(seed=SEED))))
for name, (cls, dev) in cases.items():
    gr, lg = run(cls, dev)
    print("%-34s logits max-abs %.1e | rel-L2 of d loss / d" % (name, float((lg - l64).abs().max())),
          "  ".join("%s %.1e" % (k, rel(gr[k], ref64[k])) for k in ("conv1.weight", "conv1.bias", "conv3.weight", "conv4.weight", "conv5.weight")))
