"""GPU: anatomy of an outlier draw of the bf16x3 gradient gate (tests/test_parity_gpu.py NOISE_GATE).

Round 5, 16 draws: in draw 13 the bf16x3 kernels' gradients of eyes.FeaExt_module_0..4 sit 90-215 x the reference's fp32 noise
above the fp64 oracle -- on tensors where that noise is tiny in this draw (rel-L2 1e-5: no mask flips in the reference's fp32
run).  Is that ONE flipped ReLU mask (a pre-activation inside the 3-term product's ~1e-4 noise of zero: the legitimate,
discrete noise of another arithmetic) or a kernel defect?  A single flip at (sample s, layer l, channel c) changes dpre_l[s, c]
only: the bias-gradient difference of layer l is concentrated in ONE channel, the weight-gradient difference in ONE row, and
the layers below see a rank-1 update; layers above l see nothing.  This prints exactly that, bf16x3 kernels against the fp32
kernels on the same problem.      python tests/diagnostics/gpu_x3_outlier.py [draw]
"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("tp", os.path.join(ROOT, "tests", "test_parity_gpu.py"))
tp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tp)
from gazenerf_amd import _lib, render           # noqa: E402
from oracle import oracle as O                  # noqa: E402

draw = int(sys.argv[1]) if len(sys.argv) > 1 else 13
dev = torch.device("cuda:0")
p, face, eyes, t_rand = tp._grad_problem(False, draw)
g = {}
for pr in ("fp32", "bf16x3"):
    g[pr] = tp._all_grads(tp._to(p, dev), tp._to(face, dev), tp._to(eyes, dev), t_rand.to(dev),
                          lambda xy, R, T, K, s, ga, a, f, e, tr, pr=pr: render.render_two_stream(
                              xy, R, T, K, s, ga, a, f, e, n_samples=64, t_rand=tr, precision=pr))
d64 = lambda d: {k: v.double() for k, v in d.items()}
ofn = lambda xy, R, T, K, s, ga, a, f, e, tr: O.render_two_stream(xy, R, T, K, s, ga, a, f, e, 64, t_rand=tr)
exact = tp._all_grads(d64(p), d64(face), d64(eyes), t_rand.double(), ofn)
print("build: %s; draw %d" % (_lib.build_info(), draw))
print("%-30s %10s %10s | bias-gradient difference x3 - fp32: share of its norm in the largest channel(s)" % ("tensor", "fp32 err", "x3 err"))
for tag in ("face", "eyes"):
    for l in list(range(8)) + ["r0", "r1"]:
        name = "%s.%s" % (tag, "FeaExt_module_%d" % l if isinstance(l, int) else {"r0": "RGB_layer_0", "r1": "RGB_layer_1"}[l])
        r = exact[name + ".bias"]
        n = float(r.norm())
        e32 = float((g["fp32"][name + ".bias"].cpu().double() - r).norm()) / n
        e3 = float((g["bf16x3"][name + ".bias"].cpu().double() - r).norm()) / n
        D = (g["bf16x3"][name + ".bias"] - g["fp32"][name + ".bias"]).cpu().double()
        Dn = max(float(D.norm()), 1e-300)
        top = torch.topk(D.abs(), 3)
        DW = (g["bf16x3"][name + ".weight"] - g["fp32"][name + ".weight"]).cpu().double().reshape(D.numel(), -1)
        rows = DW.norm(dim=1)
        rtop = torch.topk(rows, 2)
        print("%-30s %10.2e %10.2e | bias: ch %3d %.3f, ch %3d %.3f, ch %3d %.3f | weight rows: row %3d %.3f, row %3d %.3f" % (
            name, e32, e3, int(top.indices[0]), float(top.values[0]) / Dn, int(top.indices[1]), float(top.values[1]) / Dn,
            int(top.indices[2]), float(top.values[2]) / Dn,
            int(rtop.indices[0]), float(rtop.values[0]) / max(float(DW.norm()), 1e-300), int(rtop.indices[1]),
            float(rtop.values[1]) / max(float(DW.norm()), 1e-300)))
