"""GPU: the distribution behind test_backward_error_follows_the_reference_fp32_noise_distribution.

For N independent draws of the unstable gradient problem (opaque head + train jitter: ReLU masks flip under fp32 noise) and
every one of the 53 gradient tensors: rel-L2 error against the oracle in fp64 of
    ref32   the oracle's own fp32 autograd,
    null    the oracle's fp32 autograd on a channel-permuted copy of the network (same function, other summation order: a
            second draw of the reference's own noise, no HIP code involved),
    fp32 / bf16x3   the HIP backward.
Prints per tensor the ratios err_x / err_ref32 over the draws (median, max) -- what "same distribution, another draw" looks
like in numbers -- and, last, the verdict of the test's own gate on these draws (tests/test_parity_gpu.py NOISE_GATE).
    python tests/diagnostics/grad_noise_draws.py [n_draws] > profiles/r5_grad_noise_draws.txt"""
import importlib.util
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("tp", os.path.join(ROOT, "tests", "test_parity_gpu.py"))
tp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tp)

n = int(sys.argv[1]) if len(sys.argv) > 1 else tp.N_NOISE_DRAWS
dev = torch.device("cuda:0")
ratios = {}
for draw in range(n):
    for k, (e_ref, e) in tp._noise_errors(False, draw, tp.PRECISIONS, dev, with_null=True).items():
        for who in ("null", "fp32", "bf16x3"):
            eps = tp.GRAD_EPS.get(who, tp.GRAD_EPS["fp32"])
            ratios.setdefault(k, {}).setdefault(who, []).append(max(e[who] - eps, 0.0) / max(e_ref, 1e-30))
            ratios[k].setdefault("abs_" + who, []).append(e[who])
        ratios[k].setdefault("ref", []).append(e_ref)
from gazenerf_amd import _lib
print("build: %s" % _lib.build_info())
print("%d draws; per tensor: median / second largest / max of err_x / err_ref32 (both against the fp64 oracle)" % n)
print("%-34s %10s | %-21s | %-21s | %-21s" % ("tensor", "med e_ref", "null", "fp32 kernels", "bf16x3 kernels"))
worst = {w: [0.0, 0.0] for w in ("null", "fp32", "bf16x3")}
for k, d in ratios.items():
    cells = []
    for who in ("null", "fp32", "bf16x3"):
        med, mx = statistics.median(d[who]), max(d[who])
        worst[who][0] = max(worst[who][0], med)
        worst[who][1] = max(worst[who][1], mx)
        cells.append("%5.2f / %5.2f / %6.2f" % (med, sorted(d[who])[-2] if len(d[who]) > 1 else mx, mx))
    print("%-34s %10.2e | %s | %s | %s" % (k, statistics.median(d["ref"]), *cells))
print("worst over tensors (median, max):", {w: ("%.2f" % v[0], "%.2f" % v[1]) for w, v in worst.items()})
for who in ("null", "fp32", "bf16x3"):
    allr = sorted(r for d in ratios.values() for r in d[who])
    q = lambda f: allr[min(len(allr) - 1, int(f * len(allr)))]
    print("%-7s all (tensor, draw) ratios: median %.2f  p90 %.2f  p99 %.2f  max %.2f   share > 3: %.3f" % (
        who, q(0.5), q(0.9), q(0.99), allr[-1], sum(r > 3 for r in allr) / len(allr)))
by_who = {who: {k: d[who] for k, d in ratios.items()} for who in ("null", "fp32", "bf16x3")}
by_abs = {who: {k: d["abs_" + who] for k, d in ratios.items()} for who in ("null", "fp32", "bf16x3")}
bad = tp.noise_gate_failures(by_who, by_abs)
print("gate (tests/test_parity_gpu.py NOISE_GATE = %s): %s" % (tp.NOISE_GATE, "PASS" if not bad else "FAIL"))
for b in bad:
    print("  " + b)
