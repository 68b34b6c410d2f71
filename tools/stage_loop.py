#!/usr/bin/env python
"""GPU: ONE stage of the fp32 / bf16x3 training step in a loop, for power / clock measurements per stage
(tools/smi_sample.py around it) -- the bench interleaves the three stages every ~120 ms, too fast for the SMI's samples.

    python tools/stage_loop.py fwd|bwd [--seconds 12] [--rays 16384] [--precision fp32|bf16x3]

fwd: gnr_fwd with save_for_backward (fwd16_kernel<true> + combine) over and over in one workspace;
bwd: gnr_bwd over and over on one saved workspace (comp_bwd + dgrad chain + weight gradients, both streams);
alt: forward, backward, forward, ... as a training loop issues them.
Prints the stage's HIP-event time and the in-kernel clock probe."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gazenerf_amd import render, synth                                   # noqa: E402
from gazenerf_amd.hiptime import ClockProbe, StageTimer                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stage", choices=("fwd", "bwd", "alt"))
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--rays", type=int, default=16384)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--nosave", action="store_true", help="fwd: the inference kernel (fwd16_kernel<false>), no dumps")
    ap.add_argument("--gap-us", type=float, default=0.0, help="fwd: synchronize and leave the GPU idle this long before every forward")
    ap.add_argument("--pre", choices=("none", "gemm", "copy"), default="none",
                    help="a full-chip torch kernel right in front of every forward (fp32 GEMM ~0.15 ms each / 0.5 GB copy ~0.2 ms each)")
    ap.add_argument("--pre-reps", type=int, default=2, help="how many of them (e.g. 250 GEMMs ~ 40 ms of hipBLASLt MFMA work in place of a backward)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    x3 = a.precision == "bf16x3"
    p = {k: v.to(dev) for k, v in synth.synth_problem(512, batch=1, camera="3", seed=100, ray_subset=torch.arange(a.rays)).items()}
    face = {k: v.to(dev) for k, v in synth.hash_mlp_params("face", seed=0, density_scale=50.0).items()}
    eyes = {k: v.to(dev) for k, v in synth.hash_mlp_params("eyes", seed=0, density_scale=50.0).items()}
    t_rand = synth.synth_jitter(1, a.rays, 64, seed=7).to(dev)
    prob = render._Problem(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], 64, 2.5, -3.5,
                           t_rand, None, 384, 258)
    streams = [render._prep_params(render.params_to_list(d), 384, prob.vp, prob.c.appea_dims, 258, "s") for d in (face, eyes)]
    os.environ["GNR_BINDING"] = "ctypes"
    res, ws = render._run_forward(prob, streams, True, False, False, x3)
    gout = [(torch.randn_like(r[0]) * 1e-3, torch.randn_like(r[1]) * 1e-3) for r in res]
    keys = ("fwd_mlp",) if a.stage == "fwd" else (("dgrad", "wgrad", "comp_bwd") if a.stage == "bwd" else ("fwd_mlp", "dgrad", "wgrad"))
    timers = {k: StageTimer(k, pool=4096) for k in keys}
    for t in timers.values():
        t.reset(True)

    ga = torch.randn(2048, 2048, device=dev)
    gb = torch.randn(2048, 4096, device=dev)
    ca = torch.empty(128 << 20, device=dev)
    cb = torch.empty(128 << 20, device=dev)

    def one():
        for t in timers.values():
            t.arm()
        if a.gap_us > 0:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            while (time.perf_counter() - t1) * 1e6 < a.gap_us:
                pass
        if a.pre == "gemm":
            for _ in range(a.pre_reps):
                torch.mm(ga, gb)
        elif a.pre == "copy":
            for _ in range(max(1, a.pre_reps // 2)):
                cb.copy_(ca)
        if a.stage in ("fwd", "alt"):
            # (the caching allocator hands the same workspace back every time)
            lib_res, ws2 = render._run_forward(prob, streams, not a.nosave, False, False, x3)
        if a.stage in ("bwd", "alt"):
            render._run_backward(prob, streams, gout, ws2 if a.stage == "alt" else ws, x3)

    one()
    torch.cuda.synchronize()
    n = 0
    with ClockProbe(dev) as probe:
        t0 = time.time()
        while time.time() - t0 < a.seconds:
            one()
            n += 1
            if n % 8 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    clocks = probe.mhz()
    out = []
    for k, t in timers.items():
        ms = t.collect()
        if ms:
            ms = ms[1:]
            out.append("%s %.3f ms (n=%d) %s MHz" % (k, sum(ms) / len(ms), len(ms), round(clocks.get(k, 0))))
    if a.nosave:
        a.stage += " (inference kernel)"
    extra = (" gap %.0f us" % a.gap_us if a.gap_us else "") + (" pre=%s x %d" % (a.pre, a.pre_reps) if a.pre != "none" else "")
    print("stage_loop %s%s %s %d rays: %d calls in %.1f s; %s" % (a.stage, extra, a.precision, a.rays, n, a.seconds, "; ".join(out)), flush=True)


if __name__ == "__main__":
    main()
