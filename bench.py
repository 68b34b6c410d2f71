#!/usr/bin/env python
"""bench.py -- rays/s of the GazeNeRF volumetric hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode fwd|fwdbwd] [--side S]

A "step" is one pass of the hot path (both MLP streams) over one batch of synthetic input:
the 512x512-ray, 64-samples-per-ray render BASELINE.json quotes the metric on (SURVEY.md 8(d)
cfg2b: featmap_size=512 -> 262 144 rays), inputs resident in HBM before the timed region.
``--mode fwdbwd`` adds the backward pass (ray micro-batches of ``--micro`` rays, gradients
accumulated, then all-reduced over RCCL when N>1).  With N>1 every rank renders its own image
(images are sharded, no data-path collective in forward): weak scaling.

Prints ONE JSON line on rank 0 with the driver's fields plus
  roofline     -- the fused MLP kernel against the gfx950 fp32-MFMA peak (157.3 TFLOP/s):
                  algorithmic FLOPs per launch / HIP-event duration of that kernel alone;
  cpu_baseline -- the CPU oracle (oracle/oracle.py, a port of the reference's PyTorch path, pinned
                  to it by tests/golden) timed on this host's cores on a bounded ray sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE_STREAM = 2 * 1351680          # folded count, fwd
PEAK_FP32_MFMA_TFLOPS = 157.3                 # MI355X_MICROARCH.md chip-level parameters
# bf16x3: three bf16 MFMAs per fp32-equivalent product -> a third of the 16x fp32 rate (dense bf16 peak / 3)
PEAK_BF16X3_TFLOPS = 16.0 * PEAK_FP32_MFMA_TFLOPS / 3.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=("fwd", "fwdbwd"), default=os.environ.get("GNR_BENCH_MODE", "fwdbwd"))
    ap.add_argument("--side", type=int, default=512, help="rays per image = side^2")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--micro", type=int, default=16384, help="rays per backward micro-batch")
    ap.add_argument("--precision", choices=("fp32", "bf16x3"), default=os.environ.get("GNR_BENCH_PRECISION", "fp32"),
                    help="fp32: exact fp32 MFMA everywhere; bf16x3: forward + dgrad chain on bf16 MFMA with a "
                         "3-term hi/lo split (fp32 accumulate)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra bf16x3 leg of an fp32 run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=512)
    return ap.parse_args()


def cpu_baseline(mode, n_rays, n_samples):
    """The oracle on this host's cores, bounded sample (SURVEY.md 8(d) CPU-baseline plan (ii))."""
    from gazenerf_amd import synth
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    face = synth.hash_mlp_params("face", seed=0, density_scale=50.0)
    eyes = synth.hash_mlp_params("eyes", seed=0, density_scale=50.0)

    def make(n):
        sub = (torch.arange(n) * 8) % 4096
        p = synth.synth_problem(64, batch=1, seed=5, ray_subset=sub)
        t_rand = synth.synth_jitter(1, n, n_samples, seed=5) if mode == "fwdbwd" else None

        def one():
            if mode == "fwd":
                with torch.no_grad():
                    O.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"],
                                        p["appea_code"], face, eyes, n_samples)
            else:
                fp = {k: v.clone().requires_grad_(True) for k, v in face.items()}
                ep = {k: v.clone().requires_grad_(True) for k, v in eyes.items()}
                leaves = {k: p[k].clone().requires_grad_(True)
                          for k in ("R", "T", "shape_code", "gaze", "appea_code")}
                out = O.render_two_stream(p["xy"], leaves["R"], leaves["T"], p["Kinv"], leaves["shape_code"],
                                          leaves["gaze"], leaves["appea_code"], fp, ep, n_samples, t_rand=t_rand)
                O.synthetic_loss(out).backward()
        return one

    # PyTorch's intra-op threading oversubscribes badly on many-core hosts for these shapes, so the
    # baseline uses the FASTEST thread count among {all cores, 64, 32, 16, 8} found on a 64-ray probe.
    probe = make(64)
    best_t, best_dt = None, None
    for t in sorted({c for c in (cores, 64, 32, 16, 8) if c <= cores}, reverse=True):
        torch.set_num_threads(t)
        probe()
        t0 = time.time()
        probe()
        dt = time.time() - t0
        if best_dt is None or dt < best_dt:
            best_t, best_dt = t, dt
    torch.set_num_threads(best_t)
    one = make(n_rays)
    one()                                   # warm-up
    times = []
    t_all = time.time()
    while len(times) < 3 or (time.time() - t_all < 8.0 and len(times) < 10):
        t0 = time.time()
        one()
        times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    # the reference pins itself to ONE thread (train.py / gazenerf_trainer: torch.set_num_threads(1)); SURVEY.md 8(d)
    # asks for that figure beside the all-core one: same workload on a 64-ray sample
    used = torch.get_num_threads()
    torch.set_num_threads(1)
    probe()
    t0 = time.time()
    probe()
    one_thread = 64 / (time.time() - t0)
    torch.set_num_threads(used)
    return {"value": n_rays / med, "unit": "rays/s", "cores": used, "kind": "port", "value_1_thread": one_thread,
            "sample": "%d rays x %d samples, both streams, %s, median of %d runs (%.2f s each) at the "
                      "fastest of the probed thread counts (%d of %d cores), PyTorch-CPU oracle pinned to "
                      "the reference by tests/golden" % (n_rays, n_samples, mode, len(times), med, best_t, cores)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the render op)")
    # GNR_BENCH_DEVICE: test-only override (several ranks on one device to exercise the launcher path)
    dev_index = int(os.environ.get("GNR_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # backend "nccl" is RCCL on ROCm; GNR_BENCH_BACKEND=gloo is a test-only override (two ranks on
        # one GPU cannot form an RCCL communicator)
        backend = os.environ.get("GNR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from gazenerf_amd import render, synth
    from gazenerf_amd.hiptime import KernelTimer
    from gazenerf_amd.parallel import GradAllReducer

    side, n_p = args.side, args.samples
    n_rays = side * side
    to = lambda d: {k: v.to(dev) for k, v in d.items()}
    p = to(synth.synth_problem(side, batch=1, camera=str(3 + rank), seed=100 + rank))
    face = to(synth.hash_mlp_params("face", seed=0, density_scale=50.0))
    eyes = to(synth.hash_mlp_params("eyes", seed=0, density_scale=50.0))
    plist = [face[k] for k in render.PARAM_ORDER] + [eyes[k] for k in render.PARAM_ORDER]
    leaves = [p[k] for k in ("R", "T", "shape_code", "gaze", "appea_code")]
    micro = min(args.micro, n_rays)
    t_rand = synth.synth_jitter(1, micro, n_p, seed=7).to(dev) if args.mode == "fwdbwd" else None
    reducer = GradAllReducer(plist, world) if (args.mode == "fwdbwd") else None
    timer = KernelTimer()
    aux_timer = KernelTimer(aux=True)
    kernel_ms, aux_ms = [], []

    def step(timed, precision):
        if args.mode == "fwd":
            with torch.no_grad():
                with timer:
                    render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"],
                                             p["appea_code"], face, eyes, n_samples=n_p, precision=precision)
                if timed:
                    kernel_ms.append(timer.elapsed_ms())
            return
        for t in plist + leaves:
            t.requires_grad_(True)
            t.grad = None
        for r0 in range(0, n_rays, micro):
            xy = p["xy"][:, :, r0:r0 + micro].contiguous()
            with timer:
                out = render.render_two_stream(xy, p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"],
                                               p["appea_code"], face, eyes, n_samples=n_p,
                                               t_rand=t_rand[:, :xy.shape[2]], precision=precision)
            if timed:
                kernel_ms.append(timer.elapsed_ms())
            loss = sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))
            with aux_timer:
                loss.backward()
            if timed:
                aux_ms.append(aux_timer.elapsed_ms())
        reducer.all_reduce()

    def timed_run(precision):
        """W untimed + K timed steps, barrier + synchronize on both sides, max over ranks (seconds)."""
        del kernel_ms[:], aux_ms[:]
        for _ in range(args.warmup):
            step(False, precision)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True, precision)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, list(kernel_ms), list(aux_ms)

    dt, kernel_ms_main, aux_ms_main = timed_run(args.precision)
    alt = None
    if args.precision == "fp32" and not args.no_alt:
        # second, separately timed leg: the same workload on the bf16x3 kernels (reported beside the
        # headline, never as it)
        alt = timed_run("bf16x3")
    kernel_ms, aux_ms = kernel_ms_main, aux_ms_main

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * n_rays * args.steps / dt
        # dominant kernel: the fused MLP forward kernel; one launch covers `rays_per_launch` rays
        rays_per_launch = n_rays if args.mode == "fwd" else micro
        flop_per_launch = rays_per_launch * n_p * 2 * FLOP_PER_SAMPLE_STREAM
        avg_ms = sum(kernel_ms) / max(1, len(kernel_ms))
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        # HBM-side bytes per launch: PMC FETCH_SIZE/WRITE_SIZE of this kernel, collected with rocprofv3
        # in separate passes and committed under profiles/ (counters cannot be read from inside this
        # process); scaled per ray to this launch size.
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r1_pmc_fwd.json" if args.mode == "fwd" else "r1_pmc_fwd_save.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                pj = json.load(f)
            traffic = pj["hbm_bytes_per_ray"] * rays_per_launch
            traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, per ray x rays per launch)" % os.path.basename(pmc)
        x3 = args.precision == "bf16x3"
        peak = PEAK_BF16X3_TFLOPS if x3 else PEAK_FP32_MFMA_TFLOPS
        kname = ("gnr::fwd3_kernel<%s>" if x3 else "gnr::fwd_kernel<%s>") % ("true" if args.mode == "fwdbwd" else "false")
        if x3:
            traffic, traffic_src = None, None
        res = {
            "metric": "rays/sec (512x512, 64 samples/ray) %s" % ("fwd+bwd" if args.mode == "fwdbwd" else "fwd"),
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3 (3-term hi/lo split on bf16 MFMA, f32 accumulate)" if x3 else "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d rays x %d samples/ray, two streams (face+eyes), %s, "
                                   "1 image per GPU%s" % ("cfg2b" if (side, n_p) == (512, 64) else "custom", side, side, n_p, args.mode,
                                                          ", %d-ray micro-batches" % micro if args.mode == "fwdbwd" else ""),
                       "rays_per_step_per_gpu": n_rays, "samples_per_ray": n_p,
                       "parallelism": "dp%d (images sharded)" % world, "precision": args.precision},
            "roofline": {"bound": "mfma", "kernel": kname, "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s",
                         "peak_basis": ("dense bf16 MFMA peak (16 x 157.3) / 3 terms; achieved counts the "
                                        "fp32-equivalent algorithmic FLOPs") if x3 else "fp32 MFMA peak",
                         "frac": achieved / peak, "frac_of_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "flop_per_launch": flop_per_launch, "avg_launch_ms": avg_ms,
                         "launches_timed": len(kernel_ms),
                         # whole step against the same peak: algorithmic FLOPs of the step (fwd, or
                         # 3x fwd for fwd+bwd: dgrad + wgrad) / wall time of the step
                         "step_achieved": (3 if args.mode == "fwdbwd" else 1) * n_rays * n_p * 2
                                          * FLOP_PER_SAMPLE_STREAM / (ms * 1e-3) / 1e12,
                         "step_frac": (3 if args.mode == "fwdbwd" else 1) * n_rays * n_p * 2
                                      * FLOP_PER_SAMPLE_STREAM / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "step_frac_basis": "fp32 MFMA peak"},
        }
        if aux_ms:
            # HBM-bound compositing pass (CalcRayColor backward): algorithmic bytes per sample =
            # 288 saved features + sigma_raw + delta read, w_i + dL/dsigma written
            m = micro * n_p
            nbytes = m * (288 * 4 + 8 + 8) + micro * (288 * 4 + 4 + 8)
            a = sum(aux_ms) / len(aux_ms)
            gbs = nbytes / (a * 1e-3) / 1e9
            hbm_traffic = None
            pmc_cb = os.path.join(ROOT, "profiles", "r1_pmc_comp_bwd.json")
            if os.path.exists(pmc_cb):
                with open(pmc_cb) as f:
                    hbm_traffic = json.load(f)["hbm_bytes_per_ray"] * micro
            res["roofline_hbm"] = {"bound": "hbm", "kernel": "gnr::comp_bwd_kernel", "achieved": gbs, "peak": 8000.0,
                                   "unit": "GB/s", "frac": gbs / 8000.0, "bytes_per_launch": nbytes,
                                   "avg_launch_ms": a, "launches_timed": len(aux_ms), "traffic": hbm_traffic,
                                   "traffic_source": "profiles/r1_pmc_comp_bwd.json (rocprofv3 --pmc, per ray x rays per launch)"}
        if alt is not None:
            adt, akm, _ = alt
            a_avg = sum(akm) / max(1, len(akm))
            a_ach = flop_per_launch / (a_avg * 1e-3) / 1e12 if a_avg > 0 else 0.0
            res["bf16x3"] = {
                "note": "same workload, same timing protocol, dense layers (forward, dgrad chain, weight-gradient "
                        "GEMMs) on bf16 MFMA with a 3-term hi/lo split, fp32 accumulate; feature map within "
                        "1e-4 of the reference (<= 6e-6 on the fixtures); against fp64 as close to exact as the "
                        "reference's own fp32 run; gradients inside the reference's fp32-vs-fp64 noise "
                        "(tests/test_parity_gpu.py, DESIGN.md section 4)",
                "value": world * n_rays * args.steps / adt, "unit": "rays/s", "ms_per_step": adt / args.steps * 1e3,
                "speedup_vs_fp32": dt / adt,
                "roofline": {"bound": "mfma", "kernel": "gnr::fwd3_kernel<%s>" % ("true" if args.mode == "fwdbwd" else "false"),
                             "achieved": a_ach, "peak": PEAK_BF16X3_TFLOPS, "unit": "TFLOP/s",
                             "peak_basis": "dense bf16 MFMA peak (16 x 157.3) / 3 terms; fp32-equivalent FLOPs",
                             "frac": a_ach / PEAK_BF16X3_TFLOPS, "frac_of_fp32_mfma_peak": a_ach / PEAK_FP32_MFMA_TFLOPS,
                             "avg_launch_ms": a_avg, "launches_timed": len(akm)}}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.mode, args.cpu_rays, n_p)
        print(json.dumps(res), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
