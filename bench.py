#!/usr/bin/env python
"""bench.py -- rays/s of the GazeNeRF volumetric hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2b|cfg4] [--mode fwd|fwdbwd] [--precision ...]

``--gpus N`` with N > 1 launches N ranks itself (``python -m torch.distributed.run --nproc-per-node N``, one rank per
GPU, RCCL) unless it is already running under such a launcher (WORLD_SIZE set; then WORLD_SIZE must equal N).

cfg2b (default, the configuration BASELINE.json quotes the metric on): a "step" is one pass of the hot path (both MLP
streams) over the 512x512-ray, 64-samples-per-ray render of one image per GPU (featmap_size=512 ->
262 144 rays), inputs resident in HBM before the timed region.  ``--mode fwdbwd`` adds the backward pass: the loss of
SURVEY.md 8(a) A8 is a sum over rays, so the image is processed as ``--micro``-ray micro-batches (forward-with-save,
loss, backward each; gradients accumulate; ONE all-reduce per step when N > 1) -- nothing is recomputed.  N > 1:
every rank renders its own image (images are sharded, no data-path collective in forward): weak scaling.
``--scaling strong`` (the reference's B=1 render loop, utils/render_utils.py:199-219): ONE image, its
rays sharded over the ranks in contiguous row blocks (``parallel.shard_rays``), value = rays of that one image per
second; ``--gather`` adds the all_gather that hands rank 0 the full feature maps (``parallel.gather_rays``).

cfg4: the reference's training step -- B=2 images per rank at featmap_size=64 through the WHOLE
network (hot path -> merge -> upsampler x4 -> image loss), backward, all-reduce of all 5 015 714 trainable floats
(both MLPs + NeuralRenderer incl. bg_featmap), Adam step.  Reference: trainer/gazenerf_trainer.py:478-534.

Prints ONE JSON line on rank 0 with the driver's fields plus
  roofline     -- the dominant kernel against the gfx950 fp32-MFMA peak (157.3 TFLOP/s): algorithmic FLOPs per launch
                  / HIP-event duration of that kernel alone, recorded on the launch stream inside the library;
  stages       -- the same for every stage of the step (forward-with-save, dgrad chain, weight-gradient GEMMs,
                  compositing backward), each with its own achieved / peak / frac and share of the step;
  allreduce    -- bytes, buckets and separately timed milliseconds of the gradient exchange (N > 1);
  cpu_baseline -- the CPU oracle (oracle/oracle.py, a port of the reference's PyTorch path, pinned to it by
                  tests/golden) timed on this host's cores (rank 0, N = 1 only) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE_STREAM = 2 * 1351680          # folded count, fwd
PEAK_FP32_MFMA_TFLOPS = 157.3                 # MI355X_MICROARCH.md chip-level parameters
# bf16x3: three bf16 MFMAs per fp32-equivalent product -> a third of the 16x fp32 rate (dense bf16 peak / 3)
PEAK_BF16X3_TFLOPS = 16.0 * PEAK_FP32_MFMA_TFLOPS / 3.0
HBM_PEAK_GBS = 8000.0
HBM_STREAM_GBS = 6290.0                       # measured float4 copy on MI355X (same table): what a streaming kernel can reach
PEAK_CLOCK_MHZ = 2400.0                       # the clock both MFMA peaks are quoted at
FEAT_NC, FEAT_PAD = 258, 288                  # opt.featmap_nc; the saved activation layout's padded feature axis (csrc/gnr_internal.h)
# the weight-gradient stage: its main kernel instance first (the 384^2 layers), then what else runs inside the bracket
WGRAD_KERNEL = {False: "gnr::wgrad2w_kernel<false, false, 6, 3> + its other instances (<.., 3, 3>: 96 x 192, <.., 6, 1>: 192 x 64) + gnr::wgrad_reduce_batch_kernel (one launch for the 13 split-K reductions)",
                True: "gnr::wgrad3_tr_kernel<3, false> + its other instances + gnr::wgrad_reduce_batch_kernel"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=("cfg2b", "cfg4"), default=os.environ.get("GNR_BENCH_CONFIG", "cfg2b"))
    ap.add_argument("--mode", choices=("fwd", "fwdbwd"), default=os.environ.get("GNR_BENCH_MODE", "fwdbwd"))
    ap.add_argument("--side", type=int, default=None,
                    help="cfg2b: rays per image = side^2 (default 512).  cfg4: feature-map side (default 64 = the reference's; "
                         "the image is 8 x side); any other value is a TEST-ONLY size (launcher rehearsals) and the line says so")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--micro", type=int, default=32768,
                    help="cfg2b fwdbwd: rays per tile of the tiled training entry point (forward-with-save, loss, backward per tile). "
                         "32768 rays keep 66 GB of activations + 33 GB of backward scratch resident (of 288 GB); rounds 1-3 used "
                         "16384: the larger tile halves the per-launch ramps and tails (+1 % on the step, profiles/r4_micro_sweep.txt)")
    ap.add_argument("--precision", choices=("fp32", "bf16x3"), default=os.environ.get("GNR_BENCH_PRECISION", "fp32"),
                    help="fp32: exact fp32 MFMA everywhere; bf16x3: dense layers on bf16 MFMA with a "
                         "3-term hi/lo split (fp32 accumulate)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra bf16x3 leg of an fp32 run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=110.0, help="seconds of CPU-baseline work (wall cap)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("GNR_BENCH_SCALING", "weak"),
                    help="cfg2b: weak = one image per GPU; strong = ONE image, rays sharded over the GPUs")
    ap.add_argument("--gather", action="store_true", help="strong scaling, fwd: all_gather the feature maps to every rank inside the step")
    ap.add_argument("--ref-loss", action="store_true",
                    help="cfg4: the reference trainer's DEFAULT loss switches (train.py:38-42: use_vgg_loss, use_l1_loss) -- adds the three "
                         "VGG-perceptual terms (gazenerf_amd.perceptual; randomly initialised extractor: same FLOPs as the ImageNet one)")
    ap.add_argument("--gan-loss", action="store_true", help="cfg4: + use_patch_gan_loss (discriminator update, then the generator term)")
    ap.add_argument("--no-one-call", action="store_true", help="skip the extra timing of the one-call (in-op tiled) training path")
    ap.add_argument("--pg-timeout", type=float, default=180.0, help="seconds before a stuck rendezvous / collective aborts")
    ap.add_argument("--sync", choices=("auto", "spin", "yield", "blocking"), default=os.environ.get("GNR_BENCH_SYNC", "auto"),
                    help="hipSetDeviceFlags(hipDeviceSchedule*) before the device context exists: how a host thread waits when the HIP "
                         "runtime makes it wait (a full queue, a synchronize).  `blocking` parks the thread on an interrupt instead of "
                         "polling.  auto = the runtime's default at ONE rank, `blocking` at N > 1 (ranks share the host's cores; "
                         "host.cpu_share and host.sync in the line)")
    ap.add_argument("--force-dist", action="store_true", default=os.environ.get("GNR_BENCH_FORCE_DIST", "") == "1",
                    help="--gpus 1: form a ONE-rank RCCL process group anyway and run the gradient exchange through it "
                         "(loads librccl, creates a communicator, exercises the stream hand-off on a single GPU)")
    args = ap.parse_args()
    if args.side is None:
        args.side = 64 if args.config == "cfg4" else 512
    return args


# ------------------------------------------------------------------------------------------------ launcher
def self_launch(args):
    """--gpus N without a launcher: become `torch.distributed.run --nproc-per-node N bench.py <same args>`."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------------------------------------ CPU baseline
def available_cores():
    """Cores this process may really use: the scheduler affinity mask, capped by the cgroup CPU quota (a container on a
    256-thread host with `cpu.max = 1600000 100000` has 16; running 256 OpenMP threads there is 50x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    basis = "affinity mask: %d" % n
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            with open(path) as f:
                quota, period = parse(f.read())
            if period is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().strip()
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / float(period) + 0.999))
                if q < n:
                    n, basis = q, "cgroup CPU quota %s / %s = %d of %d logical CPUs" % (quota, period, q, os.cpu_count() or 0)
            break
        except (OSError, ValueError):
            continue
    return n, basis


def cpu_baseline(mode, n_samples, budget_s):
    """The oracle on this host's cores (SURVEY.md 8(d) CPU-baseline plan (ii)): BASELINE cfg2a -- one 64x64-ray image,
    64 samples, both streams, what the reference renders per 512x512 image -- at every core the process may use and
    at one thread (1024 rays), median of 3 where the wall cap allows.  The workload shrinks (rays, then repeats) to stay
    inside ``budget_s``: the full 4096-ray image forward+backward takes ~17 s per run at 16 cores."""
    import torch
    from gazenerf_amd import synth
    from oracle import oracle as O
    cores, cores_basis = available_cores()
    face = synth.hash_mlp_params("face", seed=0, density_scale=50.0)
    eyes = synth.hash_mlp_params("eyes", seed=0, density_scale=50.0)

    def make(n):
        sub = (torch.arange(n) * 8 + (torch.arange(n) // 512)) % 4096 if n < 4096 else None
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

    def timed(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    def leg(threads, budget, n_want):
        """-> (rays/s, rays, runs, seconds per run).  Probe on 128 rays, then ``n_want`` rays (halved until three runs fit
        the budget; at least one run), median of up to three runs."""
        torch.set_num_threads(threads)
        t_start = time.perf_counter()
        probe = make(128)
        probe()                                       # warm-up: thread pool, oneDNN primitives, allocator
        per_ray = timed(probe) / 128
        n = n_want
        while n > 128 and 3.2 * n * per_ray > budget - (time.perf_counter() - t_start):
            n //= 2
        one = make(n)
        times = [timed(one)]
        while len(times) < 3 and (time.perf_counter() - t_start) + 1.1 * times[-1] < budget:
            times.append(timed(one))
        times.sort()
        med = times[len(times) // 2]
        return n / med, n, len(times), med

    # cfg2a (4096 rays) x 3 at every core, 1024 rays x 3 at one thread (the reference pins itself to 1)
    v_all, n_all, r_all, s_all = leg(cores, 0.52 * budget_s, 4096)
    res = {"value": v_all, "unit": "rays/s", "cores": cores, "kind": "port", "host_logical_cpus": os.cpu_count(),
           "cores_basis": cores_basis,
           "sample": "cfg2a subset: %d of 4096 rays x %d samples, both streams, %s; median of %d run(s) of %.2f s at "
                     "torch.set_num_threads(%d) = every core this process may use (%s); PyTorch-CPU oracle pinned to the "
                     "reference by tests/golden; wall cap %.0f s" % (n_all, n_samples, mode, r_all, s_all, cores, cores_basis, budget_s)}
    # the reference pins itself to ONE thread (train.py / gazenerf_trainer: torch.set_num_threads(1))
    v1, n1, r1, s1 = leg(1, 0.48 * budget_s, 1024)
    res["value_1_thread"] = v1
    res["sample_1_thread"] = "%d rays, %d run(s) of %.2f s" % (n1, r1, s1)
    torch.set_num_threads(cores)
    return res



# ------------------------------------------------------------------------------------------------ host side of a rank
def parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_rank_to_cpus(torch, dev_index, local_rank, world):
    """One rank per GPU shares the host with world - 1 others: give every rank its own contiguous block of CPUs, taken
    from the ones local to ITS GPU (sysfs `local_cpulist` of the device's PCI function) where the node says which those
    are, otherwise from the process's affinity mask.  Without it the launch threads of eight ranks migrate over the whole
    mask (and the cgroup of a bench box grants 16 cores for all of them).  -> what was done, for the JSON line."""
    if world <= 1 or not hasattr(os, "sched_setaffinity") or os.environ.get("GNR_BENCH_NO_AFFINITY") == "1":
        return None
    mask = sorted(os.sched_getaffinity(0))
    cand, basis = mask, "affinity mask (%d CPUs)" % len(mask)
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        with open("/sys/bus/pci/devices/%s/local_cpulist" % bdf) as f:
            local = sorted(parse_cpulist(f.read()) & set(mask))
        if len(local) >= world:
            cand, basis = local, "local_cpulist of GPU %s within the affinity mask (%d CPUs)" % (bdf, len(local))
    except (OSError, AttributeError, ValueError):
        pass
    i = local_rank % world
    lo, hi = len(cand) * i // world, len(cand) * (i + 1) // world
    mine = cand[lo:hi] or cand
    try:
        os.sched_setaffinity(0, mine)
    except OSError as e:
        return {"pinned": False, "error": str(e)}
    torch.set_num_threads(max(1, min(len(mine), 4)))        # the step is launch code: a rank needs no OpenMP team of 256
    return {"pinned": True, "cpus": "%d-%d" % (mine[0], mine[-1]) if mine == list(range(mine[0], mine[-1] + 1)) else mine,
            "n_cpus": len(mine), "basis": basis}


def print_topology(world):
    """Rank 0, N > 1: the xGMI topology of the node into stderr (stdout carries the one JSON line)."""
    import shutil
    exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
    if not exe:
        sys.stderr.write("bench.py: rocm-smi not found: no topology printed\n")
        return
    try:
        r = subprocess.run([exe, "--showtopo"], capture_output=True, text=True, timeout=60)
        sys.stderr.write("bench.py: rocm-smi --showtopo (world size %d)\n%s\n" % (world, r.stdout[-6000:]))
    except (OSError, subprocess.SubprocessError) as e:
        sys.stderr.write("bench.py: rocm-smi --showtopo failed: %s\n" % e)
    sys.stderr.flush()


# ------------------------------------------------------------------------------------------------ helpers
def pmc_traffic(kernel, rays_per_launch, n_samples):
    """HBM-side bytes per launch of `kernel` from a committed rocprofv3 --pmc capture (separate FETCH_SIZE /
    WRITE_SIZE passes, tools/pmc_capture.py) -- only if it was captured at THIS launch size; counters cannot be
    read from inside the process."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, "no capture committed"
    with open(path) as f:
        table = json.load(f)
    for e in table.get("kernels", []):
        if e["kernel"] == kernel and e["rays_per_launch"] == rays_per_launch and e["samples_per_ray"] == n_samples:
            return e["hbm_bytes_per_launch"], "profiles/pmc_traffic.json: %s" % e.get("source", "rocprofv3 --pmc")
    return None, "profiles/pmc_traffic.json has no capture of %s at %d rays x %d samples per launch" % (
        kernel, rays_per_launch, n_samples)


def mean(xs):
    return sum(xs) / len(xs) if xs else 0.0


def chain_suffix(x3):
    """Kernel-name infix of the dense-chain kernels: bf16x3 -> "3"; fp32 -> "16" (v_mfma_f32_16x16x4_f32, two waves per
    SIMD: gnr_chain16.h)."""
    return "3" if x3 else "16"


_RESULT_FD = None


def guard_stdout():
    """stdout carries ONE JSON line.  Libraries write there too -- RCCL prints its NCCL_DEBUG=VERSION banner to the process's
    stdout (whatever NCCL_DEBUG_FILE says: profiles/r5_bench_forced_rccl_ws1.json before this guard), buffered, i.e. BEHIND the
    line -- so file descriptor 1 is pointed at stderr for the life of the process and the line goes to a duplicate of the
    original descriptor."""
    global _RESULT_FD
    if _RESULT_FD is None:
        try:
            sys.stdout.flush()
            fd = os.dup(1)
            os.dup2(2, 1)
            _RESULT_FD = fd
        except OSError:          # no usable stderr / stdout descriptor: print the line the ordinary way
            _RESULT_FD = None


def emit_result(res):
    line = (json.dumps(res) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
        return
    view = memoryview(line)
    while view:
        view = view[os.write(_RESULT_FD, view):]


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    guard_stdout()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher formed WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the render op)")
    # GNR_BENCH_DEVICE: test-only override (several ranks on one device to exercise the launcher path)
    dev_index = int(os.environ.get("GNR_BENCH_DEVICE", local_rank))
    # pre-flight: one line instead of a HIP "invalid device ordinal" from somewhere inside rank k
    n_dev = torch.cuda.device_count()
    if dev_index >= n_dev:
        raise SystemExit("bench.py: rank %d needs GPU %d but this node exposes %d device(s) (--gpus %d, one rank per GPU)"
                         % (rank, dev_index, n_dev, args.gpus))
    if args.scaling == "strong" and args.config != "cfg2b":
        raise SystemExit("bench.py: --scaling strong shards the rays of ONE cfg2b image; cfg4 shards images (weak)")
    sync_rc = None
    # auto: ONE rank keeps the runtime's default (it polls: the lowest wake-up latency, and the host has nothing else to do);
    # N > 1 ranks share the host's cores -- and a bench box's cgroup -- with each other and with RCCL's proxy threads, and a cfg2b
    # rank that only launches otherwise burns 1.6-1.9 cores polling (host.cpu_share): they park instead
    sync_mode = args.sync if args.sync != "auto" else ("blocking" if world > 1 else "auto")
    if sync_mode != "auto":
        # before the primary context of the device exists (torch creates it at the first allocation / set_device)
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        flag = {"spin": 0x1, "yield": 0x2, "blocking": 0x4}[sync_mode]       # hipDeviceScheduleSpin / Yield / BlockingSync
        sync_rc = [int(hip.hipSetDevice(dev_index)), int(hip.hipSetDeviceFlags(flag))]
        if sync_rc != [0, 0]:
            sys.stderr.write("bench.py: rank %d: hipSetDevice / hipSetDeviceFlags(%s) returned %s: the runtime's default stays\n" % (rank, sync_mode, sync_rc))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # pre-flight, every rank, stderr: what the device has free BEFORE this rank allocates (hipMemGetInfo) -- several ranks on
    # one device (the test-only rehearsals) or a neighbour's leftovers show up here instead of as an out-of-memory in step 1
    free_b, total_b = torch.cuda.mem_get_info(dev)
    sys.stderr.write("bench.py: rank %d of %d on GPU %d: %.1f GiB free of %.1f GiB (hipMemGetInfo), pid %d\n"
                     % (rank, world, dev_index, free_b / 2**30, total_b / 2**30, os.getpid()))
    sys.stderr.flush()
    affinity = pin_rank_to_cpus(torch, dev_index, local_rank, world)
    dist = None
    backend = None
    try:
        if world > 1 or args.force_dist:
            import datetime

            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if world == 1 and "MASTER_PORT" not in os.environ:      # --force-dist without a launcher: a one-rank rendezvous
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC only on this driver (RCCL needs it)
            # backend "nccl" is RCCL on ROCm; GNR_BENCH_BACKEND=gloo is a test-only override (two ranks on
            # one GPU cannot form an RCCL communicator)
            backend = os.environ.get("GNR_BENCH_BACKEND", "nccl")
            if rank == 0 and world > 1:
                print_topology(world)
            if backend == "nccl" and rank == 0:
                # RCCL's version banner (and nothing else) from rank 0; it lands in stderr (guard_stdout)
                os.environ.setdefault("NCCL_DEBUG", "VERSION")
            timeout = datetime.timedelta(seconds=args.pg_timeout)      # a rank that never arrives fails the job, not hangs it
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=timeout)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
            # first collective right here: a communicator that cannot form (IPC, topology, a dead peer) fails in
            # pre-flight with the rank's message, before any workload has been built
            probe = torch.ones(1, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(probe)
            if int(probe.item()) != world:
                raise RuntimeError("pre-flight all-reduce returned %s, expected %d" % (probe.item(), world))
        ctx = dict(args=args, rank=rank, world=world, dev=dev, dist=dist, backend=backend, torch=torch, host={})
        res = run_cfg4(ctx) if args.config == "cfg4" else run_cfg2b(ctx)
        # every rank, stderr: what this rank held at its peak (several ranks on one device -- the test-only rehearsals -- add up)
        free_e, _ = torch.cuda.mem_get_info(dev)
        sys.stderr.write("bench.py: rank %d of %d: peak %.1f GiB allocated / %.1f GiB reserved by this rank; %.1f GiB free on GPU %d now\n"
                         % (rank, world, torch.cuda.max_memory_allocated(dev) / 2**30, torch.cuda.max_memory_reserved(dev) / 2**30,
                            free_e / 2**30, dev_index))
        sys.stderr.flush()
        if rank == 0:
            from gazenerf_amd import _lib
            res["build"] = _lib.build_info()          # gnr_build_info(): source hash (checked against the tree at load), flags
            res["host"] = dict(ctx["host"], cpu_affinity=affinity, cores_available="%d (%s)" % available_cores(),
                               sync={"asked": args.sync, "mode": sync_mode, "hipSetDevice_hipSetDeviceFlags_rc": sync_rc})
            if dist:
                res["distributed"] = {"backend": backend, "world_size_formed": dist.get_world_size(),
                                      "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None,
                                      "devices_visible": n_dev, "forced_at_world_size_1": bool(world == 1),
                                      "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(args.mode if args.config == "cfg2b" else "fwdbwd", 64 if args.config == "cfg4" else args.samples,
                                                   args.cpu_budget)
            emit_result(res)
    except BaseException as e:            # noqa: BLE001 -- every failure of a rank must end the job with ITS message
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        import traceback
        sys.stderr.write("bench.py: rank %d of %d failed: %s: %s\n" % (rank, world, type(e).__name__, e))
        traceback.print_exc()
        sys.stderr.flush()
        # no destroy_process_group here: it would wait for peers that are stuck in a collective with us; a non-zero exit
        # makes torch.distributed.run tear the other ranks down
        os._exit(1)
    if dist:
        dist.destroy_process_group()


def timed_loop(ctx, step, reset):
    """W untimed + K timed steps, barrier + synchronize on both sides, max over ranks -> seconds."""
    args, dist, torch, dev = ctx["args"], ctx["dist"], ctx["torch"], ctx["dev"]
    reset(False)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    reset(True)
    t0 = time.perf_counter()
    enq = 0.0
    c0 = time.process_time()
    for _ in range(args.steps):
        e0 = time.perf_counter()
        step()                               # enqueues only: no host synchronisation inside a step
        enq += time.perf_counter() - e0
    cpu = time.process_time() - c0           # user + system CPU time of this process (all its threads) over the steps' host code
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt, enq, cpu], device=dev, dtype=torch.float64)
        if ctx["backend"] != "nccl":
            tt = tt.cpu()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, enq, cpu = float(tt[0].item()), float(tt[1].item()), float(tt[2].item())
    # host_enqueue_ms_per_step (max over ranks): `ms` = wall time of the step's host code up to its last launch.  Well below
    # ms_per_step = the host runs ahead of the GPU.  Equal to it = EITHER the rank is launch-bound OR the HIP queue is full and
    # the launches block on the GPU (what a healthy GPU-bound step looks like once the queue has filled) -- `cpu_ms` tells them
    # apart: the CPU time this process really spent over the steps, ALL its threads (the autograd engine's included).
    # cpu_share = cpu_ms / ms_per_step.  Measured (rounds 5-6): cfg4 0.20 -- the host has slack; cfg2b 1.6-1.9 -- NOT slack: the
    # step's host code is blocked for 80-86 % of the step and the process still burns more than one and a half cores, i.e. the
    # blocked threads poll.  `--sync blocking` asks the runtime to park them instead; host.sync in the
    # line says what was asked, cpu_share what it bought.
    ctx["host"].setdefault("host_enqueue_ms_per_step", []).append(
        {"leg": ctx.get("leg", "main"), "ms": enq / args.steps * 1e3, "cpu_ms": cpu / args.steps * 1e3,
         "ms_per_step": dt / args.steps * 1e3, "share": enq / dt if dt > 0 else None,
         "cpu_share": cpu / dt if dt > 0 else None})
    return dt


class AllReduceClock:
    """torch.cuda.Event pairs around the gradient exchange, read after the final synchronize."""

    def __init__(self, torch):
        self.torch, self.pairs, self.keep = torch, [], False

    def reset(self, keep):
        self.pairs, self.keep = [], keep

    def __call__(self, fn):
        if not self.keep:
            return fn()
        a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.pairs.append((a, b))

    def ms(self):
        return [a.elapsed_time(b) for a, b in self.pairs]


def dist_info(ctx, reducer, clock):
    dist = ctx["dist"]
    if not dist:
        return None
    ms = clock.ms()
    return {"backend": "rccl (torch.distributed nccl)" if ctx["backend"] == "nccl" else ctx["backend"],
            "world_size_formed": dist.get_world_size(), "bytes": reducer.bytes_per_step,
            "floats": reducer.bytes_per_step // 4, "buckets": reducer.n_buckets,
            "ms": mean(ms), "calls_timed": len(ms),
            "note": "ms = stream time from the launch of the first not-yet-launched bucket to the last write-back "
                    "(flatten + all-reduce + average + copy back), rank 0"}


# ------------------------------------------------------------------------------------------------ cfg2b
def run_cfg2b(ctx):
    args, rank, world, dev, dist, torch = (ctx[k] for k in ("args", "rank", "world", "dev", "dist", "torch"))
    from gazenerf_amd import render, synth
    from gazenerf_amd.hiptime import ClockProbe, StageTimer
    from gazenerf_amd.parallel import GradAllReducer

    side, n_p = args.side, args.samples
    n_rays = side * side
    fwdbwd = args.mode == "fwdbwd"
    to = lambda d: {k: v.to(dev) for k, v in d.items()}
    strong = args.scaling == "strong"
    # weak: every rank its own image (camera / codes by rank); strong: the SAME image on every rank, rays sharded
    p = to(synth.synth_problem(side, batch=1, camera="3" if strong else str(3 + rank), seed=100 if strong else 100 + rank))
    if strong:
        from gazenerf_amd import parallel
        p["xy"] = parallel.shard_rays(p["xy"], rank, world)          # contiguous row blocks of the side x side grid
    n_local = p["xy"].shape[2]
    face = to(synth.hash_mlp_params("face", seed=0, density_scale=50.0))
    eyes = to(synth.hash_mlp_params("eyes", seed=0, density_scale=50.0))
    plist = [face[k] for k in render.PARAM_ORDER] + [eyes[k] for k in render.PARAM_ORDER]
    leaves = [p[k] for k in ("R", "T", "shape_code", "gaze", "appea_code")]
    micro = min(args.micro, n_local)
    t_rand = synth.synth_jitter(1, n_local, n_p, seed=7).to(dev) if fwdbwd else None      # [1, n_local, n_p + 1]
    reducer = GradAllReducer([plist[:24], plist[24:]], world, force_collective=bool(dist)) if fwdbwd else None
    clock = AllReduceClock(torch)
    timers = {k: StageTimer(k, pool=128) for k in (("fwd_mlp", "dgrad", "comp_bwd", "wgrad") if fwdbwd else ("fwd_mlp",))}

    def reset(keep):
        for t in timers.values():
            t.reset(keep)
        clock.reset(keep)

    def step(precision):
        if not fwdbwd:
            with torch.no_grad():
                timers["fwd_mlp"].arm()
                out = render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"],
                                               p["appea_code"], face, eyes, n_samples=n_p, precision=precision)
                if strong and args.gather and world > 1:
                    # one rank (here: every rank) wants the whole map: all_gather of the [C, N_r / world] slices
                    for k in ("feat_face", "bg_alpha_face", "feat_eyes", "bg_alpha_eyes"):
                        parallel.gather_rays(out[k], n_rays, world)
            return
        for t in plist + leaves:
            t.requires_grad_(True)
            t.grad = None
        # the library's tiled training entry point (round 4): per ray tile forward-with-save -> the caller's per-ray
        # loss -> backward, 3/3 of the FLOPs; the stage timers are re-armed for every tile from inside the loss closure
        for t in timers.values():
            t.arm()

        def tile_loss(out, sl):
            # called between a tile's forward and its backward: hand the library fresh event pairs for this tile's
            # backward stages (tile 0's were armed above) and for the next tile's forward
            if sl.start > 0:
                for k in ("dgrad", "comp_bwd", "wgrad"):
                    timers[k].arm()
            if sl.stop < n_local:
                timers["fwd_mlp"].arm()
            return sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))

        render.render_two_stream_tiled(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"],
                                       face, eyes, loss_fn=tile_loss, n_samples=n_p, ray_tile=micro,
                                       t_rand=t_rand, precision=precision)
        if dist:
            clock(reducer.all_reduce)

    def leg(precision):
        ctx["leg"] = precision
        dt = timed_loop(ctx, lambda: step(precision), reset)
        stage_ms = {k: t.collect() for k, t in timers.items()}
        info = dist_info(ctx, reducer, clock) if fwdbwd else None
        # one more, untimed, step with the shader-clock probe armed: the clock the power management sustains under
        # each stage's kernels (the MFMA peaks are quoted at 2.4 GHz)
        reset(False)
        with ClockProbe(dev) as probe:
            step(precision)
            torch.cuda.synchronize()
        return dt, stage_ms, info, probe.mhz()

    dt, stage_ms, ar, clocks = leg(args.precision)
    one_call = None
    if fwdbwd and world == 1 and not args.no_one_call and n_local > micro:
        # What a caller of GazeNeRFNetAMD(featmap_size=512) gets: ONE render_two_stream call for the whole image.  Its
        # saved activations (540 GB) exceed the workspace budget, so the op tiles the rays itself: an inference forward
        # of the whole image, then per tile forward-with-save + backward = 4/3 of the FLOPs of the micro-batched
        # loop above (which keeps each micro-batch's activations until its own backward).  Timed, never the headline.
        import ctypes

        from gazenerf_amd import _lib
        q = _lib.GnrProblem()
        q.batch, q.n_rays, q.n_samples, q.hidden, q.feat_nc = 1, n_local, n_p, 384, 258
        q.xy = q.R = q.T = q.Kinv = 1                      # sizes only: checked for NULL, never dereferenced
        saved = _lib.load().gnr_workspace_bytes(ctypes.byref(q), 2, _lib.WS_FWD_SAVE)
        tiled = saved > render.DEFAULT_WS_BUDGET           # the op's own rule (render.plan_ray_tiles)

        def call():
            for t in plist + leaves:
                t.requires_grad_(True)
                t.grad = None
            out = render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"],
                                           face, eyes, n_samples=n_p, t_rand=t_rand, precision=args.precision)
            sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes")).backward()

        call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        oc = (time.perf_counter() - t0) / 2
        one_call = {"ms_per_step": oc * 1e3, "value": n_local / oc, "unit": "rays/s", "calls_timed": 2,
                    "ms_per_step_tiled_entry": dt / args.steps * 1e3,
                    "tiled_entry": "render_two_stream_tiled(loss_fn=...) -- the timed step of this bench IS that entry point: per "
                                   "ray tile forward-with-save, the caller's per-ray loss, backward; 3/3 of the FLOPs",
                    "tiled_in_op": bool(tiled), "saved_activation_bytes_untiled": int(saved),
                    "workspace_budget_bytes": int(render.DEFAULT_WS_BUDGET),
                    "flop_factor_vs_step": 4.0 / 3.0 if tiled else 1.0,
                    "achieved_tflops_executed": (4 if tiled else 3) * n_local * n_p * 2 * FLOP_PER_SAMPLE_STREAM / oc / 1e12,
                    "note": "one render_two_stream call + backward for the whole %dx%d-ray image; when the saved activations exceed "
                            "the workspace budget the op tiles the rays itself (inference forward once, then forward-with-save + "
                            "backward per tile = 4/3 of the step's FLOPs)" % (side, side)}
    alt = None
    if args.precision == "fp32" and not args.no_alt:
        alt = leg("bf16x3")            # second, separately timed leg on the bf16x3 kernels (never the headline)
    if rank != 0:
        return None

    def describe(precision, dt, stage_ms, clocks):
        x3 = precision == "bf16x3"
        peak = PEAK_BF16X3_TFLOPS if x3 else PEAK_FP32_MFMA_TFLOPS
        ms = dt / args.steps * 1e3
        launches_per_step = (n_local + micro - 1) // micro if fwdbwd else 1
        # a ragged last tile (micro does not divide the image) makes the timed launches unequal: the stage rates are then
        # total FLOPs / total time, i.e. priced on the MEAN launch (rays_per_launch is fractional in that case only)
        rays_per_launch = n_local / launches_per_step if fwdbwd else n_local
        if rays_per_launch == int(rays_per_launch):
            rays_per_launch = int(rays_per_launch)
        m = rays_per_launch * n_p                                  # samples per launch (per stream)
        flop_2s = m * 2 * FLOP_PER_SAMPLE_STREAM                   # both streams, one pass
        flop_1s = m * FLOP_PER_SAMPLE_STREAM
        sfx = chain_suffix(x3)
        save = "true" if fwdbwd else "false"
        defs = [("fwd_mlp", "forward%s: march + encode + 2-stream MLP + composite" % (" with activation save" if fwdbwd else ""),
                 "gnr::fwd%s_kernel<%s>" % (sfx, save), flop_2s, 1)]
        if fwdbwd:
            defs += [("dgrad", "dgrad chain (timed on the first stream; runs once per stream)",
                      "gnr::bwd%s_chain_kernel" % sfx, flop_1s, 2),
                     ("wgrad", "weight-gradient GEMMs + split reductions (12 layer GEMMs, timed on the first stream; once per stream)",
                      WGRAD_KERNEL[x3], flop_1s, 2)]
        stages = []
        for key, what, kernel, flop, per_step_mult in defs:
            a = mean(stage_ms.get(key, []))
            ach = flop / (a * 1e-3) / 1e12 if a > 0 else 0.0
            stages.append({"stage": key, "what": what, "kernel": kernel, "bound": "mfma", "achieved": ach, "peak": peak,
                           "unit": "TFLOP/s", "frac": ach / peak, "avg_ms": a, "launches_timed": len(stage_ms.get(key, [])),
                           "flop_per_launch": flop,
                           "share_of_step": a * per_step_mult * launches_per_step / ms if ms > 0 else 0.0})
            if clocks.get(key):
                # in-kernel s_memtime / s_memrealtime of every 64th workgroup (gnr_set_clock_probe), separate untimed step
                stages[-1]["clock_mhz"] = clocks[key]
                stages[-1]["frac_at_clock"] = ach / (peak * clocks[key] / PEAK_CLOCK_MHZ)
        if fwdbwd and stage_ms.get("comp_bwd"):
            # HBM-bound compositing pass (CalcRayColor backward, utils/model_utils.py:498-534).  ALGORITHMIC bytes (SURVEY.md
            # 8(d)): per sample the 258 saved features + sigma_raw + delta read, w_i + dL/dsigma written; per ray 258 upstream
            # gradients + 3 scalars.  The layout pads the feature axis to 288 (18 tiles of 16): the padded figure is
            # reported beside it, and `frac` is the algorithmic one.
            per = lambda ch: m * (ch * 4 + 8 + 8) + rays_per_launch * (ch * 4 + 4 + 8)
            # three channel counts, all derived from the problem: the algorithm's feat_nc; what comp_bwd_kernel
            # READS since round 5 (the 16-channel groups that hold real channels); the saved layout's FEAT_PAD
            feat_nc, feat_read, feat_layout = FEAT_NC, (FEAT_NC + 15) & ~15, FEAT_PAD
            nbytes, nbytes_read, nbytes_pad = per(feat_nc), per(feat_read), per(feat_layout)
            a = mean(stage_ms["comp_bwd"])
            gbs, gbs_pad = nbytes / (a * 1e-3) / 1e9, nbytes_pad / (a * 1e-3) / 1e9
            tr, src = pmc_traffic("gnr::comp_bwd_kernel", rays_per_launch, n_p)
            stages.append({"stage": "comp_bwd", "what": "compositing backward (timed on the first stream; once per stream)",
                           "kernel": "gnr::comp_bwd_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "peak_measured_stream": HBM_STREAM_GBS,
                           "frac_of_measured_stream": gbs / HBM_STREAM_GBS,
                           "achieved_incl_padding": gbs_pad, "frac_incl_padding": gbs_pad / HBM_PEAK_GBS,
                           "peak_basis": "frac: algorithmic bytes (%d feature channels) against the 8.0 TB/s HBM3E spec; "
                                         "frac_of_measured_stream: against the 6.29 TB/s a float4 copy achieves on this chip "
                                         "(MI355X_MICROARCH.md chip-level table); bytes_per_launch_read: the %d channels the kernel "
                                         "reads; *_incl_padding: the %d-channel layout's bytes" % (feat_nc, feat_read, feat_layout),
                           "avg_ms": a, "launches_timed": len(stage_ms["comp_bwd"]),
                           "bytes_per_launch": nbytes, "bytes_per_launch_read": nbytes_read, "bytes_per_launch_incl_padding": nbytes_pad,
                           "channels": {"algorithmic": feat_nc, "read_by_kernel": feat_read, "saved_layout": feat_layout},
                           "traffic_over_read": tr / nbytes_read if tr else None, "traffic": tr,
                           "traffic_over_algorithmic": tr / nbytes if tr else None, "traffic_source": src,
                           "share_of_step": a * 2 * launches_per_step / ms})
        step_flop = (3 if fwdbwd else 1) * n_local * n_p * 2 * FLOP_PER_SAMPLE_STREAM      # this rank's share
        step_ach = step_flop / (ms * 1e-3) / 1e12
        return ms, stages, step_ach, rays_per_launch

    ms, stages, step_ach, rays_per_launch = describe(args.precision, dt, stage_ms, clocks)
    x3 = args.precision == "bf16x3"
    peak = PEAK_BF16X3_TFLOPS if x3 else PEAK_FP32_MFMA_TFLOPS
    # the dominant stage = the largest share of the step among the MFMA-bound stages
    dom = max((s for s in stages if s["bound"] == "mfma"), key=lambda s: s["share_of_step"])
    traffic, traffic_src = pmc_traffic(dom["kernel"].split(" + ")[0], rays_per_launch, n_p)
    res = {
        "metric": "rays/sec (512x512, 64 samples/ray) %s" % ("fwd+bwd" if fwdbwd else "fwd"),
        "value": (1 if strong else world) * n_rays * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "bf16x3 (3-term hi/lo split on bf16 MFMA, f32 accumulate)" if x3 else "f32", "data": "synthetic",
        "config": {"workload": "%s: %dx%d rays x %d samples/ray, two streams (face+eyes), %s, %s%s" % (
                       "cfg2b" if (side, n_p) == (512, 64) else "custom", side, side, n_p, args.mode,
                       "ONE image, its rays sharded over the GPUs in contiguous row blocks" if strong else "1 image per GPU",
                       ", %d-ray micro-batches (the loss is a sum over rays: nothing recomputed)" % micro if fwdbwd else ""),
                   "rays_per_step_per_gpu": n_local, "samples_per_ray": n_p,
                   "parallelism": (("rays%d (one image, ray blocks sharded; %s)" % (world, "one gradient all-reduce per step" if fwdbwd else
                                    ("all_gather of the four output maps inside the step" if args.gather else "no collective: every rank keeps its slice")))
                                   if strong else
                                   ("dp%d (images sharded; one gradient all-reduce per step, no overlap: the gradients "
                                    "of both MLPs leave the last micro-batch's backward together)" % world if fwdbwd
                                    else "dp%d (images sharded, no collective)" % world)),
                   "precision": args.precision},
        "roofline": {"bound": "mfma", "kernel": dom["kernel"], "stage": dom["stage"], "achieved": dom["achieved"], "peak": peak,
                     "unit": "TFLOP/s",
                     "peak_basis": ("dense bf16 MFMA peak (16 x 157.3) / 3 terms; achieved counts the fp32-equivalent "
                                    "algorithmic FLOPs") if x3 else "fp32 MFMA peak",
                     "frac": dom["frac"], "frac_of_fp32_mfma_peak": dom["achieved"] / PEAK_FP32_MFMA_TFLOPS,
                     "clock_mhz": dom.get("clock_mhz"), "frac_at_clock": dom.get("frac_at_clock"),
                     "traffic": traffic, "traffic_source": traffic_src,
                     "flop_per_launch": dom["flop_per_launch"], "avg_launch_ms": dom["avg_ms"],
                     "launches_timed": dom["launches_timed"], "share_of_step": dom["share_of_step"],
                     # whole step against the same peak: algorithmic FLOPs of the step (fwd, or 3x fwd for
                     # fwd+bwd: dgrad + wgrad) / wall time of the step
                     "step_achieved": step_ach, "step_frac": step_ach / PEAK_FP32_MFMA_TFLOPS,
                     "step_frac_basis": "fp32 MFMA peak"},
        "stages": stages,
    }
    hbm = [s for s in stages if s["bound"] == "hbm"]
    if hbm:
        res["roofline_hbm"] = hbm[0]
    if ar:
        res["allreduce"] = ar
    if one_call is not None:
        res["one_call"] = one_call
    if alt is not None:
        adt, ams, _, aclk = alt
        a_ms, a_stages, a_step, _ = describe("bf16x3", adt, ams, aclk)
        a_dom = max((s for s in a_stages if s["bound"] == "mfma"), key=lambda s: s["share_of_step"])
        res["bf16x3"] = {
            "note": "same workload, same timing protocol, dense layers (forward, dgrad chain, weight-gradient "
                    "GEMMs) on bf16 MFMA with a 3-term hi/lo split, fp32 accumulate; feature map within "
                    "1e-4 of the reference (<= 6e-6 on the fixtures); against fp64 as close to exact as the "
                    "reference's own fp32 run; gradients inside the reference's fp32-vs-fp64 noise "
                    "(tests/test_parity_gpu.py, DESIGN.md section 4)",
            "value": (1 if strong else world) * n_rays * args.steps / adt, "unit": "rays/s", "ms_per_step": a_ms,
            "speedup_vs_fp32": dt / adt,
            "roofline": {"bound": "mfma", "kernel": a_dom["kernel"], "achieved": a_dom["achieved"], "peak": PEAK_BF16X3_TFLOPS,
                         "unit": "TFLOP/s", "peak_basis": "dense bf16 MFMA peak (16 x 157.3) / 3 terms; fp32-equivalent FLOPs",
                         "frac": a_dom["frac"], "frac_of_fp32_mfma_peak": a_dom["achieved"] / PEAK_FP32_MFMA_TFLOPS,
                         "avg_launch_ms": a_dom["avg_ms"], "launches_timed": a_dom["launches_timed"],
                         "step_achieved": a_step, "step_frac_of_bf16x3_peak": a_step / PEAK_BF16X3_TFLOPS},
            "stages": a_stages}
    return res


# ------------------------------------------------------------------------------------------------ cfg4
def run_cfg4(ctx):
    """The reference's training step on B=2 images per rank (trainer/gazenerf_trainer.py:478-534): whole network
    forward in "train" mode, image loss (the terms that need no pretrained network), backward, all-reduce of every
    trainable parameter's gradient, Adam."""
    args, rank, world, dev, dist, torch = (ctx[k] for k in ("args", "rank", "world", "dev", "dist", "torch"))
    from gazenerf_amd import GazeNeRFNetAMD, losses, synth
    from gazenerf_amd.hiptime import ClockProbe, StageTimer
    from gazenerf_amd.parallel import GradAllReducer

    B, S, n_p = 2, args.side, 64
    if S < 16 or S & (S - 1):
        raise SystemExit("bench.py: --config cfg4 --side %d: the feature-map side must be a power of two >= 16" % S)
    I = 8 * S                                     # three pixel-shuffle blocks, as 64 -> 512 in the reference
    test_size = S != 64                           # any side but the reference's 64 is a TEST-ONLY size (launcher rehearsals)
    n_rays = S * S
    torch.manual_seed(1234)                       # identical initial parameters on every rank
    net = GazeNeRFNetAMD(featmap_size=S, pred_img_size=I, precision=args.precision).to(dev)
    p = {k: v.to(dev) for k, v in synth.synth_problem(S, batch=B, camera=str(3 + 2 * rank), seed=100 + rank).items()}
    t_rand = synth.synth_jitter(B, n_rays, n_p, seed=7 + rank).to(dev)
    g = torch.Generator(device="cpu").manual_seed(50 + rank)
    gt = torch.rand(B, 3, I, I, generator=g).to(dev)
    yy, xx = torch.meshgrid(torch.arange(float(I)), torch.arange(float(I)), indexing="ij")
    k = I / 512.0
    disk = lambda cx, cy, r: (((xx - cx * k) ** 2 + (yy - cy * k) ** 2) <= (r * k) ** 2).float().expand(B, 1, -1, -1).to(dev)
    face_mask, left_eye, right_eye = disk(256, 256, 200), disk(190, 220, 28), disk(322, 220, 28)
    full_eye = torch.clamp(left_eye + right_eye, max=1.0)
    zeros = lambda n: torch.zeros(B, n, device=dev)
    opt_codes = {"bg": None, "iden": zeros(100), "expr": zeros(79), "appea": zeros(127)}
    params = list(net.parameters())
    n_train = sum(q.numel() for q in params)
    nr = [q for n, q in net.named_parameters() if n.startswith("neural_render.")]
    face = [q for n, q in net.named_parameters() if n.startswith("fg_CD_predictor_face.")]
    eyes = [q for n, q in net.named_parameters() if n.startswith("fg_CD_predictor_eyes.")]
    assert len(nr) + len(face) + len(eyes) == len(params)
    # buckets in the order the backward produces them: NeuralRenderer first (its all-reduce flies during the hot
    # path's backward), then the two MLPs
    reducer = GradAllReducer([nr, face, eyes], world, force_collective=bool(dist))
    reducer.arm_overlap()
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.999), fused=True)
    vgg = gan = None
    if args.ref_loss:
        from gazenerf_amd.perceptual import VGGPerceptualLoss
        vgg = VGGPerceptualLoss(resize=True).to(dev)
    if args.gan_loss:
        from gazenerf_amd.gan import DiscriminatorStep
        gan = DiscriminatorStep(dev, lr=1e-4)          # rank-local, as in the reference (its discriminator is not under DDP either)
    batch_no = [0]
    clock = AllReduceClock(torch)
    timers = {k: StageTimer(k, pool=64) for k in ("fwd_mlp", "dgrad", "wgrad")}

    def reset(keep):
        for t in timers.values():
            t.reset(keep)
        clock.reset(keep)

    def step():
        opt.zero_grad(set_to_none=True)
        reducer.begin_step()              # nothing in flight, nothing half counted (an aborted step would leave both)
        for t in timers.values():
            t.arm()
        pred = net("train", p["xy"], None, None, p["shape_code"], p["appea_code"], p["gaze"], p["R"], p["T"], p["Kinv"],
                   t_rand=t_rand)["coarse_dict"]
        if gan is not None:
            gan.step(gt, face_mask, pred["merge_img"])
        loss = losses.total_loss(pred, gt, face_mask, full_eye, left_eye, right_eye, opt_codes, use_l1=args.ref_loss, epoch=1,
                                 discriminator=gan.discriminator if gan is not None else None, batch_num=batch_no[0],
                                 vgg=vgg, vgg_importance=1.0)["total_loss"]
        batch_no[0] += 1
        loss.backward()
        if dist:
            clock(reducer.all_reduce)
        opt.step()

    dt = timed_loop(ctx, step, reset)
    stage_ms = {k: t.collect() for k, t in timers.items()}
    ar = dist_info(ctx, reducer, clock)       # (before the probe steps below reset the clock: round 4 read it afterwards -- 0 calls timed)
    # two more, untimed, steps with the shader-clock probe armed (as in the cfg2b legs): at this launch size every stage starts right
    # after lower-power kernels and pays the power management's transition dip -- the probe shows it
    reset(False)
    with ClockProbe(dev) as probe:
        step()
        step()
        torch.cuda.synchronize()
    clocks = probe.mhz()
    if rank != 0:
        return None
    ms = dt / args.steps * 1e3

    # Outside the timed region: the upsampler alone on the 3B+1 stacked feature maps it sees in this
    # step -- the largest part of the step that is not the hot path.
    def time_upsampler():
        xs = torch.randn(3 * B + 1, 258, S, S, device=dev, requires_grad=True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf, tb = [], []
        for it in range(7):
            ev[0].record()
            y = net.neural_render(xs)
            ev[1].record()
            y.backward(torch.ones_like(y))
            ev[2].record()
            torch.cuda.synchronize(dev)
            if it >= 2:
                tf.append(ev[0].elapsed_time(ev[1]))
                tb.append(ev[0].elapsed_time(ev[2]))
        opt.zero_grad(set_to_none=True)
        return {"images": 3 * B + 1, "fwd_ms": sorted(tf)[len(tf) // 2], "fwdbwd_ms": sorted(tb)[len(tb) // 2],
                "kernels": "gnr::conv16_kernel<MT,NT,..> + gnr::wgrad2w_kernel / wgrad_kernel (image layout) + stencil / RGB kernels",
                "timing": "median of 5 calls of GazeNeRFNetAMD.neural_render on a random [3B+1,258,%d,%d] map, HIP events, outside the timed steps" % (S, S)}
    upsampler = time_upsampler()
    x3 = args.precision == "bf16x3"
    peak = PEAK_BF16X3_TFLOPS if x3 else PEAK_FP32_MFMA_TFLOPS
    m = B * n_rays * n_p
    sfx = chain_suffix(x3)
    stages = []
    for key, kernel, flop, mult in (("fwd_mlp", "gnr::fwd%s_kernel<true>" % sfx, m * 2 * FLOP_PER_SAMPLE_STREAM, 1),
                                    ("dgrad", "gnr::bwd%s_chain_kernel" % sfx, m * FLOP_PER_SAMPLE_STREAM, 2),
                                    ("wgrad", WGRAD_KERNEL[x3], m * FLOP_PER_SAMPLE_STREAM, 2)):
        a = mean(stage_ms[key])
        ach = flop / (a * 1e-3) / 1e12 if a > 0 else 0.0
        stages.append({"stage": key, "kernel": kernel, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                       "frac": ach / peak, "avg_ms": a, "launches_timed": len(stage_ms[key]), "flop_per_launch": flop,
                       "share_of_step": a * mult / ms})
        if key in clocks:
            stages[-1]["clock_mhz"] = clocks[key]
            stages[-1]["frac_at_clock"] = ach / (peak * clocks[key] / PEAK_CLOCK_MHZ)
    dom = max(stages, key=lambda s: s["share_of_step"])
    hot_flop = 3 * m * 2 * FLOP_PER_SAMPLE_STREAM
    res = {
        "metric": "rays/sec, whole-network training step (cfg4: B=2 x %dx%d rays x 64 samples per GPU, fwd+bwd+all-reduce+Adam)%s" % (
            S, S, " -- TEST-ONLY SIZE, not cfg4's 64x64" if test_size else ""),
        "value": world * B * n_rays * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16x3 hot path, f32 elsewhere" if x3 else "f32", "data": "synthetic",
        "images_per_s": world * B * args.steps / dt,
        "config": {"workload": "%s: global batch %d = %d images per GPU x %d GPUs, featmap %dx%d x 64 samples, whole network "
                               "(hot path -> merge -> upsampler x4 -> %dx%d image loss), backward, gradient all-reduce of "
                               "all %d trainable floats, fused Adam" % ("cfg4 at a TEST-ONLY size" if test_size else "cfg4", B * world, B, world,
                                                                        S, S, I, I, n_train),
                   "featmap_side": S, "test_only_size": test_size,
                   "rays_per_step_per_gpu": B * n_rays, "samples_per_ray": n_p, "trainable_floats": n_train,
                   "parallelism": "dp%d (images sharded; 3 flat buckets: NeuralRenderer (launched from autograd hooks, in "
                                  "flight during the hot path's backward), face MLP, eyes MLP)" % world,
                   "precision": args.precision,
                   "loss": ("masked L1 image terms + the three VGG-perceptual terms (the reference trainer's default switches; "
                            "randomly initialised VGG-16 extractor through MIOpen)" if args.ref_loss else
                            "masked L2 image terms (no pretrained-network term)") +
                           (" + PatchGAN: discriminator update and generator term" if args.gan_loss else "") + " + code regularisers"},
        "roofline": {"bound": "mfma", "kernel": dom["kernel"], "stage": dom["stage"], "achieved": dom["achieved"], "peak": peak,
                     "unit": "TFLOP/s", "frac": dom["frac"], "avg_launch_ms": dom["avg_ms"], "launches_timed": dom["launches_timed"],
                     "flop_per_launch": dom["flop_per_launch"], "traffic": None,
                     "traffic_source": "not captured for the cfg4 launch size",
                     "step_achieved": hot_flop / (ms * 1e-3) / 1e12,
                     "step_frac": hot_flop / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                     "step_frac_basis": "hot-path FLOPs only (3 x forward) over the whole step's wall time, fp32 MFMA peak"},
        "stages": stages,
        "upsampler": upsampler,
        "outside_hot_path_ms": ms - sum(st["avg_ms"] * mult for st, mult in zip(stages, (1, 2, 2))),
    }
    if ar:
        res["allreduce"] = ar
    return res


if __name__ == "__main__":
    main()
