"""GPU: bench.py end to end at a small size -- exactly one JSON line on stdout with the contract's fields, the
roofline / stages / cpu_baseline objects and the separately timed bf16x3 leg; `--gpus 2` WITHOUT a launcher spawning
two ranks itself (gloo override, both on one GPU: test-only) and reporting the whole-job aggregate with n_gpus = 2; the
same under an external torch.distributed.run; the cfg4 whole-network step with all 5 015 714 gradients exchanged."""
import json
import os
import re
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "stages"}
TWO_ON_ONE = {"GNR_BENCH_DEVICE": "0", "GNR_BENCH_BACKEND": "gloo"}


def _free_port():
    """A port that was free a moment ago (bound, read, released) -- never a fixed number: a collision on a shared host is one
    more way to lose the run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _why(stderr):
    """What a failed bench.py run said about ITSELF: every `bench.py: rank k of N failed: ...` line with the traceback behind
    the first one, the pre-flight memory lines, then the tail.  torch.distributed.run's summary table, which is all the last
    2000 characters hold, names the ranks but not the error (round 5 lost the cause of a red driver run that way)."""
    lines = stderr.splitlines()
    failed = [i for i, l in enumerate(lines) if re.search(r"bench\.py: rank \d+ of \d+ failed", l)]
    out = ["--- ranks that failed (%d) ---" % len(failed)] + [lines[i] for i in failed]
    if failed:
        out += ["--- traceback behind the first ---"] + lines[failed[0] + 1:failed[0] + 40]
    first_tb = next((i for i, l in enumerate(lines) if l.startswith("Traceback (most recent call last)")), None)
    if first_tb is not None and not failed:
        out += ["--- first traceback ---"] + lines[first_tb:first_tb + 40]
    out += ["--- memory pre-flight ---"] + [l for l in lines if "hipMemGetInfo" in l][:16]
    out += ["--- stderr tail ---", stderr[-1500:]]
    return "\n".join(out)


def _run(cmd, env=None, strict=False):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, _why(r.stderr)
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]   # launcher / gloo chatter is not JSON
    assert len(lines) == 1, r.stdout[-2000:]
    if strict:      # NOTHING but the line on stdout (round 5: RCCL's NCCL_DEBUG=VERSION banner used to follow it -- bench.guard_stdout)
        assert r.stdout.strip() == lines[0].strip(), r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["fwdbwd", "fwd"])
def test_single_gpu_line(mode):
    extra = ["--cpu-budget", "6"] if mode == "fwd" else ["--no-cpu-baseline"]
    d = _run([sys.executable, "bench.py", "--side", "64", "--steps", "2", "--warmup", "1", "--mode", mode] + extra, strict=True)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "rays/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 64 * 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3
    # 4096 rays per launch is not a size a PMC capture is committed for: traffic must be null, not a stale number
    assert rf["traffic"] is None and "traffic_source" in rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.1 < rf["frac"] < 1.0
    stages = {s["stage"]: s for s in d["stages"]}
    assert set(stages) == ({"fwd_mlp", "dgrad", "wgrad", "comp_bwd"} if mode == "fwdbwd" else {"fwd_mlp"})
    for s in stages.values():
        assert s["launches_timed"] == 2 and s["avg_ms"] > 0 and 0.0 < s["frac"] < 1.0
        assert abs(s["frac"] - s["achieved"] / s["peak"]) < 1e-9
        if s["stage"] != "comp_bwd":      # shader clock under the stage's kernels (in-kernel probe): a plausible gfx950 clock
            assert 500.0 < s["clock_mhz"] <= 2600.0 and abs(s["frac_at_clock"] - s["frac"] * 2400.0 / s["clock_mhz"]) < 1e-9
    assert 0.5 < sum(s["share_of_step"] for s in stages.values()) <= 1.0 + 1e-6
    if mode == "fwd":
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["value_1_thread"] > 0 and "sample" in cb
    x3 = d["bf16x3"]
    assert x3["value"] > d["value"] and any(t in x3["roofline"]["kernel"] for t in ("fwd3", "bwd3", "wgrad3"))
    if mode == "fwdbwd":
        assert d["roofline_hbm"]["bound"] == "hbm" and d["roofline_hbm"]["kernel"] == "gnr::comp_bwd_kernel"


def test_gpus_flag_launches_the_ranks_itself():
    """python bench.py --gpus 2, no torchrun: two ranks, n_gpus 2, whole-job aggregate, the exchange described."""
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--side", "64", "--steps", "1", "--warmup", "1", "--no-alt"], env=TWO_ON_ONE)
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 64 * 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    # N > 1 ranks park instead of polling when the runtime makes them wait (round 6); one rank keeps the runtime's default
    assert d["host"]["sync"]["asked"] == "auto" and d["host"]["sync"]["mode"] == "blocking"
    assert len(d["host"]["sync"]["hipSetDevice_hipSetDeviceFlags_rc"]) == 2      # (the two return codes are in the line; a refusal is not fatal)
    ar = d["allreduce"]
    assert ar["world_size_formed"] == 2 and ar["floats"] == 2 * 1518979 and ar["buckets"] == 2 and ar["ms"] > 0


def test_two_ranks_under_an_external_launcher_report_the_aggregate():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--side", "64", "--steps", "1", "--warmup", "1", "--no-alt"],
             env=TWO_ON_ONE)
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 64 * 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def test_gpus_flag_must_match_the_world_size():
    e = dict(os.environ)
    e.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--side", "64", "--steps", "1"], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_cfg4_whole_network_step_reduces_every_gradient():
    d = _run([sys.executable, "bench.py", "--config", "cfg4", "--gpus", "2", "--steps", "2", "--warmup", "1"], env=TWO_ON_ONE)
    assert d["n_gpus"] == 2 and d["config"]["trainable_floats"] == 5015714
    ar = d["allreduce"]
    assert ar["floats"] == 5015714 and ar["buckets"] == 3 and ar["world_size_formed"] == 2
    assert abs(d["value"] - 2 * 2 * 4096 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert {s["stage"] for s in d["stages"]} == {"fwd_mlp", "dgrad", "wgrad"}
    up = d["upsampler"]                       # N1 alone on the step's 3B+1 stacked maps, timed outside the steps
    assert up["images"] == 7 and 0 < up["fwd_ms"] < up["fwdbwd_ms"] and 0 < d["outside_hot_path_ms"] < d["ms_per_step"]


@pytest.mark.parametrize("mode", ["fwd", "fwdbwd"])
def test_strong_scaling_shards_one_image_over_the_ranks(mode):
    """--scaling strong (the reference's B=1 render loop utils/render_utils.py:199-219): ONE image, its
    rays in contiguous blocks per rank; value = that image's rays per second, NOT multiplied by the world size."""
    extra = ["--gather"] if mode == "fwd" else []
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--scaling", "strong", "--mode", mode, "--side", "64", "--steps", "2",
              "--warmup", "1", "--no-alt", "--micro", "1024"] + extra, env=TWO_ON_ONE)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["rays_per_step_per_gpu"] == 64 * 64 // 2
    assert abs(d["value"] - 64 * 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert "ONE image" in d["config"]["workload"] and d["config"]["parallelism"].startswith("rays2")
    assert d["distributed"]["world_size_formed"] == 2 and d["distributed"]["devices_visible"] >= 1
    if mode == "fwdbwd":
        assert d["allreduce"]["floats"] == 2 * 1518979
        assert {s["stage"]: s["launches_timed"] for s in d["stages"]}["fwd_mlp"] == 2 * 2      # 2 micro-batches x 2 steps


def test_one_call_entry_and_preflight_errors():
    d = _run([sys.executable, "bench.py", "--side", "64", "--steps", "1", "--warmup", "1", "--no-alt", "--no-cpu-baseline",
              "--micro", "1024"])
    oc = d["one_call"]
    assert oc["value"] > 0 and oc["calls_timed"] == 2
    # the library's tiled entry point (what the timed step runs) is reported beside the one-call path
    assert abs(oc["ms_per_step_tiled_entry"] - d["ms_per_step"]) <= 1e-9 and "render_two_stream_tiled" in oc["tiled_entry"]
    assert d["build"].startswith("src=") and "experimental=0" in d["build"]
    assert oc["tiled_in_op"] is False and oc["flop_factor_vs_step"] == 1.0          # 4096 rays fit the budget: no tiling
    # the same call with a budget the 4096-ray image exceeds: the op tiles, 4/3 of the FLOPs
    d = _run([sys.executable, "bench.py", "--side", "64", "--steps", "1", "--warmup", "1", "--no-alt", "--no-cpu-baseline",
              "--micro", "1024"], env={"GNR_WS_BUDGET_GB": "2"})
    assert d["one_call"]["tiled_in_op"] is True and abs(d["one_call"]["flop_factor_vs_step"] - 4 / 3) < 1e-9
    # a rank whose GPU does not exist fails in pre-flight with one line, and the job exits non-zero
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update({"GNR_BENCH_DEVICE": "63"})
    r = subprocess.run([sys.executable, "bench.py", "--side", "64", "--steps", "1"], cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0 and "needs GPU 63" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, "bench.py", "--config", "cfg4", "--scaling", "strong"], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


# ----------------------------------------------------------------------------------------------- world size 8 (round 5)
@pytest.mark.parametrize("what", ["cfg4", "strong"])
def test_world_size_8_rehearsal_through_the_self_launcher(what):
    """the launcher path had only ever run at world size 2.  Eight ranks on ONE GPU through
    `bench.py --gpus 8` itself (gloo override: eight ranks cannot share a device under RCCL) -- rendezvous, port, timeouts,
    one line from rank 0 only, n_gpus 8, every trainable float exchanged, per-rank CPU affinity and the host enqueue time
    reported -- so the driver's first 8-GPU RCCL run measures instead of debugging.  Step being scaled:
    trainer/gazenerf_trainer.py:478-534 (cfg4); the B = 1 render loop utils/render_utils.py:199-219 (strong)."""
    if what == "cfg4":
        # the LAUNCHER is under test, not 8 x 16 GiB of saved activations on one device (round 5: the driver's box lost three
        # ranks at the full size): cfg4's test-only 16 x 16 feature map (128 x 128 images), 0.3 GiB per rank
        d = _run([sys.executable, "bench.py", "--gpus", "8", "--config", "cfg4", "--side", "16", "--steps", "2", "--warmup", "1"],
                 env=TWO_ON_ONE)
        n_train = 5015714 - 258 * (64 * 64 - 16 * 16)                 # bg_featmap [1, 258, S, S] is a parameter
        assert d["config"]["trainable_floats"] == n_train and d["config"]["test_only_size"] is True and "TEST-ONLY" in d["metric"]
        ar = d["allreduce"]
        assert ar["floats"] == n_train and ar["bytes"] == 4 * n_train and ar["buckets"] == 3 and ar["world_size_formed"] == 8
        assert ar["calls_timed"] == 2 and ar["ms"] > 0
        assert abs(d["value"] - 8 * 2 * 256 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    else:
        d = _run([sys.executable, "bench.py", "--gpus", "8", "--scaling", "strong", "--side", "128", "--micro", "1024", "--steps", "1",
                  "--warmup", "1", "--no-alt"], env=TWO_ON_ONE)
        assert d["scaling"] == "strong" and d["config"]["rays_per_step_per_gpu"] == 128 * 128 // 8
        assert d["allreduce"]["floats"] == 2 * 1518979 and d["allreduce"]["world_size_formed"] == 8
        assert abs(d["value"] - 128 * 128 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
        assert {s["stage"]: s["launches_timed"] for s in d["stages"]}["fwd_mlp"] == 2        # 2048 rays per rank / 1024
    assert d["n_gpus"] == 8 and "cpu_baseline" not in d and d["distributed"]["world_size_formed"] == 8
    host = d["host"]
    enq = host["host_enqueue_ms_per_step"][0]
    assert 0 < enq["ms"] <= enq["ms_per_step"] * 1.001 and abs(enq["ms_per_step"] - d["ms_per_step"]) < 1e-6
    assert 0 < enq["cpu_ms"] and 0 < enq["cpu_share"]
    aff = host["cpu_affinity"]                   # rank 0's block of the CPUs the job may use
    assert aff is not None and (aff.get("pinned") is False or aff["n_cpus"] >= 1)


def test_host_enqueue_time_is_reported_at_one_rank():
    d = _run([sys.executable, "bench.py", "--side", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-one-call"])
    legs = {e["leg"]: e for e in d["host"]["host_enqueue_ms_per_step"]}
    assert set(legs) == {"fp32", "bf16x3"} and all(0 < e["ms"] <= e["ms_per_step"] * 1.001 and e["cpu_ms"] > 0 for e in legs.values())
    assert d["host"]["cpu_affinity"] is None           # one rank keeps the whole mask (the CPU baseline needs it)
    assert d["host"]["sync"] == {"asked": "auto", "mode": "auto", "hipSetDevice_hipSetDeviceFlags_rc": None}
    hb = d["roofline_hbm"]                             # priced on the algorithm's 258 channels, the layout's 288 beside it
    assert abs(hb["frac_incl_padding"] / hb["frac"] - hb["bytes_per_launch_incl_padding"] / hb["bytes_per_launch"]) < 1e-9
    assert hb["bytes_per_launch"] == 4096 * 64 * (258 * 4 + 16) + 4096 * (258 * 4 + 12)
    assert hb["bytes_per_launch_read"] == 4096 * 64 * (272 * 4 + 16) + 4096 * (272 * 4 + 12)          # what comp_bwd_kernel reads
    assert hb["channels"] == {"algorithmic": 258, "read_by_kernel": 272, "saved_layout": 288}


# ----------------------------------------------------------------------------------------------- RCCL (round 4)
def test_a_one_rank_rccl_communicator_forms_and_carries_the_gradient_buckets():
    """VERDICT round 3, missing #2: until round 4 no test had ever taken the `nccl` branch of bench.py -- every multi-rank
    test forces gloo.  A ONE-rank RCCL process group needs no second GPU: GNR_BENCH_FORCE_DIST=1 makes `bench.py --gpus 1`
    call init_process_group("nccl", world_size=1, device_id=...), run the pre-flight all-reduce and push the real
    GradAllReducer buckets (2 x 1 518 979 floats at cfg2b) through the communicator every step.  That loads librccl,
    creates a communicator, exercises the hand-off between the compute stream and RCCL's, and shows whether
    HSA_ENABLE_IPC_MODE_LEGACY=0 is harmless."""
    d = _run([sys.executable, "bench.py", "--side", "64", "--steps", "2", "--warmup", "1", "--no-alt", "--no-cpu-baseline",
              "--no-one-call"], env={"GNR_BENCH_FORCE_DIST": "1"}, strict=True)
    dd = d["distributed"]
    assert dd["backend"] == "nccl" and dd["world_size_formed"] == 1 and dd["forced_at_world_size_1"] is True
    assert dd["rccl_version"] and dd["rccl_version"][0].isdigit()
    ar = d["allreduce"]
    assert ar["backend"].startswith("rccl") and ar["floats"] == 2 * 1518979 and ar["buckets"] == 2
    assert ar["calls_timed"] == 2 and ar["ms"] > 0
    assert d["n_gpus"] == 1 and abs(d["value"] - 64 * 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    # the same through the whole network: 3 buckets, the NeuralRenderer one launched from autograd hooks
    d = _run([sys.executable, "bench.py", "--config", "cfg4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
             env={"GNR_BENCH_FORCE_DIST": "1"}, strict=True)
    assert d["distributed"]["backend"] == "nccl" and d["allreduce"]["floats"] == 5015714 and d["allreduce"]["buckets"] == 3


def test_one_rank_rccl_exchange_leaves_the_gradients_unchanged():
    """GradAllReducer(force_collective=True) over a one-rank RCCL group: sum over one rank, divided by one -- the gradients
    that come back are the gradients that went in, bit for bit (flatten -> all_reduce on RCCL's stream -> copy back,
    with the stream synchronisation torch.distributed inserts)."""
    code = r"""
import os, sys, socket
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from gazenerf_amd.parallel import GradAllReducer
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = torch.Generator(device="cpu").manual_seed(0)
params = [torch.nn.Parameter(torch.randn(n, generator=g).to(dev)) for n in (1518979, 384 * 384, 7, 1)]
for p in params:
    p.grad = torch.randn(p.shape, generator=g).to(dev)
want = [p.grad.clone() for p in params]
red = GradAllReducer([params[:1], params[1:]], 1, force_collective=True)
red.all_reduce()
torch.cuda.synchronize()
assert all(torch.equal(p.grad, w) for p, w in zip(params, want))
off = GradAllReducer([params], 1)          # default at world size 1: no collective at all
assert not off.active
off.all_reduce()
print("OK", ".".join(str(v) for v in torch.cuda.nccl.version()))
dist.destroy_process_group()
"""
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-c", code, ROOT], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("what", ["weak", "strong", "cfg4"])
def test_two_ranks_over_rccl_when_the_node_has_two_gpus(what):
    """The contract tests above WITHOUT the gloo override, one rank per GPU over RCCL -- runs wherever the box exposes
    >= 2 devices (the driver's 8-GPU node; the one-GPU boxes of a build round skip with the reason printed), so the
    8-GPU scaling run is never the first time an N-rank RCCL communicator is formed."""
    n = _n_gpus()
    if n < 2:
        pytest.skip("this box exposes %d GPU(s): an N-rank RCCL communicator needs one device per rank" % n)
    if what == "cfg4":
        d = _run([sys.executable, "bench.py", "--config", "cfg4", "--gpus", "2", "--steps", "2", "--warmup", "1"])
        assert d["allreduce"]["floats"] == 5015714 and d["allreduce"]["buckets"] == 3
        assert abs(d["value"] - 2 * 2 * 4096 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    else:
        d = _run([sys.executable, "bench.py", "--gpus", "2", "--scaling", what, "--side", "64", "--steps", "2", "--warmup", "1",
                  "--no-alt", "--micro", "1024"])
        rays = 64 * 64 * (2 if what == "weak" else 1)
        assert abs(d["value"] - rays / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
        assert d["allreduce"]["floats"] == 2 * 1518979
    assert d["n_gpus"] == 2 and d["distributed"]["backend"] == "nccl" and d["distributed"]["world_size_formed"] == 2
    assert d["distributed"]["rccl_version"]
