"""GPU: bench.py end to end at a small size -- exactly one JSON line on stdout with the contract's fields, the
roofline / stages / cpu_baseline objects and the separately timed bf16x3 leg; `--gpus 2` WITHOUT a launcher spawning
two ranks itself (gloo override, both on one GPU: test-only) and reporting the whole-job aggregate with n_gpus = 2; the
same under an external torch.distributed.run; the cfg4 whole-network step with all 5 015 714 gradients exchanged."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "stages"}
TWO_ON_ONE = {"GNR_BENCH_DEVICE": "0", "GNR_BENCH_BACKEND": "gloo"}


def _run(cmd, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]   # launcher / gloo chatter is not JSON
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["fwdbwd", "fwd"])
def test_single_gpu_line(mode):
    extra = ["--cpu-budget", "6"] if mode == "fwd" else ["--no-cpu-baseline"]
    d = _run([sys.executable, "bench.py", "--side", "64", "--steps", "2", "--warmup", "1", "--mode", mode] + extra)
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
    ar = d["allreduce"]
    assert ar["world_size_formed"] == 2 and ar["floats"] == 2 * 1518979 and ar["buckets"] == 2 and ar["ms"] > 0


def test_two_ranks_under_an_external_launcher_report_the_aggregate():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29533", "bench.py", "--gpus", "2", "--side", "64", "--steps", "1", "--warmup", "1", "--no-alt"],
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
