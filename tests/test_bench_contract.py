"""GPU: bench.py end to end at a small size -- exactly one JSON line on stdout with the contract's fields, the
roofline / cpu_baseline objects and the separately timed bf16x3 leg; and the 2-rank launcher path on one GPU
(gloo override, test-only) producing the whole-job aggregate."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip() and not l.startswith("[Gloo]")]   # gloo's own chatter (test-only backend)
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["fwdbwd", "fwd"])
def test_single_gpu_line(mode):
    # the CPU leg (thread-count probing on a many-core host) takes a minute: keep it in the cheaper forward mode only
    extra = ["--cpu-rays", "16"] if mode == "fwd" else ["--no-cpu-baseline"]
    d = _run([sys.executable, "bench.py", "--side", "64", "--steps", "2", "--warmup", "1", "--mode", mode] + extra)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "rays/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 64 * 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3 and rf["traffic"] is not None
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.1 < rf["frac"] < 1.0
    if mode == "fwd":
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["value_1_thread"] > 0 and "sample" in cb
    x3 = d["bf16x3"]
    assert x3["value"] > d["value"] and x3["roofline"]["kernel"].startswith("gnr::fwd3_kernel")
    if mode == "fwdbwd":
        assert d["roofline_hbm"]["bound"] == "hbm" and d["roofline_hbm"]["kernel"] == "gnr::comp_bwd_kernel"


def test_two_ranks_on_one_gpu_report_the_aggregate():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29533", "bench.py", "--gpus", "2", "--side", "64", "--steps", "1", "--warmup", "1", "--no-alt"],
             env={"GNR_BENCH_DEVICE": "0", "GNR_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 64 * 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
