"""GPU: the product path under a guard-band allocator.

Every device allocation of the child process -- the workspaces, scratch buffers, saved activations and outputs the bindings
hand to gnr_fwd / gnr_bwd / gnr_upsample_* / gnr_merge_* / gnr_resample / gnr_sample_zvals, and torch's own temporaries --
is its own hipMalloc with 1 MiB of a known pattern on both sides (tests/guard/guard_alloc.cpp, plugged into torch by
tests/guard/run_guarded.py); the bands are checked after every libgnr call and at every free.  An out-of-bounds write of
any kernel within 1 MiB of a buffer it was given fails the scenario and names the call; `--poison` additionally fills every
new buffer with NaN bytes, so workspace read before it is written shows up as a non-finite (or wrong) result.

The shapes: the generators of tests/diagnostics/fuzz_hot_path_split.py / fuzz_upsample.py, the cfg4 step (7 stacked maps,
B = 2, whole network + Adam), and the more-images-than-workgroup-slots cases of commit 1503d29 (the out-of-bounds scratch
write round 5 found in gnr_wgrad.hip: that one landed INSIDE the caller's scratch and needed a results test; an overrun
past a buffer's end is what this file catches).  Long forms (more cases, both bindings): tools/session.sh guard."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
RUNNER = os.path.join(ROOT, "tests", "guard", "run_guarded.py")


def _guarded(*argv, expect_rc=0):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, RUNNER] + list(argv), cwd=ROOT, env=e, capture_output=True, text=True, timeout=1500)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, "no result line\n--- stderr tail ---\n" + r.stderr[-3000:]
    d = json.loads(lines[-1])
    assert r.returncode == expect_rc, "%s\n%s\n--- stderr tail ---\n%s" % (d.get("report"), d.get("error"), r.stderr[-3000:])
    return d


def test_the_harness_sees_a_one_element_overrun():
    """The allocator really is plugged in and really reports: three deliberate out-of-bounds writes (one float behind a
    tensor, one byte in front of it, four bytes 8 bytes past a second tensor that is then freed) = three violations."""
    d = _guarded("selftest")
    assert d["violations"] == 3 and d["found_by_check"] == 2 and d["allocations"] >= 2
    assert "TRAILING" in d["report"] and "LEADING" in d["report"] and "at free" in d["report"]
    assert d["guard_bytes_per_side"] == 1 << 20


@pytest.mark.parametrize("binding", ["ctypes", "torch_ext"])
def test_hot_path_over_random_shapes(binding):
    d = _guarded("hot_path", "--binding", binding, "--cases", "10", "--seed", "1" if binding == "ctypes" else "2")
    assert d["ok"] and d["violations"] == 0, d["report"]
    assert d["calls_checked"] >= 20 and all(c.startswith("ok") for c in d["cases"])


def test_hot_path_reads_no_workspace_it_did_not_write():
    """--poison: every fresh buffer is NaN bytes; the results must stay finite."""
    d = _guarded("hot_path", "--poison", "--cases", "6", "--seed", "3")
    assert d["ok"] and d["violations"] == 0 and d["poison"] is True, (d["report"], d["cases"])


def test_more_images_than_workgroup_slots():
    d = _guarded("many_images")
    assert d["ok"] and d["violations"] == 0, (d["report"], d["cases"])


@pytest.mark.parametrize("poison", [False, True])
def test_upsampler_over_random_shapes_and_the_cfg4_shape(poison):
    d = _guarded("upsample", "--cases", "10", "--seed", "4", *(["--poison"] if poison else []))
    assert d["ok"] and d["violations"] == 0, (d["report"], d["cases"])
    assert any("side 64" in c and "batch 7" in c for c in d["cases"])


def test_merge_resample_zvals_fine_pass_and_view_direction():
    d = _guarded("aux")
    assert d["ok"] and d["violations"] == 0, (d["report"], d["cases"])


@pytest.mark.parametrize("poison", [False, True])
def test_whole_network_training_step(poison):
    """cfg4's step (the configuration the 8-rank rehearsal of round 5 died in), fp32 and bf16x3, plus two small sides."""
    d = _guarded("network_step", *(["--poison"] if poison else []))
    assert d["ok"] and d["violations"] == 0, (d["report"], d["cases"])
    assert d["peak_live_bytes"] > 16 << 30          # the 16 GiB of saved activations really went through the guarded allocator
