#!/usr/bin/env python
"""Energy per step of the bf16x3 leg (GPU box): J / step = mean socket power over the timed steps x ms_per_step.

    python tools/x3_energy.py [--steps 40] [--precision bf16x3|fp32]

Runs `bench.py --precision <p> --steps N --warmup 5 --no-cpu-baseline --no-one-call --no-alt` under tools/smi_sample.py
(hwmon of the GPU's own PCI device, ~20 Hz), takes the busy samples (power above the idle third) as the timed region and
prints power, clock, ms per step, J per step, J per ray and the stage fractions of the bench line.  The chain kernels are
power-limited (profiles/r4_x3_power.txt): a change that saves instructions shows up as J / step, not necessarily as ms."""
import argparse
import json
import os
import statistics
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--precision", default="bf16x3")
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="x3e_")
csv_path, out_path = os.path.join(tmp, "smi.csv"), os.path.join(tmp, "bench.json")
cmd = [sys.executable, os.path.join(ROOT, "tools", "smi_sample.py"), csv_path, "--", sys.executable, os.path.join(ROOT, "bench.py"),
       "--precision", a.precision, "--steps", str(a.steps), "--warmup", "5", "--no-cpu-baseline", "--no-one-call", "--no-alt"]
r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
if not lines:
    sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
    raise SystemExit("x3_energy: no bench line")
d = json.loads(lines[-1])
rows = []
with open(csv_path) as f:
    keys = f.readline().strip().split(",")
    for ln in f:
        v = ln.strip().split(",")
        rows.append({k: float(x) for k, x in zip(keys, v) if x != ""})
pw = [q["power_uW"] / 1e6 for q in rows if "power_uW" in q]
thr = min(pw) + (max(pw) - min(pw)) / 3.0
busy = [q for q in rows if q.get("power_uW", 0) / 1e6 >= thr]
# the timed steps are the LAST ms_per_step * steps seconds of the busy region (warm-up and the clock-probe step come before / after:
# the probe step is one more step at the same power, so the mean over the tail window is the steps' power)
span = d["ms_per_step"] * 1e-3 * a.steps
t_end = busy[-1]["t"]
win = [q for q in busy if q["t"] >= t_end - span * 1.02]
p_mean = statistics.mean(q["power_uW"] / 1e6 for q in win)
clk = statistics.median(q["sclk_Hz"] / 1e6 for q in win if "sclk_Hz" in q)
j_step = p_mean * d["ms_per_step"] * 1e-3
print("build: %s" % d.get("build"))
print("%s leg, %d steps: %.1f ms per step, %.1f k rays/s; socket power mean %.0f W over %d samples (min %.0f, max %.0f), sclk median %.0f MHz"
      % (a.precision, a.steps, d["ms_per_step"], d["value"] / 1e3, p_mean, len(win), min(q["power_uW"] for q in win) / 1e6,
         max(q["power_uW"] for q in win) / 1e6, clk))
print("energy: %.1f J per step = %.3f mJ per ray" % (j_step, j_step / (d["config"]["rays_per_step_per_gpu"]) * 1e3))
for s in d["stages"]:
    print("  stage %-8s %-52s avg %.3f ms  frac %.3f  clock %s MHz" % (s["stage"], s["kernel"][:52], s["avg_ms"], s["frac"],
                                                                        ("%.0f" % s["clock_mhz"]) if s.get("clock_mhz") else "-"))
print([l for l in r.stdout.splitlines() if l.startswith("smi ")][-1] if any(l.startswith("smi ") for l in r.stdout.splitlines()) else "")
