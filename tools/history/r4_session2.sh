#!/bin/bash
# Round 4, GPU session 2 (after the S16 dump layout): tests, bench at two micro-batch sizes, gradient-noise draws,
# per-stage power / clock, WRITE_SIZE of the training kernels, LDS counters of the bf16x3 kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s2
mkdir -p $O
export TMPDIR=/tmp
cd $R
STAGES=${STAGES:-"tests bench noise stage pmc x3lds"}
for st in $STAGES; do
case $st in
tests)
  timeout 1800 python -m pytest tests -m gpu -q --maxfail=12 --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
  tail -5 $O/pytest.log ;;
bench)
  timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
  timeout 600 python bench.py --steps 10 --warmup 3 --micro 32768 --no-cpu-baseline --no-alt --no-one-call > $O/bench_micro32k.json 2> $O/bench_micro32k.err
  timeout 600 python bench.py --steps 10 --warmup 3 --micro 49152 --no-cpu-baseline --no-alt --no-one-call > $O/bench_micro48k.json 2> $O/bench_micro48k.err
  timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
  python - $O <<'PY' | tee -a $O/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], "value %.0f ms %.1f step_frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["step_frac"]),
                  " ".join("%s %.3f (%.3f) %s" % (s["stage"], s["avg_ms"], s["frac"], round(s.get("clock_mhz") or 0)) for s in d["stages"]),
                  ("outside %.2f up %s" % (d["outside_hot_path_ms"], d["upsampler"])) if "upsampler" in d else "")
PY
  ;;
noise)
  timeout 900 python tests/diagnostics/grad_noise_draws.py 8 > $O/grad_noise_draws.txt 2> $O/grad_noise_draws.err; tail -6 $O/grad_noise_draws.txt ;;
stage)
  for s in fwd bwd; do
    timeout 200 python tools/smi_sample.py $O/smi_stage_$s.csv -- python tools/stage_loop.py $s --seconds 12 2> $O/stage_$s.err | tee -a $O/stage_power.txt
  done ;;
pmc)
  timeout 900 python tools/pmc_capture.py r4s2/pmc > $O/pmc_capture.log 2>&1; tail -25 $O/pmc_capture.log ;;
x3lds)
  bash tools/pmc_lds_x3.sh r4s2/x3lds > $O/x3lds.log 2>&1; tail -30 $O/x3lds.log ;;
esac
done
echo done
