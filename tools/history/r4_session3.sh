#!/bin/bash
# Round 4, GPU session 3: tests, bench at the new default tile (32768) and at 65536, cfg4, kernel stats, PMC traffic at the
# bench's launch size, stage loops per launch size with power, bf16x3 leg under the power sampler, N1 traces.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s3
mkdir -p $O
export TMPDIR=/tmp
cd $R
STAGES=${STAGES:-"tests bench stats pmc sizes x3power n1"}
for st in $STAGES; do
case $st in
tests)
  timeout 1800 python -m pytest tests -m gpu -q --maxfail=12 --durations=5 > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
  tail -4 $O/pytest.log ;;
bench)
  timeout 600 python tools/smi_sample.py $O/smi_bench_default.csv -- python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
  timeout 600 python bench.py --steps 10 --warmup 3 --micro 65536 --no-cpu-baseline --no-alt --no-one-call > $O/bench_micro64k.json 2> $O/bench_micro64k.err
  timeout 600 python bench.py --steps 10 --warmup 3 --micro 16384 --no-cpu-baseline --no-alt --no-one-call > $O/bench_micro16k.json 2> $O/bench_micro16k.err
  timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
  timeout 600 python bench.py --mode fwd --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_fwd.json 2> $O/bench_fwd.err
  python - $O <<'PY' | tee -a $O/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], "value %.0f ms %.1f step_frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["step_frac"]),
                  " ".join("%s %.3f (%.3f) %s" % (s["stage"], s["avg_ms"], s["frac"], round(s.get("clock_mhz") or 0)) for s in d["stages"]),
                  ("outside %.2f up fwd %.3f fwdbwd %.3f" % (d["outside_hot_path_ms"], d["upsampler"]["fwd_ms"], d["upsampler"]["fwdbwd_ms"])) if "upsampler" in d else "",
                  ("one_call %.1f" % d["one_call"]["ms_per_step"]) if "one_call" in d else "", ("x3 %.0f" % d["bf16x3"]["value"]) if "bf16x3" in d else "")
        elif l.startswith("smi"):
            print(l.strip())
PY
  ;;
stats)
  bash tools/prof_stats.sh r4s3/stats > $O/stats.log 2>&1; head -12 $O/stats/kernel_stats.csv | cut -c1-150 ;;
pmc)
  timeout 900 python tools/pmc_capture.py r4s3/pmc > $O/pmc_capture.log 2>&1; grep "fwd16\|bwd16\|comp_bwd\|wgrad2w" $O/pmc_capture.log ;;
sizes)
  for s in fwd bwd alt; do for r in 8192 16384 32768; do
    timeout 200 python tools/smi_sample.py $O/smi_${s}_$r.csv -- python tools/stage_loop.py $s --seconds 8 --rays $r 2>> $O/sizes.err | tee -a $O/sizes.txt
  done; done ;;
x3power)
  timeout 400 python tools/smi_sample.py $O/smi_x3.csv -- python bench.py --precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-one-call > $O/bench_x3.json 2> $O/bench_x3.err
  grep "^smi" $O/bench_x3.json | tee -a $O/summary.txt ;;
n1)
  bash tools/n1_trace.sh r4s3/n1_b1f --batch 1 --iters 7 --fwd-only > $O/n1_b1f.log 2>&1; tail -28 $O/n1_b1f.log
  N1_GRAPH=0 bash tools/n1_trace.sh r4s3/n1_b1f_eager --batch 1 --iters 7 --fwd-only > $O/n1_b1f_eager.log 2>&1; grep "N1 B" $O/n1_b1f_eager.log
  bash tools/n1_trace.sh r4s3/n1_b7 --batch 7 --iters 5 > $O/n1_b7.log 2>&1; grep "N1 B\|kernel time" $O/n1_b7.log ;;
esac
done
echo done
