#!/bin/bash
# Round 4, GPU session 6: staggered first round of the training forward against the transition clock dip (experiment), then the
# shipped build: tests, bench, cfg4.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s6
mkdir -p $O
export TMPDIR=/tmp
cd $R
export GNR_ALLOW_EXPERIMENTAL_LIB=1
for n in 0 30 100 300; do
  if [ $n = 0 ]; then python -m gazenerf_amd.build --no-torch-ext > $O/build_$n.log 2>&1
  else GNR_EXTRA_FILES="gnr_fwd16.hip" GNR_EXTRA_HIPCC_FLAGS="-DGNR_RAMP_SLEEPS=$n" python -m gazenerf_amd.build --no-torch-ext > $O/build_$n.log 2>&1; fi
  echo "== ramp $n" | tee -a $O/ramp.txt
  timeout 200 python tools/stage_loop.py alt --seconds 8 --rays 8192 2>> $O/err.txt | tee -a $O/ramp.txt
  timeout 200 python tools/stage_loop.py alt --seconds 8 --rays 32768 2>> $O/err.txt | tee -a $O/ramp.txt
done
unset GNR_ALLOW_EXPERIMENTAL_LIB
python -m gazenerf_amd.build > $O/build_shipped.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt; tail -3 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - $O <<'PY' | tee -a $O/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], "value %.0f ms %.1f step_frac %.4f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["step_frac"], d["roofline"].get("traffic")),
                  " ".join("%s %.3f (%.3f) %s" % (s["stage"], s["avg_ms"], s["frac"], round(s.get("clock_mhz") or 0)) for s in d["stages"]))
PY
echo done
