#!/bin/bash
# Round 4, GPU session 7: the final tree -- GPU tests, smoke, bench, kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s7
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1800 python -m pytest tests -m gpu -q --maxfail=8 --durations=5 > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt; tail -3 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee -a $O/summary.txt; tail -2 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
bash tools/prof_stats.sh r4s7/stats > $O/stats.log 2>&1; head -8 $O/stats/kernel_stats.csv | cut -c1-140
python - $O <<'PY' | tee -a $O/summary.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], "value %.0f ms %.1f step_frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["step_frac"]),
                  " ".join("%s %.3f (%.3f) %s" % (s["stage"], s["avg_ms"], s["frac"], round(s.get("clock_mhz") or 0)) for s in d["stages"]))
PY
echo done
