#!/bin/bash
# Round 4, GPU session 1: GPU tests, the driver-style bench, WRITE_SIZE calibration on the dump pattern, power / clock A/B
# of the training forward.  Run on the GPU box: gpurun --timeout 3300 -- 'bash tools/r4_session1.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s1
mkdir -p $O
export TMPDIR=/tmp
cd $R
STAGES=${STAGES:-"tests bench calib power"}
for st in $STAGES; do
case $st in
tests)
  timeout 1800 python -m pytest tests -m gpu -q --maxfail=12 --durations=15 > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
  tail -5 $O/pytest.log ;;
bench)
  timeout 600 python tools/smi_sample.py $O/smi_bench_default.csv -- python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
  tail -c 600 $O/bench_default.json; grep "^smi" $O/bench_default.json | tee -a $O/summary.txt ;;
calib)
  for m in "0 0" "0 200" "0 3000" "1 0" "1 200" "2 0" "3 0" "4 0"; do
    timeout 120 ./tools/ubench/store_pattern $m
  done 2>&1 | tee $O/store_pattern.txt
  cd /tmp
  for m in "0 0" "0 3000" "1 0" "1 3000" "2 0" "3 0" "4 0"; do
    tag=$(echo $m | tr ' ' _)
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/ws_$tag -o pmc -- $R/tools/ubench/store_pattern $m > $O/ws_$tag.log 2>&1
    python - "$O/ws_$tag" "$m" <<'PY' | tee -a $O/store_pattern.txt
import csv, glob, sys
vals = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE":
            vals.append(float(r["Counter_Value"]))
if vals:
    print("WRITE_SIZE mode/delay %s: %d dispatches, mean %.3f GB counted (KB x 1024)" % (sys.argv[2], len(vals), sum(vals) / len(vals) * 1024 / 1e9))
else:
    print("WRITE_SIZE mode/delay %s: no counter rows" % sys.argv[2])
PY
  done
  cd $R ;;
power)
  export GNR_ALLOW_EXPERIMENTAL_LIB=1
  for v in "shipped:" "nodump:-DGNR_ABL16=2" "nosign:-DGNR_ABL16=17" "bare:-DGNR_ABL16=31" "shipped2:"; do
    tag=${v%%:*}; fl=${v#*:}
    if [ -n "$fl" ]; then
      GNR_EXTRA_FILES="gnr_fwd16.hip" GNR_EXTRA_HIPCC_FLAGS="$fl" python -m gazenerf_amd.build --no-torch-ext > $O/build_$tag.log 2>&1
    else
      python -m gazenerf_amd.build --no-torch-ext > $O/build_$tag.log 2>&1
    fi
    timeout 400 python tools/smi_sample.py $O/smi_$tag.csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-one-call > $O/ab_$tag.json 2> $O/ab_$tag.err
    python - $O/ab_$tag.json $tag <<'PY' | tee -a $O/power_ab.txt
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[2], "ms_per_step %.1f" % d["ms_per_step"], " ".join("%s %.3f ms (%.3f) %s MHz" % (s["stage"], s["avg_ms"], s["frac"], round(s.get("clock_mhz") or 0)) for s in d.get("stages", [])))
    elif l.startswith("smi"):
        print(sys.argv[2], l.strip())
PY
  done ;;
esac
done
echo done
