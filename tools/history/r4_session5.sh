#!/bin/bash
# Round 4, GPU session 5: dump-store burst size of the training forward after the S16 layout (stage loop, then the bench for
# the two best) -- and the same switch for the dgrad kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s5
mkdir -p $O
export TMPDIR=/tmp
export GNR_ALLOW_EXPERIMENTAL_LIB=1
cd $R
for b in 1 2 4 8 12 24; do
  GNR_EXTRA_FILES="gnr_fwd16.hip,gnr_bwd16.hip" GNR_EXTRA_HIPCC_FLAGS="-DGNR_DUMP_BURST=$b" python -m gazenerf_amd.build --no-torch-ext > $O/build_$b.log 2>&1
  echo "== burst $b" | tee -a $O/burst.txt
  timeout 200 python tools/stage_loop.py alt --seconds 10 --rays 32768 2>> $O/err.txt | tee -a $O/burst.txt
done
for b in 8 4; do
  GNR_EXTRA_FILES="gnr_fwd16.hip,gnr_bwd16.hip" GNR_EXTRA_HIPCC_FLAGS="-DGNR_DUMP_BURST=$b" python -m gazenerf_amd.build --no-torch-ext > $O/build_$b.log 2>&1
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-one-call > $O/bench_burst$b.json 2> $O/bench_burst$b.err
done
python -m gazenerf_amd.build --no-torch-ext > $O/build_shipped.log 2>&1
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-one-call > $O/bench_burst1.json 2> $O/bench_burst1.err
python - $O <<'PY' | tee -a $O/burst.txt
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], "value %.0f ms %.1f step_frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["step_frac"]),
                  " ".join("%s %.3f (%.3f) %s" % (s["stage"], s["avg_ms"], s["frac"], round(s.get("clock_mhz") or 0)) for s in d["stages"]))
PY
echo done
