#!/bin/bash
# Round 4, GPU session 19: split-K reductions queued and run as one launch per weight set / per upsampler backward (gnr_wgrad.h).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s19
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
bash tools/n1_trace.sh r4s19/b7 --batch 7 --iters 5 > /dev/null 2>&1
grep -E "N1 B" $O/b7/wall.log; grep -E "reduce|kernel time" $O/b7/launches.txt
rm -rf $O/b7/prof
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'P'
import json
for f in ("bench_cfg4","bench_default"):
    try:
        d=json.loads(open('/root/repo/gpurun_out/r4s19/%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(d['value'],1), round(d['ms_per_step'],3), d.get('upsampler',{}).get('fwdbwd_ms'), [ (s['stage'], round(s['frac'],4), round(s['avg_ms'],3)) for s in d['stages']], d['roofline'].get('step_frac'), d.get('build'))
    except Exception as e: print(f, 'ERR', e)
P
echo done
