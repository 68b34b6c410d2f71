#!/bin/bash
# Round 4, GPU session 13: full GPU test suite on the N1 changes (fused un-shuffle, RGB rider, 64-channel chain, wgrad16),
# upsampler traces, cfg4 and default bench lines.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s13
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
bash tools/n1_trace.sh r4s13/b1_graph --batch 1 --iters 9 --fwd-only > /dev/null 2>&1
bash tools/n1_trace.sh r4s13/b7 --batch 7 --iters 5 > /dev/null 2>&1
for n in b1_graph b7; do echo "== $n"; grep "N1 B" $O/$n/wall.log; grep -v "torch:" $O/$n/launches.txt; done > $O/n1_launches.txt
grep -E "N1 B|kernel time" $O/n1_launches.txt
rm -rf $O/*/prof
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'P'
import json
for f in ("bench_cfg4","bench_default"):
    try:
        d=json.loads(open('/root/repo/gpurun_out/r4s13/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('upsampler',{}).get('fwdbwd_ms'), d['roofline']['frac'], d['roofline'].get('step_frac'))
    except Exception as e: print(f, 'ERR', e)
P
echo done
