#!/bin/bash
# Round 4, GPU session 32: end-of-round record on the final build -- full GPU suite, upsampler launch lists, bench lines, kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s32
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/n1_trace.sh r4s32/b1_graph --batch 1 --iters 9 --fwd-only > /dev/null 2>&1
bash tools/n1_trace.sh r4s32/b7 --batch 7 --iters 5 > /dev/null 2>&1
{ echo "# tools/r4_session32.sh (final build of round 4): tools/n1_trace.sh <name> --batch 1 --iters 9 --fwd-only (HIP-graph replay) and --batch 7 --iters 5"
  for n in b1_graph b7; do echo "== $n"; grep "N1 B" $O/$n/wall.log; grep -v "torch:" $O/$n/launches.txt; done; } > $O/n1_launches.txt
grep -E "N1 B|kernel time" $O/n1_launches.txt
rm -rf $O/*/prof
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --precision bf16x3 > $O/bench_cfg4_bf16x3.json 2> $O/bench_cfg4_bf16x3.err
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
bash tools/prof_stats.sh r4s32/stats > /dev/null 2>&1
rm -rf $O/stats/prof
python - <<'P'
import json
for f in ("bench_cfg4","bench_cfg4_bf16x3","bench_default"):
    try:
        d=json.loads(open('/root/repo/gpurun_out/r4s32/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('upsampler',{}).get('fwdbwd_ms'), d['roofline']['frac'], d['roofline'].get('step_frac'), d.get('build'))
    except Exception as e: print(f, 'ERR', e)
P
echo done
