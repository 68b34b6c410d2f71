#!/bin/bash
# Round 4, GPU session 18: the upsampler backward's weight gradients on a helper stream (gnr_set_upsample_overlap) -- parity
# tests, A/B of the launch list / wall time at B = 7, split-batch experiment, cfg4 bench.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s18
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_upsample.py tests/test_network.py -m gpu -x -q > $O/pytest_n1.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_n1.log
for i in 1 2; do
N1_OVERLAP=1 timeout 300 python tools/n1_trace.py --batch 7 --iters 9 2>&1 | tail -1 | sed 's/^/overlap=1 /'
N1_OVERLAP=0 timeout 300 python tools/n1_trace.py --batch 7 --iters 9 2>&1 | tail -1 | sed 's/^/overlap=0 /'
done
N1_GRAPH=0 timeout 300 python tools/n1_trace.py --batch 7 --iters 9 --fwd-only 2>&1 | tail -1 | sed 's/^/fwd whole /'
N1_GRAPH=0 timeout 300 python tools/n1_trace.py --batch 7 --iters 9 --fwd-only --split 2 2>&1 | tail -1 | sed 's/^/fwd split2 /'
N1_GRAPH=0 timeout 300 python tools/n1_trace.py --batch 8 --iters 9 --fwd-only 2>&1 | tail -1 | sed 's/^/fwd whole /'
N1_GRAPH=0 timeout 300 python tools/n1_trace.py --batch 8 --iters 9 --fwd-only --split 2 2>&1 | tail -1 | sed 's/^/fwd split2 /'
timeout 300 python tools/n1_trace.py --batch 7 --iters 9 --split 2 2>&1 | tail -1 | sed 's/^/fwdbwd split2 /'
N1_OVERLAP=0 timeout 300 python tools/n1_trace.py --batch 7 --iters 9 --split 2 2>&1 | tail -1 | sed 's/^/fwdbwd split2 overlap=0 /'
bash tools/n1_trace.sh r4s18/b7 --batch 7 --iters 5 > /dev/null 2>&1
grep -E "N1 B|kernel time" $O/b7/launches.txt $O/b7/wall.log
rm -rf $O/b7/prof
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<'P'
import json
for f in ("bench_cfg4",):
    try:
        d=json.loads(open('/root/repo/gpurun_out/r4s18/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('upsampler',{}), d.get('outside_hot_path_ms'), d.get('build'))
    except Exception as e: print(f, 'ERR', e)
P
echo done
