#!/bin/bash
# Round 4, GPU session 12: idle gaps inside the cfg4 training step (kernel trace), then the cfg4 / default bench lines.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s12
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o run -- python $R/bench.py --config cfg4 --steps 4 --warmup 2 --no-cpu-baseline > $O/trace_run.log 2>&1
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
python $R/tools/step_gaps.py $O/kernel_trace.csv | tee $O/cfg4_gaps.txt
rm -rf $O/prof
cd $R
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<'P'
import json,sys
d=json.loads(open('/root/repo/gpurun_out/r4s12/bench_cfg4.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('ms_per_step','value','upsampler','outside_hot_path_ms')}); print([(s['stage'],round(s['frac'],4),round(s['avg_ms'],3)) for s in d['stages']])
P
echo done
