#!/bin/bash
# like ab_fwd.sh, 10 steps, prints stage times WITH the in-kernel clock probe; usage: tools/ab_fwd_clock.sh <tag> "<flags>"
set -u
export GNR_ALLOW_EXPERIMENTAL_LIB=1      # _lib.load() refuses a library built with timing switches otherwise
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; FLAGSX=$2
mkdir -p $R/gpurun_out/ab
cd $R
GNR_EXTRA_FILES="gnr_fwd16.hip" GNR_EXTRA_HIPCC_FLAGS="$FLAGSX" python -m gazenerf_amd.build --no-torch-ext > gpurun_out/ab/$TAG.build.log 2>&1
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-one-call 2> gpurun_out/ab/$TAG.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$TAG', 'ms_per_step', round(d['ms_per_step'],1), ' '.join('%s %.3f ms (%.3f) %s MHz' % (s['stage'], s['avg_ms'], s['frac'], round(s.get('clock_mhz') or 0)) for s in d.get('stages', [])))
" | tee -a gpurun_out/ab/summary_clock.txt
