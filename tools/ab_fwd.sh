#!/bin/bash
# like ab_variants.sh but prints only the forward stage; usage: tools/ab_fwd.sh <tag> "<flags>"
set -u
export GNR_ALLOW_EXPERIMENTAL_LIB=1      # _lib.load() refuses a library built with timing switches otherwise
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; FLAGSX=$2
mkdir -p $R/gpurun_out/ab
cd $R
GNR_EXTRA_FILES="gnr_fwd16.hip" GNR_EXTRA_HIPCC_FLAGS="$FLAGSX" python -m gazenerf_amd.build --no-torch-ext > gpurun_out/ab/$TAG.build.log 2>&1
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-one-call 2> gpurun_out/ab/$TAG.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$TAG', 'ms_per_step', round(d['ms_per_step'],1), ' '.join('%s %.3f ms (%.3f)' % (s['stage'], s['avg_ms'], s['frac']) for s in d.get('stages', [])))
" | tee -a gpurun_out/ab/summary_fwd.txt
