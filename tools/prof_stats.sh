#!/bin/bash
# rocprofv3 kernel-trace stats of one bench.py run on the GPU box -> gpurun_out/<name>_kernel_stats.csv
# usage: tools/prof_stats.sh <name> <bench args...>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
mkdir -p $R/gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o $NAME -- python $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/$NAME.log 2>&1
f=$(find /tmp/prof_$NAME -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/${NAME}_kernel_stats.csv; head -16 "$f" | cut -c1-160; else echo "no stats file"; tail -5 $R/gpurun_out/$NAME.log; fi
