#!/bin/bash
# rocprofv3 kernel-trace + stats of a bench.py run on the GPU box; the stats CSV lands in gpurun_out/<name>/.
# usage: tools/prof_stats.sh <name> <bench args...>
# (--no-one-call: the extra one-call timing launches the same kernels at another size and would mix into the averages)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python $R/bench.py "$@" --no-cpu-baseline --no-one-call > $OUT/bench.log 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -40 $OUT/kernel_stats.csv
tail -c 600 $OUT/bench.log
