#!/bin/bash
# rocprofv3 kernel trace of the upsampler alone; prints the launches of the last iteration in order.
# usage: tools/n1_trace.sh <name> [n1_trace.py args...]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp
timeout 600 python $R/tools/n1_trace.py "$@" > $OUT/wall.log 2>&1
cat $OUT/wall.log
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o run -- python $R/tools/n1_trace.py "$@" > $OUT/run.log 2>&1
find $OUT/prof -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python $R/tools/n1_trace_list.py $OUT/kernel_trace.csv | tee $OUT/launches.txt
