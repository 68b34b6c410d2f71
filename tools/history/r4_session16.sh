#!/bin/bash
# Round 4, GPU session 16: upsampler tests + B = 7 trace, selected kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s16
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_upsample.py tests/test_network.py -m gpu -x -q 2>&1 | tail -2
bash tools/n1_trace.sh r4s16/b7 --batch 7 --iters 5 > /dev/null 2>&1
grep 'N1 B' $O/b7/wall.log
grep -E "${1:-rgb_bwd}|kernel time" $O/b7/launches.txt
rm -rf $O/*/prof
