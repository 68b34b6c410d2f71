#!/bin/bash
# Round 4, GPU session 30: the LDS-staged blur-fused feat_layers GEMM (conv16_blur_lds_kernel): parity tests, launch list at B = 7.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s30
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_upsample.py tests/test_network.py -m gpu -x -q 2>&1 | tail -3
bash tools/n1_trace.sh r4s30/b7 --batch 7 --iters 5 > /dev/null 2>&1
grep -E "N1 B" $O/b7/wall.log; grep -E "true>|blur_lds|kernel time" $O/b7/launches.txt
bash tools/n1_trace.sh r4s30/b1 --batch 1 --iters 9 --fwd-only > /dev/null 2>&1
grep -E "N1 B" $O/b1/wall.log; grep -E "true>|blur_lds|kernel time" $O/b1/launches.txt
rm -rf $O/*/prof
