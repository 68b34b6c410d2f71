#!/bin/bash
# Round 4, GPU session 15: fused un-shuffle for channel counts that are no multiple of 4 (dpre2 from the GEMM, dres collected from it)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s15
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_upsample.py tests/test_network.py -m gpu -x -q 2>&1 | tail -2
for v in default 2,8 3,8 4,8; do
  if [ $v = default ]; then unset N1_CONV16_TILE; else export N1_CONV16_TILE=$v; fi
  n=$(echo $v | tr , _)
  bash tools/n1_trace.sh r4s15/b7_$n --batch 7 --iters 5 > /dev/null 2>&1
  echo "== tile $v: $(grep 'N1 B' $O/b7_$n/wall.log)"
  grep -E "unshuffle|kernel time" $O/b7_$n/launches.txt
done
rm -rf $O/*/prof
