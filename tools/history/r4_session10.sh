#!/bin/bash
# Round 4, GPU session 10: the du GEMM with the fused un-shuffle epilogue -- parity, then B = 7 traces per row-tile instance.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s10
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_upsample.py tests/test_e2e.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -15 $O/pytest.log
for v in default 2,8 4,8 2,4; do
  if [ $v = default ]; then unset N1_CONV16_TILE; else export N1_CONV16_TILE=$v; fi
  n=$(echo $v | tr , _)
  bash tools/n1_trace.sh r4s10/b7_$n --batch 7 --iters 5 > /dev/null 2>&1
  echo "== tile $v: $(grep 'N1 B' $O/b7_$n/wall.log)" | tee -a $O/unshuffle.txt
  grep -E "conv16_unshuffle|unshuffle_bwd" $O/b7_$n/launches.txt | tee -a $O/unshuffle.txt
done
rm -rf $O/*/prof
echo done
