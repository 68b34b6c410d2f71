#!/bin/bash
# Round 4, GPU session 9: sanity of the rebuilt tree (pytest -m gpu), half-width GEMM tiles at B=1/2, full B=7 trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s9
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
export N1_GRAPH=0
for v in default 2,4 2,2 4,2 4,4; do
  if [ $v = default ]; then unset N1_CONV16_TILE; else export N1_CONV16_TILE=$v; fi
  n=$(echo $v | tr , _)
  for b in 1; do
    bash tools/n1_trace.sh r4s9/b${b}_$n --batch $b --iters 5 --fwd-only > /dev/null 2>&1
    echo "== tile $v B=$b: $(grep 'N1 B' $O/b${b}_$n/wall.log)" | tee -a $O/small_tiles.txt
    grep conv16_kernel $O/b${b}_$n/launches.txt | awk '{printf "%s%s ", $1, $(NF-3)} END {print ""}' | tee -a $O/small_tiles.txt
  done
done
unset N1_CONV16_TILE
bash tools/n1_trace.sh r4s9/b7 --batch 7 --iters 5 > /dev/null 2>&1
cat $O/b7/launches.txt | head -150
rm -rf $O/*/prof
echo done
