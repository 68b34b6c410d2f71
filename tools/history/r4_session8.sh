#!/bin/bash
# Round 4, GPU session 8: half-width GEMM tiles (2x2, 4x2) for the upsampler's single-image forward.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s8
mkdir -p $O
export TMPDIR=/tmp
cd $R
export N1_GRAPH=0
for v in default 2,4 2,2 4,2 4,4 9,2; do
  if [ $v = default ]; then unset N1_CONV16_TILE; else export N1_CONV16_TILE=$v; fi
  n=$(echo $v | tr , _)
  for b in 1 2; do
    bash tools/n1_trace.sh r4s8/b${b}_$n --batch $b --iters 5 --fwd-only > /dev/null 2>&1
    echo "== tile $v B=$b: $(grep 'N1 B' $O/b${b}_$n/wall.log)" | tee -a $O/small_tiles.txt
    grep conv16_kernel $O/b${b}_$n/launches.txt | awk '{printf "%s%s ", $1, $(NF-3)} END {print ""}' | tee -a $O/small_tiles.txt
  done
done
echo done
