#!/bin/bash
# Per-layer times of the upsampler's GEMMs for every conv16 tile variant (N1_CONV16_TILE -> gnr_set_conv16_tile), B = 7 forward+backward
# and B = 1 forward: the table the cost model in gnr_conv16.hip (conv16_plan) is fitted to.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/n1_sweep
for v in default 13,2 11,2 9,2 8,4 4,4 2,4; do
    if [ $v = default ]; then unset N1_CONV16_TILE; else export N1_CONV16_TILE=$v; fi
    n=$(echo $v | tr , _)
    $R/tools/n1_trace.sh n1_sweep/b7_$n --batch 7 --iters 3 > /dev/null 2>&1
    $R/tools/n1_trace.sh n1_sweep/b1_$n --batch 1 --iters 3 --fwd-only > /dev/null 2>&1
    echo "== $v: $(grep N1 $R/gpurun_out/n1_sweep/b7_$n/wall.log) | $(grep N1 $R/gpurun_out/n1_sweep/b1_$n/wall.log)"
    grep conv16_kernel $R/gpurun_out/n1_sweep/b7_$n/launches.txt | awk '{printf "%s ", $(NF-3)} END {print ""}'
    grep conv16_kernel $R/gpurun_out/n1_sweep/b1_$n/launches.txt | awk '{printf "%s ", $(NF-3)} END {print ""}'
done
